// engine_types.h — device data layout of the self-play engine (shared by host and kernels).
//
// HBM layout (sized for 288 GB; numbers for Othello, 4096 concurrent games, 800 traversals/move):
//   arenas   [slots + spare][node_cap] x 1 KiB nodes — one bump arena per game, no per-move compaction:
//            re-rooting (UCTTree::advanceDecision, uct/UCTTree.hpp:197-210) is O(1) by bumping the game's
//            epoch, which lazily turns every active node gray (clearSubtree :283-298); pruned siblings are
//            simply never reached again.  ~33 k nodes are created per game, so node_cap = 45056 -> 44 MiB
//            per game, 176 GiB for 4096 games.  If an arena fills, the live subtree is Cheney-copied into a
//            spare arena (rare; exercised by tests with tiny caps).
//   node     1 KiB: rows N[64] W[64] P[64] f32 (lane a <-> action a: one coalesced 256 B access per row),
//            child[64] u16, 64 B header (bitboards, legal mask, cached value, pass-edge stats, flags).
//            The cached network policy lives in the P row (it is only ever consumed through expand()).
//   paths    [slots][MAXQ][MAX_DEPTH] u32 (node<<8 | action) for leaves waiting on the network.
//   nn_in    [slots*maxQueue][2H+1][R][C] f32 sparse staging -> nn_dense (slot-major compaction of the real
//            leaves, deterministic order) -> nn_logits [leaves][A], nn_value [leaves]
//   records  compact per ply: 2 x u64 bitboards, mover, tempered pdf f32[A]; winner per game.
#ifndef SPRL_ENGINE_TYPES_H
#define SPRL_ENGINE_TYPES_H

#include <stdint.h>

#define SPRL_NODE_BYTES 1024
#define SPRL_ROW 64
#define SPRL_NONE16 0xFFFFu
#define SPRL_NONE24 0xFFFFFFu
#define SPRL_MAXQ 8
#define SPRL_FCACHE 32      // ready-to-use recycled node ids kept per game (>= max_batch + 2 for recycling to be on)

enum { F_EVAL = 1, F_TERMINAL = 2, F_PASS = 4 };
enum { ST_IDLE = 0, ST_ACTIVE = 1, ST_ERROR = 2, ST_FRESH = 3 };
enum { EVAL_RANDOM = 0, EVAL_HEURISTIC = 1, EVAL_NETWORK = 2 };
enum { MASK_REFERENCE = 0, MASK_SYMMETRISED = 1 };
enum {
    ERR_NONE = 0,
    ERR_ARENA_FULL = 1,     // live subtree does not fit the arena even after compaction
    ERR_NO_SPARE = 2,       // no spare arena free for compaction
    ERR_MAX_PLIES = 3,      // game longer than the record capacity
    ERR_MAX_DEPTH = 4,      // search line longer than the path capacity
};

struct NodeHdr {            // 64 bytes at node + 896
    uint64_t p0, p1;        // stones of Player::ZERO / Player::ONE
    uint64_t legal;         // legal lane-actions of the side to move
    float value;            // cached network value (UCTNode::m_networkValue)
    uint32_t exp_epoch;     // node is active iff exp_epoch == game epoch (UCTNode::m_isExpanded)
    float passN, passW, passP;
    uint32_t passChild;
    uint8_t player, flags;
    int8_t winner;
    uint8_t pad0;
    uint16_t action;        // action that led to this node (Go: pass detection for the double-pass end)
    uint16_t depth;         // plies since the start of the game (Go depth cap)
    uint32_t pad2[2];
};

struct GameStats {
    unsigned long long traversals, levels, expansions, nn_evals, terminal_hits, gray_hits, dup_hits,
        nodes_created, compactions, games, plies, max_alloc, nodes_recycled;
    // shader-clock cycles per phase, filled only by the diagnostic build (-DSPRL_PHASE_TIMERS), else 0
    unsigned long long cyc_total, cyc_finish, cyc_move, cyc_select, cyc_create, cyc_backup, cyc_leafio,
        cyc_noise, cyc_max, cyc_lvl_wait, cyc_lvl_pick, cyc_lvl_desc;
};

struct GameCtl {
    uint64_t rng_state, rng_inc;
    uint32_t status;
    uint32_t game_id;
    uint32_t arena;         // arena index in the pool
    uint32_t root;          // decision node
    uint32_t n_alloc;
    uint32_t epoch;         // ply + 1
    float rootN, rootW;     // decision node's own N()/W() (lives in its parent's row in the reference, Q6)
    int32_t ply;
    int32_t traversals;
    int32_t n_leaves;
    uint32_t root_player;
    uint32_t leaf_node[SPRL_MAXQ];
    uint32_t leaf_depth[SPRL_MAXQ];
    uint32_t leaf_sym[SPRL_MAXQ];
    uint32_t leaf_player[SPRL_MAXQ];
    // node recycling (single-strip kernel): ids of pruned subtrees' roots wait on the game's reclaim stack
    // (EngineParams::reclaim); a refill pops some, pushes their children and leaves the popped ids here, ready for reuse
    uint32_t rstack_n;      // height of the reclaim stack
    uint32_t fc_n;          // valid entries of fcache
    uint32_t fcache[SPRL_FCACHE];
    GameStats stats;
};

struct Counters {
    uint32_t next_game;     // next global game index to start
    uint32_t games_done;
    uint32_t error;         // first ERR_* raised by any game
    uint32_t error_game;
    uint32_t active_slots;  // slots still holding an unfinished game after the last step
    uint32_t leaf_total;    // leaves queued for the network by the last step (dense batch size)
    unsigned long long leaf_rows;   // running sum of leaf_total over the launches of a run (rows evaluated by the network)
    uint32_t active_last;   // network rounds: active_slots of the last step, moved here by the leaf scan, which also clears
    uint32_t pad_;          // active_slots for the next step (no separate memset launch per round)
};

struct alignas(32) Mailbox {   // match play: the move a side has just made, handed to the partner tree of the same game
    uint64_t ply_launch;    // low 32: index of the ply that was played, plus one (0 = nothing yet); high 32: sequence number
                            // of the launch that wrote it.  One 64-bit word: an entry is consumed only by a LATER launch
                            // (kernel boundary = the only cross-XCD visibility the protocol relies on)
    uint64_t rng_state;     // the game's RNG stream moves with the turn (one global stream in the reference)
    uint32_t game;
    uint32_t action;
    uint32_t pad[2];
};

struct EngineParams {
    int32_t num_traversals, max_batch, max_queue;
    float dir_eps, dir_alpha, u_weight;
    int32_t early_cutoff;
    float early_exp, rest_exp;
    int32_t use_sym, add_noise, eval_kind, mask_frame;
    int32_t stream_base;
    uint64_t seed;
    int32_t node_cap, num_slots, num_spare, num_games, max_plies, rounds;
    int32_t max_depth, planes;
    float resign_threshold; // 0 = off (not in the reference): resign when sum W / sum N at the decision node < -threshold
    int32_t resign_min_ply;
    int32_t go_legal_form;  // wide Go kernel, legal-move computation: 0 by board size, 1 group labels, 2 per-lane flood fill
    int32_t init_q_zero;    // 0 = InitQ::PARENT (workers), 1 = InitQ::ZERO (uct/UCTNode.hpp:24-28,267-273)
    int32_t wide_idx;       // 1: child indices are 24 bits (u16 row + u8 row in the node's last 64 bytes): node_cap > 65535
    uint32_t none_idx;      // "no child" in registers: 0xFFFF / 0xFFFFFF
    uint32_t alloc_base;    // first node index a new game uses (0; a test hook moves it next to the 16-bit boundary)
    int32_t recycle;        // 1: nodes of pruned siblings are reused (reclaim stack); 0: bump allocation + compaction only
    uint32_t* reclaim;      // [num_slots][node_cap] reclaim stacks
    // match play (Evaluate.cpp): per-agent options, agent = slot & 1, game = slot >> 1
    int32_t m_use_sym[2], m_eval_kind[2], m_init_q_zero[2];
    uint32_t launch_seq;
    Mailbox* mailbox;       // [num_slots][2]: entry (game sequence of the pair) & 1
    int16_t* match_actions; // [num_games][max_plies]
    uint8_t* arenas;
    uint32_t* arena_used;   // [num_slots + num_spare] 0 free / 1 used
    GameCtl* ctl;
    uint32_t* paths;
    float* nn_in;           // sparse staging [slot][max_queue][planes][cells], written by the tree kernel
    uint32_t* leaf_count;   // [num_slots] leaves queued by each slot in the last step
    uint32_t* leaf_offset;  // [num_slots] exclusive prefix sum of leaf_count = row of the slot's first leaf
    float* nn_dense;        // dense network batch [leaf_total (rounded up)][planes][cells]
    const float* nn_logits;
    const float* nn_value;
    uint64_t* hist_boards;  // Go: [num_slots][HIST_CAP][2] positions of the real game so far (superko, history planes)
    uint64_t* rec_boards;
    uint8_t* rec_movers;
    float* rec_pdf;
    int32_t* rec_nplies;
    int32_t* rec_offsets;   // [num_games + 1] exclusive prefix sum of rec_nplies (records_kernel.h), filled by the records scan
    int8_t* rec_winner;
    uint32_t* rec_evals;    // [num_games] network evaluations each game queued (self-play runs; null in match play)
    Counters* counters;
};

#endif  // SPRL_ENGINE_TYPES_H
