/* sprl_amd.h — C ABI of the MI355X-native self-play engine (libsprl_amd.so).
 *
 * Drop-in boundary for the reference's self-play worker path.  The reference has no FFI; each entry
 * point below replaces the C++ template interface cited next to it (paths relative to
 * /root/reference/cpp/src).  Plain pointers and sizes only — no torch or HIP types.  See INTEGRATION.md
 * for the reference-side binding a maintainer would add.
 *
 * Conventions: every call returns 0 on success or a negative SPRL_E_* code and records a message
 * retrievable with sprl_last_error() (thread-local).  No exception crosses the boundary.  One engine
 * per GPU; calls on one engine must be serialised by the caller; engines on different GPUs are
 * independent.  The engine REQUIRES a gfx950 device: there is no CPU fallback, creation fails loudly.
 */
#ifndef SPRL_AMD_H
#define SPRL_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPRL_E_CONFIG   (-1)  /* bad configuration value */
#define SPRL_E_MODEL    (-2)  /* model file could not be loaded / evaluator plugin missing */
#define SPRL_E_NODEPOOL (-3)  /* a game's node arena overflowed (raise node_cap / spare_arenas) */
#define SPRL_E_DEVICE   (-4)  /* HIP error, no device */
#define SPRL_E_STATE    (-5)  /* call sequence error */
#define SPRL_E_IO       (-6)  /* file could not be written */
#define SPRL_E_LIMIT    (-7)  /* game longer than max_plies / line deeper than the path buffer */
#define SPRL_E_NOMEM    (-8)  /* device memory: the arenas (sprl_engine_create) or the record buffers of a run (sprl_engine_begin /
                                 sprl_engine_run) do not fit - retry with fewer concurrent_games */

/* SPRL_GO7: Go as the reference compiles it — 7x7, komi 9.0, 8-ply history, positional superko, depth cap 98
 * (games/GoNode.hpp:16-22) */
enum { SPRL_OTHELLO = 0, SPRL_CONNECT_FOUR = 1, SPRL_GO7 = 2,
       /* the same Go rules at 9x9 (komi 7.5), rows of two wavefront strips (wide kernel); SPRL_GO7_WIDE runs the
        * reference-size game through that wide kernel (validation of the generic code path) */
       SPRL_GO9 = 3, SPRL_GO19 = 4 /* 19x19, rows of six strips */, SPRL_GO7_WIDE = 5 };
/* evaluator kinds: the reference's RandomNetwork (networks/RandomNetwork.hpp:15-57), OthelloHeuristic
 * (networks/OthelloHeuristic.cpp:5-53) and GridNetwork on a traced TorchScript file
 * (networks/GridNetwork.hpp:37-145) */
enum { SPRL_EVAL_RANDOM = 0, SPRL_EVAL_HEURISTIC = 1, SPRL_EVAL_NETWORK = 2 };
/* the reference applies the un-symmetrised action mask to the symmetrised policy
 * (uct/UCTTree.hpp:138,146-147,152).  REFERENCE reproduces that; SYMMETRISED is the repaired behaviour. */
enum { SPRL_MASK_REFERENCE = 0, SPRL_MASK_SYMMETRISED = 1 };

/* Replaces the compile-time constants of OTHWorker.cpp:12-28 / C4Worker.cpp:11-27 / constants.hpp:4-10
 * and the arguments of runIteration (selfplay/SelfPlay.hpp:204-209). */
typedef struct sprl_config {
    int32_t game;              /* SPRL_OTHELLO | SPRL_CONNECT_FOUR | SPRL_GO7 | SPRL_GO9 | SPRL_GO19 | SPRL_GO7_WIDE */
    int32_t device;            /* HIP device ordinal */
    int32_t concurrent_games;  /* game slots resident in HBM (one wavefront each) */
    int32_t num_traversals;    /* UCT traversals per move, lower bound (SelfPlay.hpp:100) */
    int32_t max_batch;         /* traversals per search batch (UCTTree.hpp:82) */
    int32_t max_queue;         /* network leaves per search batch, <= 8 (UCTTree.hpp:108) */
    float dir_eps;             /* Dirichlet mixing weight (UCTNode.hpp:341-343) */
    float dir_alpha;           /* Dirichlet concentration */
    float u_weight;            /* constants.hpp:6 */
    int32_t early_cutoff;      /* constants.hpp:8 */
    float early_exp;           /* constants.hpp:9 */
    float rest_exp;            /* constants.hpp:10 */
    int32_t use_symmetry;      /* symmetrizer != nullptr */
    int32_t add_noise;         /* runIteration's addNoise */
    int32_t mask_frame;        /* SPRL_MASK_* */
    int32_t node_cap;          /* nodes per game arena (1 KiB each; Go 9x9 2 KiB, 19x19 5.5 KiB); 0 = default.  Othello / Connect Four /
                                  Go 7x7: < 2^24 (child indices widen from 16 to 24 bits above 65535), nodes of pruned siblings are
                                  reused, default min(4 x, 2 x + 4096) num_traversals + 1024.  Boards wider than 8x8: <= 65535 */
    int32_t spare_arenas;      /* arenas kept free for compaction; 0 = default */
    int32_t max_plies;         /* record capacity per game; 0 = default */
    uint64_t seed;             /* game g uses Random(seed, stream_base + g) (utils/random.hpp:92-103); a later run on the same
                                  engine continues the numbering (its game g: stream_base + games of earlier runs + g) */
    int32_t stream_base;       /* must be >= 1 */
    int32_t profile;           /* 1: time every tree-kernel launch with HIP events on its stream, the trunk convolutions of a forward
                                  with one pair around all of them; 2: one pair per convolution launch (kernel durations; for samples
                                  outside a timed region - two more queue packets per launch) */
    int32_t own_stream;        /* 1: the engine works on a private non-blocking HIP stream instead of the null stream, so that
                                  several engines driven from different host threads overlap on one GPU */
    float resign_threshold;    /* NOT in the reference (BASELINE config 5), 0 = off: after a search the side to move resigns
                                  when the mean backed-up value of its decision node (sum W / sum N over its edges) is below
                                  -resign_threshold; the ply's sample is kept, the opponent wins */
    int32_t resign_min_ply;    /* no resignation before this ply */
    int32_t no_recycle;        /* 1: never reuse the nodes of pruned siblings (bump allocation + compaction only; tests) */
    int32_t alloc_base;        /* 0 = default.  Tests: every game starts allocating at this node id, so that a short game's ids
                                  cross the 16-bit boundary (single-strip kernels, alloc_base + max_batch + 8 < node_cap) */
} sprl_config;

/* Fills `cfg` with the reference worker's constants for `game` (OTHWorker.cpp:24-28, C4Worker.cpp:23-27,
 * GoWorker.cpp:23-27, constants.hpp:6-10), 800 traversals, 4096 concurrent games, seed 1. */
int sprl_config_default(int32_t game, sprl_config* cfg);

typedef struct sprl_engine sprl_engine;

/* UCTTree + selfPlay state for `concurrent_games` games on one GPU (uct/UCTTree.hpp:38-53). */
int sprl_engine_create(const sprl_config* cfg, sprl_engine** out);
void sprl_engine_destroy(sprl_engine* e);

/* INetwork selection (selfplay/GridWorker.hpp:123-131).  `model` is "random", "heuristic" or the path of a
 * traced TorchScript module forward(float32[B,2H+1,R,C]) -> (float32[B,A], float32[B,1])
 * (networks/GridNetwork.hpp:99-102), which is evaluated on the GPU through LibTorch-ROCm. */
int sprl_engine_set_model(sprl_engine* e, const char* model);

/* The same from a TorchScript archive in host memory (torch.jit.save(traced, buffer)): the trainer hot-swaps the model
 * in-process, no file and no polling (selfplay/GridWorker.hpp:35-55 is the file rendez-vous this replaces). */
int sprl_engine_set_model_buffer(sprl_engine* e, const void* torchscript_bytes, int64_t nbytes);

/* Human-readable description of the evaluator in use (which network execution path was selected). */
int sprl_engine_evaluator_info(sprl_engine* e, char* buf, int32_t len);

/* Alternative evaluator hook: `fn` is called once per search round with DEVICE pointers
 * planes float32[batch][2H+1][R][C] -> logits float32[batch][A], value float32[batch]; it must enqueue its
 * work on the HIP null stream (or synchronise before returning). */
typedef int (*sprl_forward_fn)(void* user, const float* planes, int32_t batch, float* logits, float* value);
int sprl_engine_set_forward(sprl_engine* e, sprl_forward_fn fn, void* user);

/* Match play between two agents — the data-parallel form of cpp/src/Evaluate.cpp:37-170 (two UCTTrees per game,
 * agents/UCTNetworkAgent.hpp:45-108, interface/play.hpp:22-60).  Game i seeds Random(seed, stream_base + i); agent k
 * moves first in the games with i % 2 == k (Evaluate.cpp:126-130).  Tree options as in Evaluate.cpp:94-112: the
 * caller passes dir_eps 0.25 / dir_alpha 0.1 / add_noise 1 / u_weight in `cfg`; per agent: symmetrisation and
 * InitQ.  `cfg->concurrent_games` game PAIRS of trees are resident at once.  Every game of the engine (since round 3 also
 * Go 9x9 / 19x19 through the wide kernel).
 * Outputs (host memory, caller-owned): winners[num_games] (0, 1, -1 = draw; colour, not agent),
 * nplies[num_games], actions[num_games][max_plies] (-1 padded). */
enum { SPRL_INITQ_PARENT = 0, SPRL_INITQ_ZERO = 1 };
typedef struct sprl_match_agent {
    const char* model;          /* "random", "heuristic" (Othello) or a traced model file; unused when forward != NULL */
    sprl_forward_fn forward;    /* optional forward hook (device pointers), as sprl_engine_set_forward */
    void* forward_user;
    int32_t use_symmetry;       /* Evaluate.cpp:49,51 */
    int32_t init_q;             /* Evaluate.cpp:50,52: useParentQ -> SPRL_INITQ_PARENT, else SPRL_INITQ_ZERO */
} sprl_match_agent;
int sprl_match_play(const sprl_config* cfg, const sprl_match_agent* agent0, const sprl_match_agent* agent1,
                    int32_t num_games, int8_t* winners, int32_t* nplies, int16_t* actions, int32_t max_plies);

/* Compact self-play records of a run, host memory owned by the library until sprl_records_free.
 * Sample order = the reference's: game-major, ply-major (SelfPlay.hpp:86-92,127-133,154-163).  */
typedef struct sprl_records {
    int32_t game, num_games, rows, cols, cells, actions, nsym, use_symmetry;
    int32_t history, planes;    /* H plies of history in a state (GridState.hpp:74-118), planes = 2H + 1 */
    int64_t total_plies;
    const int32_t* ply_offset;  /* [num_games + 1] */
    const int8_t* boards;       /* [total_plies][cells]  -1 empty, 0, 1 (GridState.hpp:18-22) */
    const int8_t* movers;       /* [total_plies] */
    const float* pdfs;          /* [total_plies][actions] tempered visit pdf (SelfPlay.hpp:111-121) */
    const int8_t* winners;      /* [num_games] -1 draw, 0, 1 */
    void* owner_;
} sprl_records;

/* runIteration (selfplay/SelfPlay.hpp:204-248): plays `num_games` games to the end. Blocking. */
int sprl_engine_run(sprl_engine* e, int32_t num_games, sprl_records* out);

/* The same in pieces (bench / pipelining): begin, then step until *games_done == num_games, then collect.
 * One step = `rounds` search rounds for every resident game (select kernel [+ network forward] + finish). */
int sprl_engine_begin(sprl_engine* e, int32_t num_games);
int sprl_engine_step(sprl_engine* e, int32_t rounds, int32_t* games_done, int32_t* active_slots);
int sprl_engine_collect(sprl_engine* e, sprl_records* out);
void sprl_records_free(sprl_records* r);
/* Network evaluations queued by each of the first `num_games` games of the current / last self-play run (a leaf that reaches
 * GridNetwork::evaluate in the reference, networks/GridNetwork.hpp:72-107; game index = the order the games were started in).
 * The per-game view of sprl_stats.nn_evals: what a distributional comparison with the reference's selfPlay needs. */
int sprl_engine_game_evals(sprl_engine* e, uint32_t* out, int32_t num_games);

/* Finished-game records WITHOUT a host copy (after sprl_engine_step has reported every game done, instead of / before
 * sprl_engine_collect).  All pointers are DEVICE memory owned by the caller.
 *   records_info   total plies, samples (= plies x nsym when the engine symmetrises) and the size of the packed form;
 *   pack_records   the compact wire format of the multi-GPU gather - head int64[12], offsets int32[games+1], winners int8[games],
 *                  stones uint64[plies][words] x 2, movers uint8[plies], pdfs float32[plies][actions], sections 16-byte aligned
 *                  (sprl_amd/distributed.py) - so that ranks hand RCCL a device buffer;
 *   expand_records the reference worker's samples (selfplay/SelfPlay.hpp:86-92,127-133,151-189 + the plane encoding of
 *                  selfplay/GridWorker.hpp:146-171): states float32[N][2H+1][R][C], distributions float32[N][A], outcomes
 *                  float32[N], written straight into e.g. the trainer's replay window;
 *   finish         ends the run (like collect, without producing host records). */
int sprl_engine_records_info(sprl_engine* e, int64_t* total_plies, int64_t* num_samples, int64_t* packed_bytes);
int sprl_engine_pack_records(sprl_engine* e, void* dst_device, int64_t capacity_bytes);
int sprl_engine_expand_records(sprl_engine* e, float* states_device, float* distributions_device, float* outcomes_device,
                               int64_t capacity_samples);
int sprl_engine_finish(sprl_engine* e);

typedef struct sprl_stats {
    int64_t games, plies, traversals, levels, expansions, nn_evals, terminal_hits, gray_hits, dup_hits,
        nodes_created, compactions, max_nodes_in_arena;
    int64_t rounds, kernel_launches, nn_batches, nn_rows;   /* nn_rows = rows evaluated by the network incl. bucket padding */
    double seconds_total;      /* wall time inside run/step */
    double kernel_ms;          /* sum of tree-kernel durations (HIP events; profile=1) */
    double nn_ms;              /* sum of network forward durations (HIP events; with the count left on the device only in profile=2) */
    int64_t hbm_bytes;         /* device memory allocated by the engine */
    /* shader-clock cycles summed over game slots, per phase; 0 unless built with -DSPRL_PHASE_TIMERS (diagnosis) */
    int64_t cyc_total, cyc_finish, cyc_move, cyc_select, cyc_create, cyc_backup, cyc_leafio, cyc_noise,
        cyc_max_slot_launch, cyc_lvl_wait, cyc_lvl_pick, cyc_lvl_desc;
    /* trunk-convolution kernel of the hand-written CNN (HIP events around every launch; profile=1) */
    double conv_ms;
    int64_t conv_launches, conv_boards;
    int64_t nodes_recycled;    /* node creations served from pruned siblings' nodes instead of fresh arena space */
} sprl_stats;
int sprl_engine_stats(sprl_engine* e, sprl_stats* out);

/* profile=1, several engines in one process (one HIP stream each): launches of the same kernel overlap in time, so the sum of
 * their durations (sprl_stats.kernel_ms / conv_ms) no longer says how long the device spent on them.  busy_ms = the time since
 * sprl_profile_busy_reset() during which at least one launch of the kind was executing, over ALL engines of the process, from
 * the same HIP events; sum_ms = the sum of the durations.  kind 0 = tree kernel, 1 = trunk convolution of the network plugin.
 * Call sprl_engine_stats on every engine first (it resolves the pending events).  With one engine busy_ms == sum_ms. */
int sprl_profile_busy(int kind, double* busy_ms, double* sum_ms);
/* profile = 2 (one HIP-event pair per convolution launch): time in ms and number of the trunk-convolution launches by kind -
 * [0] plain, [1] with residual input, [2] with the stem in its prologue, [3] with the head convolutions / FC layers behind it.
 * Lets a report separate the bare convolution from the launches that carry the stem / the tail (boards up to 8x8). */
int sprl_engine_conv_kinds(sprl_engine* e, double* ms4, int64_t* launches4);
void sprl_profile_busy_reset(void);

/* Expanded training samples exactly as the reference worker emits them (selfplay/GridWorker.hpp:146-196):
 * states float32[N][2H+1][R][C], distributions float32[N][A], outcomes float32[N], N = plies * nsym.
 * Buffers are caller-provided (sizes from sprl_records_num_samples). */
int64_t sprl_records_num_samples(const sprl_records* r);
int sprl_records_expand(const sprl_records* r, float* states, float* distributions, float* outcomes);
/* int8 boards + mover per sample instead of planes (for comparisons) */
int sprl_records_expand_boards(const sprl_records* r, int8_t* boards, int8_t* players);

/* utils/npy.hpp:430-476 + GridWorker.hpp:173-196: writes <prefix>_states.npy, _distributions.npy,
 * _outcomes.npy (byte-identical headers; temp file + rename, outcomes last). */
int sprl_write_npy(const char* path_prefix, const sprl_records* r);

/* A view of games [first_game, first_game + num_games) of `r` (the samples are not copied; valid while `r` is): one GPU run
 * covers several reference tasks, each task's games go to its own directory (sprl_worker --cover).  The view owns only its
 * rebased ply offsets: release it with sprl_records_free, which leaves `r` untouched. */
int sprl_records_slice(const sprl_records* r, int32_t first_game, int32_t num_games, sprl_records* out);

/* Compact record file (NOT a reference format): "SPRLv2\1\0", int64 size, then the packed wire format
 * described at sprl_engine_pack_records - one entry per ply, symmetries / history / plane encoding applied at load
 * (sprl_amd/records_v2.py).  Temp file + rename. */
int sprl_write_v2(const char* path, const sprl_records* r);

const char* sprl_last_error(void);
/* 1 when the library was built for gfx950 and a usable device is present */
int sprl_device_available(void);

#ifdef __cplusplus
}
#endif
#endif
