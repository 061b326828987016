/* TEST INFRASTRUCTURE — CPU restatement ("oracle") of the reference self-play path.
 *
 * Plain C restatement of willwin4sure/sprl's worker hot path
 *   runWorker -> runIteration -> selfPlay -> UCTTree{searchAndGetLeaves, evaluateAndBackpropLeaves,
 *   advanceDecision} -> GameNode move-gen + INetwork::evaluate + ISymmetrizer -> .npy records
 * (cpp/src/selfplay/GridWorker.hpp:84-198, selfplay/SelfPlay.hpp:51-248, uct/UCTTree.hpp, uct/UCTNode.hpp,
 *  games/{Othello,ConnectFour}Node.cpp, symmetry/..., networks/{Random,OthelloHeuristic,GridNetwork}...,
 *  utils/random.{hpp,cpp}, utils/npy.hpp).
 *
 * It is the checker, never the product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  Parity status: PINNED — bit-exact against the reference itself
 * (oracle/_ref, compiled from /root/reference in place) on RNG streams, move generation, symmetries,
 * search traces, whole self-play games and .npy byte streams; the committed fixtures under
 * tests/golden/ were produced by tests/golden/gen_golden.py from that build.
 *
 * math_mode: ORC_MATH_LIBM uses libm logf/powf/expf exactly like the reference (bit-exact vs
 * oracle/_ref); ORC_MATH_PORTABLE uses sprl_amd/csrc/sprl_math.h, the deterministic routines the
 * gfx950 kernels use (bit-exact vs the device engine; <= 1 float ulp from libm).
 */
#ifndef SPRL_ORACLE_H
#define SPRL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_A 384
#define ORC_MAX_CELLS 384

enum { ORC_GAME_OTHELLO = 0, ORC_GAME_C4 = 1, ORC_GAME_GO7 = 2, ORC_GAME_GO9 = 3, ORC_GAME_GO19 = 4 };
#define ORC_MAX_HIST 8
enum { ORC_EVAL_RANDOM = 0, ORC_EVAL_HEURISTIC = 1, ORC_EVAL_CALLBACK = 2 };
enum { ORC_MATH_LIBM = 0, ORC_MATH_PORTABLE = 1 };
enum { ORC_MASK_REFERENCE = 0, ORC_MASK_SYMMETRISED = 1 };

/* Network forward callback: planes float32[n][2H+1][R][C] -> logits float32[n][A], value float32[n]. */
typedef void (*orc_forward_fn)(void* user, int n, const float* planes, float* logits, float* values);

typedef struct {
    int32_t game;            /* ORC_GAME_* */
    int32_t num_traversals;  /* per move (lower bound, SelfPlay.hpp:100) */
    int32_t max_batch;       /* UCTTree.hpp:82 */
    int32_t max_queue;       /* UCTTree.hpp:108 */
    float dir_eps;
    float dir_alpha;
    float u_weight;          /* constants.hpp:6 U_WEIGHT = 1.1 */
    int32_t early_cutoff;    /* constants.hpp:8 */
    float early_exp;         /* constants.hpp:9 */
    float rest_exp;          /* constants.hpp:10 */
    int32_t use_sym;         /* symmetrizer != nullptr */
    int32_t add_noise;
    int32_t eval_kind;       /* ORC_EVAL_* */
    int32_t math_mode;       /* ORC_MATH_* */
    int32_t mask_frame;      /* ORC_MASK_* (Q1; reference = original-frame mask on symmetrised policy) */
    int32_t init_q;          /* 0 = InitQ::PARENT (workers), 1 = InitQ::ZERO (UCTNode.hpp:24-28,267-273) */
    float resign_threshold;  /* NOT in the reference, 0 = off: after a search, the side to move resigns when the mean
                                backed-up value of its decision node, sum W / sum N over the edges, is below -threshold */
    int32_t resign_min_ply;  /* no resignation before this ply */
    orc_forward_fn forward;  /* ORC_EVAL_CALLBACK */
    void* forward_user;
} orc_config;

typedef struct {
    int64_t games, plies, traversals, expansions, nn_evals, terminal_hits, gray_hits, dup_hits;
    int64_t levels;          /* sum over traversals of active levels descended */
    int64_t nodes_created, max_live_nodes;
} orc_stats;

/* game geometry */
int orc_game_cells(int game);
int orc_game_actions(int game);
int orc_game_nsym(int game);
int orc_game_rows(int game);
int orc_game_cols(int game);
int orc_game_hist(int game);

/* RNG: PCG32 + libstdc++ distribution algorithms (utils/random.{hpp,cpp}) */
typedef struct { uint64_t state, inc; } orc_rng;
void orc_rng_seed(orc_rng* r, uint64_t seed, int stream);
uint32_t orc_rng_next(orc_rng* r);
int orc_uniform_int(orc_rng* r, int a, int b);
float orc_uniform_float(orc_rng* r);
void orc_dirichlet(orc_rng* r, float alpha, int k, float* out, int math_mode);
int orc_sample_cdf(orc_rng* r, const float* cdf, int n);

/* rules */
void orc_start(int game, int8_t* board, int* player, float* mask);
void orc_step(int game, const int8_t* board, int player, const float* mask, int action,
              int8_t* board_out, float* mask_out, int* terminal_out, int* winner_out);
int orc_playout(int game, uint64_t seed, int stream, int max_plies, int8_t* boards, int8_t* players,
                int16_t* actions, float* masks, int8_t* terminal, float* rewards);

/* symmetries */
void orc_symmetrize_board(int game, int sym, const int8_t* in, int8_t* out);
void orc_symmetrize_dist(int game, int sym, const float* in, float* out);
int orc_inverse_symmetry(int game, int sym);

/* evaluators on explicit batches (boards already symmetrised, masks in original frame) */
/* boards: [n][hist][cells] (plies t >= sizes[i] ignored; sizes may be NULL = all `hist` plies valid) */
void orc_evaluate(const orc_config* cfg, int n, const int8_t* boards, const int8_t* players, const int8_t* sizes,
                  const float* masks, float* policies, float* values);
void orc_encode_planes(int game, int n, const int8_t* boards, const int8_t* players, const int8_t* sizes,
                       float* planes);
void orc_decode_policy(int A, const float* logits, const float* mask, float* policy, int math_mode);

/* search trace: same contract as ref_*_search_trace in ref_harness.cpp */
int orc_search_trace(const orc_config* cfg, int moves, uint64_t seed, int stream,
                     float* stats, int32_t* trav, int16_t* chosen);

/* self-play: same contract as ref_*_selfplay in ref_harness.cpp */
/* boards: [cap][hist][cells] (history plies beyond sizes[i] are written as -2), sizes: [cap] or NULL */
int orc_selfplay(const orc_config* cfg, int num_games, uint64_t seed, int stream_base, int per_game_stream,
                 int cap, int8_t* boards, int8_t* players, int8_t* sizes, float* dists, float* outcomes,
                 int32_t* game_offsets, orc_stats* stats);

/* net-vs-net matches: Evaluate.cpp:88-154 + evaluate/play.hpp:24-70 + agents/UCTNetworkAgent.hpp:42-108.
 * cfg0 / cfg1 describe agent 0 / agent 1 (evaluator, symmetrizer, init_q, budgets); agent (t % 2) plays Player ZERO
 * in game t; both trees share one RNG stream (seed, stream_base + t).  actions: [num_games][max_plies]. */
int orc_match(const orc_config* cfg0, const orc_config* cfg1, int num_games, uint64_t seed, int stream_base,
              int8_t* winners, int16_t* actions, int32_t* nplies, int max_plies);

/* records: plane encoding (GridWorker.hpp:146-171) + .npy v1.0 writer (utils/npy.hpp:430-476) */
int orc_write_npy_f32(const char* path, const float* data, int ndim, const uint64_t* shape);
int orc_write_records(const orc_config* cfg, const char* path_prefix, int n, const int8_t* boards,
                      const int8_t* players, const int8_t* sizes, const float* dists, const float* outcomes);

#ifdef __cplusplus
}
#endif
#endif
