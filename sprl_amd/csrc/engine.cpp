// engine.cpp — host side of the self-play engine: the C ABI of include/sprl_amd.h.
//
// Mirrors the reference's worker-side objects behind a C boundary:
//   sprl_engine_create/run  <->  runIteration / selfPlay / UCTTree      (selfplay/SelfPlay.hpp, uct/UCTTree.hpp)
//   sprl_engine_set_model   <->  GridNetwork / RandomNetwork selection  (selfplay/GridWorker.hpp:123-131)
//   sprl_records_expand     <->  symmetrised record assembly            (SelfPlay.hpp:86-92,127-133,151-189)
//   sprl_write_npy          <->  plane encoding + npy writer            (GridWorker.hpp:146-196, utils/npy.hpp:430-476)
// The search itself runs in the gfx950 kernels (step_kernel.h); this file only owns buffers and the round loop.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/sprl_amd.h"
#include "backend.h"
#include "engine_types.h"
#include "games.h"
#include "games_wide.h"
#include "records_kernel.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

struct Geom {
    int rows, cols, cells, A, nsym, max_depth, default_max_plies, hist, planes, hist_cap, node_bytes, words;
};

template <class G>
Geom wide_geom() {
    return { G::ROWS, G::COLS, G::CELLS, G::A, G::NSYM, G::MAX_DEPTH, G::GAME_MAX_DEPTH + 2, G::HIST, G::PLANES,
             G::HIST_CAP, G::NODE_BYTES, G::WORDS };
}

Geom geom_of(int game) {
    if (game == SPRL_CONNECT_FOUR)
        return { ConnectFour::ROWS, ConnectFour::COLS, ConnectFour::CELLS, ConnectFour::A, ConnectFour::NSYM,
                 ConnectFour::MAX_DEPTH, 48, 1, 3, 1, SPRL_NODE_BYTES, 1 };
    if (game == SPRL_GO7)
        return { Go7::ROWS, Go7::COLS, Go7::CELLS, Go7::A, Go7::NSYM, Go7::MAX_DEPTH, Go7::GAME_MAX_DEPTH + 2, Go7::HIST,
                 Go7::PLANES, Go7::HIST_CAP, SPRL_NODE_BYTES, 1 };
    if (game == SPRL_GO9) return wide_geom<GoN<9>>();
    if (game == SPRL_GO19) return wide_geom<GoN<19>>();
    if (game == SPRL_GO7_WIDE) return wide_geom<GoN<7>>();
    return { Othello::ROWS, Othello::COLS, Othello::CELLS, Othello::A, Othello::NSYM, Othello::MAX_DEPTH, 128, 1, 3, 1,
             SPRL_NODE_BYTES, 1 };
}

bool is_go(int game) { return game == SPRL_GO7 || game == SPRL_GO9 || game == SPRL_GO19 || game == SPRL_GO7_WIDE; }
bool known_game(int game) { return game == SPRL_OTHELLO || game == SPRL_CONNECT_FOUR || is_go(game); }

int map_cell(int game, int sym, int cell) {
    return game == SPRL_CONNECT_FOUR                     ? ConnectFour::map_cell(sym, cell)
           : (game == SPRL_GO7 || game == SPRL_GO7_WIDE) ? Go7::map_cell(sym, cell)
           : game == SPRL_GO9                            ? GoN<9>::map_cell(sym, cell)
           : game == SPRL_GO19                           ? GoN<19>::map_cell(sym, cell)
                                                         : Othello::map_cell(sym, cell);
}
int map_action(int game, int sym, int a) {
    return game == SPRL_CONNECT_FOUR                     ? ConnectFour::map_action(sym, a)
           : (game == SPRL_GO7 || game == SPRL_GO7_WIDE) ? Go7::map_action(sym, a)
           : game == SPRL_GO9                            ? GoN<9>::map_action(sym, a)
           : game == SPRL_GO19                           ? GoN<19>::map_action(sym, a)
                                                         : Othello::map_action(sym, a);
}

// LibTorch-ROCm evaluator plugin (libsprl_amd_torch.so), resolved lazily so the core has no torch dependency
struct TorchPlugin {
    void* lib = nullptr;
    void* (*load)(const char*, int, char*, int) = nullptr;
    void* (*load_buffer)(const void*, long long, int, char*, int) = nullptr;
    int (*forward)(void*, const float*, int, int, int, int, float*, int, float*, char*, int) = nullptr;
    int (*forward_on)(void*, const float*, int, int, int, int, float*, int, float*, void*, char*, int) = nullptr;
    void (*release)(void*) = nullptr;
    int (*is_native)(void*) = nullptr;
    int (*path_info)(void*, char*, int) = nullptr;
    void (*profile_read_kinds)(void*, double*, int64_t*) = nullptr;
    int (*forward_dev)(void*, const float*, const unsigned*, int, int, int, int, float*, int, float*, void*, char*, int) = nullptr;
    void (*profile_enable)(void*, int) = nullptr;
    void (*profile_read)(void*, double*, int64_t*, int64_t*) = nullptr;
};

struct RecordsOwner {
    std::vector<int32_t> ply_offset;
    std::vector<int8_t> boards, movers, winners;
    std::vector<float> pdfs;
};

}  // namespace

struct sprl_engine {
    sprl_config cfg;
    Geom g;
    EngineParams P;          // device pointers + parameters, passed by value to the kernel
    int eval_kind = SPRL_EVAL_RANDOM;
    sprl_forward_fn forward_cb = nullptr;
    void* forward_user = nullptr;
    TorchPlugin torch;
    void* torch_model = nullptr;
    int num_games = 0;
    bool running = false;
    size_t hbm_bytes = 0;
    std::vector<void*> allocs;
    float* nn_logits = nullptr;
    float* nn_value = nullptr;
    // accounting
    int64_t rounds = 0, launches = 0, nn_batches = 0, nn_rows = 0;
    int nn_bucket = 1024;
    bool nn_bucket_set = false;
    sprl_stats carried{};       // slot counters of finished runs (the slots' own counters restart with every run)
    int64_t games_begun = 0;    // games of earlier runs on this engine: a later run continues with fresh RNG streams
    void* stream = nullptr;     // private non-blocking stream (cfg.own_stream) or null = the null stream
    bool dev_batch = false;     // the evaluator takes the batch size from device memory: rounds are enqueued without a host sync
    unsigned long long last_leaf_rows = 0;
    double seconds = 0.0, kernel_ms = 0.0, nn_ms = 0.0;
    int64_t rec_total = -1;     // total plies of the finished run once the device-side offsets scan has run, else -1
    std::vector<void*> marks;  // k0,k1,(n0,n1) per round, resolved lazily
    std::vector<int> mark_kind;
    void* chain = nullptr;     // be::chain_new(): this engine's tree-kernel intervals on the process-wide busy clock
    // lab switches of the engine itself, read ONCE in sprl_engine_create (never in the round loop) and reported by
    // sprl_engine_evaluator_info: SPRL_SYNC_ROUNDS (host reads the leaf count every round), SPRL_GO_LEGAL (legal-move algorithm)
    bool lab_sync_rounds = false;
    std::string lab;           // the names that were set, "" = defaults
    // lab, SPRL_TREE_STREAM=1|2: the tree / scan / gather launches of a round go to a second stream of the
    // engine (2: at the device's highest priority), the forward stays on `stream`; ev_tree / ev_fwd order the two every round
    void* tree_stream = nullptr;
    void *ev_tree = nullptr, *ev_fwd = nullptr;
};

namespace {

void* dev_alloc(sprl_engine* e, size_t bytes) {
    void* p = be::dmalloc(bytes ? bytes : 16);
    if (p) {
        e->allocs.push_back(p);
        e->hbm_bytes += bytes;
    }
    return p;
}

// trunk-convolution busy clock of the evaluator plugin (one per process)
double (*g_conv_busy)(double*) = nullptr;
void (*g_conv_busy_reset)() = nullptr;

void resolve_marks(sprl_engine* e) {
    for (size_t i = 0; i + 1 < e->marks.size(); i += 2) {
        if (e->mark_kind[i / 2] == 0) {          // tree kernel: also on the process-wide busy clock (several engines overlap)
            if (!e->chain) e->chain = be::chain_new();
            e->kernel_ms += be::resolve_logged(e->chain, e->marks[i], e->marks[i + 1]);
            continue;
        }
        e->nn_ms += be::elapsed_ms(e->marks[i], e->marks[i + 1]);
        be::mark_free(e->marks[i]);
        be::mark_free(e->marks[i + 1]);
    }
    e->marks.clear();
    e->mark_kind.clear();
}

int load_torch_plugin(sprl_engine* e) {
    if (e->torch.lib) return 0;
    Dl_info info;
    std::string dir = ".";
    if (dladdr((void*)&sprl_config_default, &info) && info.dli_fname) {
        std::string f(info.dli_fname);
        size_t k = f.rfind('/');
        if (k != std::string::npos) dir = f.substr(0, k);
    }
    std::string path = dir + "/libsprl_amd_torch.so";
    void* lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!lib) return fail(SPRL_E_MODEL, std::string("cannot load LibTorch evaluator plugin: ") + dlerror());
    e->torch.lib = lib;
    e->torch.load = (void* (*)(const char*, int, char*, int))dlsym(lib, "sprl_torch_load");
    e->torch.load_buffer = (void* (*)(const void*, long long, int, char*, int))dlsym(lib, "sprl_torch_load_buffer");
    e->torch.forward = (int (*)(void*, const float*, int, int, int, int, float*, int, float*, char*, int))dlsym(lib, "sprl_torch_forward");
    e->torch.forward_on = (int (*)(void*, const float*, int, int, int, int, float*, int, float*, void*, char*, int))dlsym(lib, "sprl_torch_forward_on");
    e->torch.release = (void (*)(void*))dlsym(lib, "sprl_torch_free");
    e->torch.is_native = (int (*)(void*))dlsym(lib, "sprl_torch_is_native");
    e->torch.path_info = (int (*)(void*, char*, int))dlsym(lib, "sprl_torch_path_info");
    e->torch.profile_read_kinds = (void (*)(void*, double*, int64_t*))dlsym(lib, "sprl_torch_profile_read_kinds");
    e->torch.forward_dev = (int (*)(void*, const float*, const unsigned*, int, int, int, int, float*, int, float*, void*, char*, int))dlsym(
        lib, "sprl_torch_forward_dev");
    e->torch.profile_enable = (void (*)(void*, int))dlsym(lib, "sprl_torch_profile_enable");
    e->torch.profile_read = (void (*)(void*, double*, int64_t*, int64_t*))dlsym(lib, "sprl_torch_profile_read");
    g_conv_busy = (double (*)(double*))dlsym(lib, "sprl_torch_profile_busy");
    g_conv_busy_reset = (void (*)())dlsym(lib, "sprl_torch_profile_busy_reset");
    if (!e->torch.load || !e->torch.forward || !e->torch.release)
        return fail(SPRL_E_MODEL, "LibTorch evaluator plugin lacks required symbols");
    return 0;
}

// Load a traced model through the LibTorch plugin and touch every batch shape the round loop can produce (multiples
// of the bucket) once, so that the convolution library's per-shape solver selection happens here and not inside a run.
int load_network(sprl_engine* e, const char* model, void** out, const void* bytes = nullptr, int64_t nbytes = 0) {
    int rc = load_torch_plugin(e);
    if (rc) return rc;
    char err[512] = { 0 };
    if (bytes && !e->torch.load_buffer) return fail(SPRL_E_MODEL, "LibTorch evaluator plugin lacks sprl_torch_load_buffer");
    void* m = bytes ? e->torch.load_buffer(bytes, (long long)nbytes, e->cfg.device, err, (int)sizeof(err))
                    : e->torch.load(model, e->cfg.device, err, (int)sizeof(err));
    if (!m) return fail(SPRL_E_MODEL, std::string("cannot load TorchScript model '") + (model ? model : "<memory>") + "': " + err);
    // The hand-written CNN (plugin kind 2) has no per-shape solver selection, so its batches are padded to 64 rows only;
    // library convolutions pick a solver per shape, so those batches come in 1024-row buckets, each touched once here.
    const int kind = e->torch.is_native ? e->torch.is_native(m) : 0;
    const int bucket = kind == 2 ? 64 : 1024;
    if (bucket > e->nn_bucket || !e->nn_bucket_set) e->nn_bucket = bucket;
    e->nn_bucket_set = true;
    void* const saved_stream = be::current_stream();
    be::set_stream(nullptr);                     // the plain forward works on the null stream
    const int max_rows = e->P.num_slots * e->P.max_queue;
    const int step_rows = kind == 2 ? (max_rows > 4096 ? max_rows / 4 : max_rows) : e->nn_bucket;
    int last = 0;
    for (int rows = step_rows;; rows += step_rows) {
        int b = rows < max_rows ? rows : max_rows;
        if (b == last) break;
        last = b;
        if (e->torch.forward(m, e->P.nn_dense, b, e->g.planes, e->g.rows, e->g.cols, e->nn_logits, e->g.A, e->nn_value, err,
                             (int)sizeof(err)) != 0) {
            e->torch.release(m);
            be::set_stream(saved_stream);
            return fail(SPRL_E_MODEL, std::string("network warm-up forward failed: ") + err);
        }
        if (b == max_rows) break;
    }
    be::sync();
    be::set_stream(saved_stream);
    *out = m;
    return 0;
}

// per-slot search counters of the current run, added into `out`
void add_slot_stats(const std::vector<GameCtl>& ctl, sprl_stats* out) {
    for (const GameCtl& c : ctl) {
        out->games += (int64_t)c.stats.games;
        out->plies += (int64_t)c.stats.plies;
        out->traversals += (int64_t)c.stats.traversals;
        out->levels += (int64_t)c.stats.levels;
        out->expansions += (int64_t)c.stats.expansions;
        out->nn_evals += (int64_t)c.stats.nn_evals;
        out->terminal_hits += (int64_t)c.stats.terminal_hits;
        out->gray_hits += (int64_t)c.stats.gray_hits;
        out->dup_hits += (int64_t)c.stats.dup_hits;
        out->nodes_created += (int64_t)c.stats.nodes_created;
        out->compactions += (int64_t)c.stats.compactions;
        out->nodes_recycled += (int64_t)c.stats.nodes_recycled;
        out->cyc_total += (int64_t)c.stats.cyc_total;
        out->cyc_finish += (int64_t)c.stats.cyc_finish;
        out->cyc_move += (int64_t)c.stats.cyc_move;
        out->cyc_select += (int64_t)c.stats.cyc_select;
        out->cyc_create += (int64_t)c.stats.cyc_create;
        out->cyc_backup += (int64_t)c.stats.cyc_backup;
        out->cyc_leafio += (int64_t)c.stats.cyc_leafio;
        out->cyc_noise += (int64_t)c.stats.cyc_noise;
        out->cyc_lvl_wait += (int64_t)c.stats.cyc_lvl_wait;
        out->cyc_lvl_pick += (int64_t)c.stats.cyc_lvl_pick;
        out->cyc_lvl_desc += (int64_t)c.stats.cyc_lvl_desc;
        if ((int64_t)c.stats.cyc_max > out->cyc_max_slot_launch) out->cyc_max_slot_launch = (int64_t)c.stats.cyc_max;
        if ((int64_t)c.stats.max_alloc > out->max_nodes_in_arena) out->max_nodes_in_arena = (int64_t)c.stats.max_alloc;
    }
}

int check_device_error(sprl_engine* e, const Counters& c) {
    if (c.error == ERR_NONE) return 0;
    char buf[256];
    const char* what = c.error == ERR_ARENA_FULL ? "node arena full even after compaction (raise node_cap)"
                       : c.error == ERR_NO_SPARE ? "no spare arena free for compaction (raise spare_arenas)"
                       : c.error == ERR_MAX_PLIES ? "game longer than max_plies"
                                                  : "search line deeper than the path buffer";
    snprintf(buf, sizeof(buf), "game %u: %s", c.error_game, what);
    e->running = false;
    return fail((c.error == ERR_ARENA_FULL || c.error == ERR_NO_SPARE) ? SPRL_E_NODEPOOL : SPRL_E_LIMIT, buf);
}

}  // namespace

extern "C" {

const char* sprl_last_error(void) { return g_last_error.c_str(); }

int sprl_device_available(void) {
    std::string why;
    return be::available(&why) ? 1 : 0;
}

int sprl_config_default(int32_t game, sprl_config* cfg) {
    if (!cfg || !known_game(game)) return fail(SPRL_E_CONFIG, "unknown game");
    memset(cfg, 0, sizeof(*cfg));
    cfg->game = game;
    cfg->device = 0;
    cfg->concurrent_games = 4096;
    cfg->num_traversals = 800;
    cfg->max_batch = is_go(game) ? 16 : 8;           // OTHWorker.cpp:24, C4Worker.cpp:23, GoWorker.cpp:23
    cfg->max_queue = is_go(game) ? 8 : 4;             // OTHWorker.cpp:25, C4Worker.cpp:24, GoWorker.cpp:24
    cfg->dir_eps = 0.25f;                                 // OTHWorker.cpp:27, C4Worker.cpp:26, GoWorker.cpp:26
    cfg->dir_alpha = game == SPRL_OTHELLO ? 0.3f : (is_go(game) ? 0.2f : 0.5f);   // :28 / :27 / GoWorker.cpp:27
    cfg->u_weight = 1.1f;                                 // constants.hpp:6
    cfg->early_cutoff = 15;                               // constants.hpp:8
    cfg->early_exp = 0.98f;                               // constants.hpp:9
    cfg->rest_exp = 10.0f;                                // constants.hpp:10
    cfg->use_symmetry = 1;
    cfg->add_noise = 1;
    cfg->mask_frame = SPRL_MASK_REFERENCE;
    cfg->seed = 1;
    cfg->stream_base = 1;
    return 0;
}

int sprl_engine_create(const sprl_config* cfg, sprl_engine** out) {
    if (!cfg || !out) return fail(SPRL_E_CONFIG, "null argument");
    *out = nullptr;
    if (!known_game(cfg->game)) return fail(SPRL_E_CONFIG, "unknown game");
    if (cfg->concurrent_games < 1) return fail(SPRL_E_CONFIG, "concurrent_games must be >= 1");
    if (cfg->num_traversals < 1) return fail(SPRL_E_CONFIG, "num_traversals must be >= 1");
    if (cfg->max_batch < 1) return fail(SPRL_E_CONFIG, "max_batch must be >= 1");
    if (cfg->max_queue < 1 || cfg->max_queue > SPRL_MAXQ) return fail(SPRL_E_CONFIG, "max_queue must be in [1, 8]");
    if (cfg->num_traversals <= cfg->max_queue)
        return fail(SPRL_E_CONFIG, "num_traversals must exceed max_queue: a fresh root is queued max_queue times (SURVEY Q7), so "
                                   "the first move would be sampled from an all-zero visit count (NaN pdf in the reference)");
    if (cfg->stream_base < 1) return fail(SPRL_E_CONFIG, "stream_base must be >= 1 (stream 0 means 'pick one' in the reference)");
    const bool single_strip = cfg->game == SPRL_OTHELLO || cfg->game == SPRL_CONNECT_FOUR || cfg->game == SPRL_GO7;
    if (cfg->node_cap < 0 || cfg->node_cap > (single_strip ? 0xFFFFFE : 65535))
        return fail(SPRL_E_CONFIG, single_strip ? "node_cap must be < 2^24" : "node_cap must be <= 65535 for boards wider than 8x8");
    if (!(cfg->dir_alpha > 0.0f)) return fail(SPRL_E_CONFIG, "dir_alpha must be > 0");
    if (cfg->resign_threshold < 0.0f || cfg->resign_threshold >= 1.0f) return fail(SPRL_E_CONFIG, "resign_threshold must be in [0, 1)");
    std::string err;
    if (be::init(cfg->device, &err) != 0) return fail(SPRL_E_DEVICE, "no usable gfx950 device: " + err);

    sprl_engine* e = new sprl_engine();
    e->cfg = *cfg;
    if (cfg->own_stream) {
        e->stream = be::stream_create();
        if (!e->stream) {
            delete e;
            return fail(SPRL_E_DEVICE, std::string("cannot create a HIP stream (") + be::last_error() + ")");
        }
    }
    be::bind(e->cfg.device, e->stream);
    e->g = geom_of(cfg->game);
    EngineParams& P = e->P;
    memset(&P, 0, sizeof(P));
    P.num_traversals = cfg->num_traversals;
    P.max_batch = cfg->max_batch;
    P.max_queue = cfg->max_queue;
    P.dir_eps = cfg->dir_eps;
    P.dir_alpha = cfg->dir_alpha;
    P.u_weight = cfg->u_weight;
    P.early_cutoff = cfg->early_cutoff;
    P.early_exp = cfg->early_exp;
    P.rest_exp = cfg->rest_exp;
    if (const char* f = getenv("SPRL_GO_LEGAL")) {         // test hook: both algorithms are checked on every board size
        P.go_legal_form = strcmp(f, "label") == 0 ? 1 : (strcmp(f, "flood") == 0 ? 2 : 0);
        e->lab += std::string(e->lab.empty() ? "" : " ") + "SPRL_GO_LEGAL=" + f;
    }
    if (getenv("SPRL_SYNC_ROUNDS")) {
        e->lab_sync_rounds = true;
        e->lab += std::string(e->lab.empty() ? "" : " ") + "SPRL_SYNC_ROUNDS";
    }
    if (const char* f = getenv("SPRL_TREE_STREAM")) {
        if (e->stream && (f[0] == '1' || f[0] == '2')) {
            e->tree_stream = be::stream_create_priority(f[0] == '2');
            e->ev_tree = be::event_new();
            e->ev_fwd = be::event_new();
            if (!e->tree_stream || !e->ev_tree || !e->ev_fwd) {
                std::string why = be::last_error();
                sprl_engine_destroy(e);
                return fail(SPRL_E_DEVICE, "SPRL_TREE_STREAM: cannot create the second stream (" + why + ")");
            }
            e->lab += std::string(e->lab.empty() ? "" : " ") + "SPRL_TREE_STREAM=" + f[0];
        }
    }
    P.resign_threshold = cfg->resign_threshold;
    P.resign_min_ply = cfg->resign_min_ply;
    P.use_sym = cfg->use_symmetry ? 1 : 0;
    P.add_noise = cfg->add_noise ? 1 : 0;
    P.eval_kind = EVAL_RANDOM;
    P.mask_frame = cfg->mask_frame;
    P.stream_base = cfg->stream_base;
    P.seed = cfg->seed;
    P.num_slots = cfg->concurrent_games;
    P.num_spare = cfg->spare_arenas > 0 ? cfg->spare_arenas : (cfg->concurrent_games / 64 > 8 ? cfg->concurrent_games / 64 : 8);
    // Node recycling (single-strip kernel): an arena holds the live subtree + the garbage not yet reused, so its size follows
    // the per-move budget, not the game length.  High-water marks measured over whole games: Othello @800 1 291 nodes,
    // Connect Four @512 673, Go 7x7 @400 487, maximum over 4096 Othello games @800 2 261 - 4 x traversals
    // + 1024 leaves a 2-5x margin at the steady-state budgets.  The margin a tree needs does not grow with the budget (the
    // high-water mark stays near 1.6 x traversals), so above 2048 traversals the default is 2 x + 4096 + 1024: the reference's
    // iteration-0 budgets (131 072 / 262 144 traversals per move, OTHWorker.cpp:17, GoWorker.cpp:17) then take 0.26 / 0.5 GiB
    // per game instead of 0.5 / 1 GiB; compaction into a spare arena remains the fallback.  Boards wider than 8x8 (multi-strip kernel): the same, with 16-bit child indices.
    P.recycle = (cfg->max_batch + 2 <= SPRL_FCACHE && !cfg->no_recycle) ? 1 : 0;
    long cap = cfg->node_cap > 0 ? cfg->node_cap
               : P.recycle     ? std::min((long)cfg->num_traversals * 4, (long)cfg->num_traversals * 2 + 4096) + 1024
                               : (long)cfg->num_traversals * 52 + 1024;
    if (!single_strip && cap > 65535) cap = 65535;
    if (cap > 0xFFFFFE) cap = 0xFFFFFE;
    if (cap < cfg->max_batch + 8) cap = cfg->max_batch + 8;
    P.node_cap = (int)cap;
    if (cfg->alloc_base != 0) {                         // tests: games start allocating here (crosses the 16-bit boundary early)
        if (!single_strip || cfg->alloc_base < 0 || (long)cfg->alloc_base + cfg->max_batch + 8 >= cap) {
            sprl_engine_destroy(e);
            return fail(SPRL_E_CONFIG, "alloc_base needs a single-strip game and alloc_base + max_batch + 8 < node_cap");
        }
        P.alloc_base = (uint32_t)cfg->alloc_base;
    }
    P.wide_idx = cap > 65535 ? 1 : 0;                 // child indices: u16 row, + a u8 row above 65535 nodes
    P.none_idx = P.wide_idx ? SPRL_NONE24 : SPRL_NONE16;
    P.max_plies = cfg->max_plies > 0 ? cfg->max_plies : e->g.default_max_plies;
    P.max_depth = e->g.max_depth;
    P.planes = e->g.planes;
    P.rounds = 1;

    const size_t arenas = (size_t)(P.num_slots + P.num_spare);
    const size_t nq = (size_t)P.num_slots * (size_t)P.max_queue;   // dense network batch: slot-major, queue-minor
    const size_t npaths = (size_t)P.num_slots * SPRL_MAXQ;
    bool ok = true;
    ok = ok && (P.arenas = (uint8_t*)dev_alloc(e, arenas * (size_t)P.node_cap * (size_t)e->g.node_bytes));
    ok = ok && (P.arena_used = (uint32_t*)dev_alloc(e, arenas * sizeof(uint32_t)));
    if (P.recycle) ok = ok && (P.reclaim = (uint32_t*)dev_alloc(e, (size_t)P.num_slots * (size_t)P.node_cap * sizeof(uint32_t)));
    ok = ok && (P.ctl = (GameCtl*)dev_alloc(e, (size_t)P.num_slots * sizeof(GameCtl)));
    ok = ok && (P.paths = (uint32_t*)dev_alloc(e, npaths * (size_t)P.max_depth * sizeof(uint32_t)));
    ok = ok && (P.nn_in = (float*)dev_alloc(e, nq * (size_t)e->g.planes * (size_t)e->g.cells * sizeof(float)));
    ok = ok && (P.nn_dense = (float*)dev_alloc(e, nq * (size_t)e->g.planes * (size_t)e->g.cells * sizeof(float)));
    ok = ok && (P.leaf_count = (uint32_t*)dev_alloc(e, (size_t)P.num_slots * sizeof(uint32_t)));
    ok = ok && (P.leaf_offset = (uint32_t*)dev_alloc(e, (size_t)P.num_slots * sizeof(uint32_t)));
    ok = ok && (e->nn_logits = (float*)dev_alloc(e, nq * (size_t)e->g.A * sizeof(float)));
    ok = ok && (e->nn_value = (float*)dev_alloc(e, nq * sizeof(float)));
    ok = ok && (P.counters = (Counters*)dev_alloc(e, sizeof(Counters)));
    ok = ok && (P.hist_boards = (uint64_t*)dev_alloc(e, (size_t)P.num_slots * (size_t)e->g.hist_cap * 2 * (size_t)e->g.words * sizeof(uint64_t)));
    if (!ok) {
        std::string m = std::string("device allocation failed (") + be::last_error() + ")";
        sprl_engine_destroy(e);
        return fail(SPRL_E_NOMEM, m);
    }
    P.nn_logits = e->nn_logits;
    P.nn_value = e->nn_value;
    be::dmemset(P.nn_in, 0, nq * (size_t)e->g.planes * (size_t)e->g.cells * sizeof(float));
    be::dmemset(P.nn_dense, 0, nq * (size_t)e->g.planes * (size_t)e->g.cells * sizeof(float));
    be::dmemset(P.leaf_count, 0, (size_t)P.num_slots * sizeof(uint32_t));
    be::dmemset(P.leaf_offset, 0, (size_t)P.num_slots * sizeof(uint32_t));
    be::dmemset(e->nn_logits, 0, nq * (size_t)e->g.A * sizeof(float));
    be::dmemset(e->nn_value, 0, nq * sizeof(float));
    be::dmemset(P.counters, 0, sizeof(Counters));
    be::sync();
    *out = e;
    return 0;
}

void sprl_engine_destroy(sprl_engine* e) {
    if (!e) return;
    be::bind(e->cfg.device, e->stream);
    be::sync();
    resolve_marks(e);
    be::chain_free(e->chain);
    if (e->torch_model && e->torch.release) e->torch.release(e->torch_model);
    for (void* p : e->allocs) be::dfree(p);
    if (e->tree_stream) be::stream_destroy(e->tree_stream);
    be::event_free(e->ev_tree);
    be::event_free(e->ev_fwd);
    if (e->stream) be::stream_destroy(e->stream);
    be::set_stream(nullptr);
    delete e;
}

static int set_model_common(sprl_engine* e, const char* model, const void* bytes, int64_t nbytes);

int sprl_engine_set_model(sprl_engine* e, const char* model) {
    if (!e || !model) return fail(SPRL_E_CONFIG, "null argument");
    return set_model_common(e, model, nullptr, 0);
}

int sprl_engine_set_model_buffer(sprl_engine* e, const void* torchscript_bytes, int64_t nbytes) {
    if (!e || !torchscript_bytes || nbytes <= 0) return fail(SPRL_E_CONFIG, "null argument");
    return set_model_common(e, nullptr, torchscript_bytes, nbytes);
}

static int set_model_common(sprl_engine* e, const char* model, const void* bytes, int64_t nbytes) {
    be::bind(e->cfg.device, e->stream);
    if (e->running) return fail(SPRL_E_STATE, "cannot change the evaluator while a run is in progress");
    e->dev_batch = false;
    if (model && strcmp(model, "random") == 0) {          // GridWorker.hpp:36-38,125-127
        e->eval_kind = SPRL_EVAL_RANDOM;
        return 0;
    }
    if (model && strcmp(model, "heuristic") == 0) {
        if (e->cfg.game != SPRL_OTHELLO) return fail(SPRL_E_CONFIG, "the heuristic evaluator exists for Othello only");
        e->eval_kind = SPRL_EVAL_HEURISTIC;
        return 0;
    }
    void* m = nullptr;
    int rc = load_network(e, model, &m, bytes, nbytes);
    if (rc) return rc;
    if (e->torch_model) e->torch.release(e->torch_model);
    e->torch_model = m;
    {   // can this model run with the leaf count left on the device?  (probe with the current count, normally 0)
        char perr[256] = { 0 };
        e->dev_batch = e->torch.forward_dev && !e->lab_sync_rounds &&
                       e->torch.forward_dev(m, e->P.nn_dense, &e->P.counters->leaf_total, e->P.num_slots * e->P.max_queue, e->g.planes,
                                            e->g.rows, e->g.cols, e->nn_logits, e->g.A, e->nn_value, e->stream, perr, (int)sizeof(perr)) == 0;
        be::sync();
    }
    if (e->stream && !e->dev_batch && !e->torch.forward_on) {
        e->torch.release(m);
        e->torch_model = nullptr;
        return fail(SPRL_E_CONFIG, "own_stream needs an evaluator that runs on the engine's stream: this network plugin has no "
                                   "sprl_torch_forward_on");
    }
    if (e->cfg.profile && e->torch.profile_enable) e->torch.profile_enable(m, e->cfg.profile == 2 ? 2 : 1);
    e->forward_cb = nullptr;
    e->eval_kind = SPRL_EVAL_NETWORK;
    return 0;
}

int sprl_engine_evaluator_info(sprl_engine* e, char* buf, int32_t len) {
    if (!e || !buf || len < 1) return fail(SPRL_E_CONFIG, "null argument");
    const char* what = e->eval_kind == SPRL_EVAL_RANDOM      ? "random (in-kernel)"
                       : e->eval_kind == SPRL_EVAL_HEURISTIC ? "heuristic (in-kernel)"
                       : e->forward_cb                      ? "forward callback"
                       : (e->torch.is_native && e->torch_model && e->torch.is_native(e->torch_model) == 2 && e->g.rows <= 8 && e->g.cols <= 8)
                           ? "hand-written gfx950 CNN: MFMA stem + Winograd F(4x4,3x3) fp32-MFMA trunk with fused BN/residual/ReLU + fused heads/FC tail"
                       : (e->torch.is_native && e->torch_model && e->torch.is_native(e->torch_model) == 2)
                           ? (e->dev_batch ? "hand-written gfx950 CNN for any board: NCHW MFMA stem + Winograd fp32-MFMA trunk + fused heads/FC tail, "
                                             "batch size read on the device (no host round trip per search round)"
                                           : "hand-written gfx950 CNN for any board: NCHW MFMA stem + Winograd fp32-MFMA trunk + fused heads/FC tail, "
                                             "batch size on the host")
                       : (e->torch.is_native && e->torch_model && e->torch.is_native(e->torch_model))
                           ? "LibTorch-ROCm: MIOpen convolutions + fused bias/BN/ReLU epilogue kernel"
                           : "LibTorch-ROCm: TorchScript graph (bias hoisted for the JIT fuser)";
    // the resolved path of the plugin (kind, fused tail, lab switches set when the model was loaded) and the engine's own lab switches:
    // a timed run checks "lab=[]" in both (bench.py)
    char path[384] = { 0 };
    if (e->eval_kind == SPRL_EVAL_NETWORK && !e->forward_cb && e->torch_model && e->torch.path_info)
        e->torch.path_info(e->torch_model, path, (int)sizeof(path));
    snprintf(buf, (size_t)len, "%s%s%s%s; engine lab=[%s]", what, path[0] ? " {" : "", path, path[0] ? "}" : "", e->lab.c_str());
    return 0;
}

int sprl_engine_set_forward(sprl_engine* e, sprl_forward_fn fn, void* user) {
    if (!e || !fn) return fail(SPRL_E_CONFIG, "null argument");
    if (e->stream) return fail(SPRL_E_CONFIG, "a forward hook works on the null stream: not available with own_stream");
    if (e->running) return fail(SPRL_E_STATE, "cannot change the evaluator while a run is in progress");
    e->forward_cb = fn;
    e->forward_user = user;
    e->eval_kind = SPRL_EVAL_NETWORK;
    return 0;
}

int sprl_engine_begin(sprl_engine* e, int32_t num_games) {
    if (!e) return fail(SPRL_E_CONFIG, "null engine");
    be::bind(e->cfg.device, e->stream);
    if (num_games < 1) return fail(SPRL_E_CONFIG, "num_games must be >= 1");
    EngineParams& P = e->P;
    // (re)allocate record buffers for this run
    const size_t np = (size_t)num_games * (size_t)P.max_plies;
    if (num_games > e->num_games || !P.rec_boards) {
        // earlier record buffers stay in `allocs` and are released at destroy; runs normally reuse the size
        bool ok = true;
        ok = ok && (P.rec_boards = (uint64_t*)dev_alloc(e, np * 2 * (size_t)e->g.words * sizeof(uint64_t)));
        ok = ok && (P.rec_movers = (uint8_t*)dev_alloc(e, np));
        ok = ok && (P.rec_pdf = (float*)dev_alloc(e, np * (size_t)e->g.A * sizeof(float)));
        ok = ok && (P.rec_nplies = (int32_t*)dev_alloc(e, (size_t)num_games * sizeof(int32_t)));
        ok = ok && (P.rec_offsets = (int32_t*)dev_alloc(e, ((size_t)num_games + 1) * sizeof(int32_t)));
        ok = ok && (P.rec_winner = (int8_t*)dev_alloc(e, (size_t)num_games));
        ok = ok && (P.rec_evals = (uint32_t*)dev_alloc(e, (size_t)num_games * sizeof(uint32_t)));
        if (!ok) return fail(SPRL_E_NOMEM, std::string("record allocation failed (") + be::last_error() + ")");
    }
    e->rec_total = -1;
    if (e->games_begun > 0) {                    // keep the search counters of the run that is being replaced
        std::vector<GameCtl> old((size_t)P.num_slots);
        if (be::sync() != 0 || be::d2h(old.data(), P.ctl, old.size() * sizeof(GameCtl)) != 0) return fail(SPRL_E_DEVICE, be::last_error());
        add_slot_stats(old, &e->carried);
    }
    e->num_games = num_games;
    e->last_leaf_rows = 0;
    P.stream_base = e->cfg.stream_base + (int32_t)e->games_begun;   // run k's game g: stream_base + games of runs < k + g
    e->games_begun += num_games;
    P.num_games = num_games;
    P.eval_kind = e->eval_kind;
    std::vector<GameCtl> ctl((size_t)P.num_slots);
    memset(ctl.data(), 0, ctl.size() * sizeof(GameCtl));
    for (int s = 0; s < P.num_slots; ++s) {
        ctl[(size_t)s].status = ST_FRESH;
        ctl[(size_t)s].arena = (uint32_t)s;
    }
    std::vector<uint32_t> used((size_t)(P.num_slots + P.num_spare), 0u);
    for (int s = 0; s < P.num_slots; ++s) used[(size_t)s] = 1u;
    Counters c;
    memset(&c, 0, sizeof(c));
    int rc = 0;
    rc |= be::h2d(P.ctl, ctl.data(), ctl.size() * sizeof(GameCtl));
    rc |= be::h2d(P.arena_used, used.data(), used.size() * sizeof(uint32_t));
    rc |= be::h2d(P.counters, &c, sizeof(c));
    rc |= be::dmemset(P.leaf_count, 0, (size_t)P.num_slots * sizeof(uint32_t));
    rc |= be::dmemset(P.leaf_offset, 0, (size_t)P.num_slots * sizeof(uint32_t));
    rc |= be::dmemset(P.rec_nplies, 0, (size_t)num_games * sizeof(int32_t));
    rc |= be::dmemset(P.rec_evals, 0, (size_t)num_games * sizeof(uint32_t));
    rc |= be::dmemset(P.rec_pdf, 0, np * (size_t)e->g.A * sizeof(float));
    rc |= be::sync();
    if (rc) return fail(SPRL_E_DEVICE, be::last_error());
    e->running = true;
    return 0;
}

int sprl_engine_step(sprl_engine* e, int32_t rounds, int32_t* games_done, int32_t* active_slots) {
    if (!e) return fail(SPRL_E_CONFIG, "null engine");
    be::bind(e->cfg.device, e->stream);
    if (!e->running) return fail(SPRL_E_STATE, "sprl_engine_begin has not been called");
    if (rounds < 1) rounds = 1;
    EngineParams& P = e->P;
    auto t0 = std::chrono::steady_clock::now();
    const bool net = e->eval_kind == SPRL_EVAL_NETWORK;
    const int launches = net ? rounds : 1;
    P.rounds = net ? 1 : rounds;
    const int max_batch_rows = P.num_slots * P.max_queue;
    const int floats_per_leaf = e->g.planes * e->g.cells;
    Counters c;
    memset(&c, 0, sizeof(c));
    const bool split = net && e->tree_stream;             // lab: tree / scan / gather on the engine's second stream
    if (split) be::event_record(e->ev_fwd, e->stream);     // (everything issued so far on the engine's stream comes first)
    for (int r = 0; r < launches; ++r) {
        // network rounds: the leaf scan behind the step kernel moves active_slots to active_last and clears it (one launch less
        // per round); the chained rounds of the in-kernel evaluators have no scan
        if (!net) be::dmemset(&P.counters->active_slots, 0, sizeof(uint32_t));
        if (split) {
            be::stream_wait(e->tree_stream, e->ev_fwd);   // the answers of the previous round's forward
            be::set_stream(e->tree_stream);
        }
        void* k0 = e->cfg.profile ? be::mark() : nullptr;
        if (be::launch_step(e->cfg.game, P) != 0) return fail(SPRL_E_DEVICE, be::last_error());
        if (e->cfg.profile) {
            e->marks.push_back(k0);
            e->marks.push_back(be::mark());
            e->mark_kind.push_back(0);
        }
        e->launches++;
        if (net) {
            // dense batch: only the leaves that were really queued, slot-major (deterministic), in buckets of
            // `bucket` rows so the convolution library sees a handful of shapes
            if (be::launch_compact(P, floats_per_leaf) != 0) return fail(SPRL_E_DEVICE, be::last_error());
            if (split) {
                be::event_record(e->ev_tree, e->tree_stream);
                be::set_stream(e->stream);
                be::stream_wait(e->stream, e->ev_tree);    // the forward (and everything else on the engine's stream) behind the gather
            }
            if (e->dev_batch && !e->forward_cb) {
                // the evaluator reads the leaf count from the device: no host round trip between rounds; completion and
                // errors are looked at every 8 rounds (rounds after the last game ended find nothing to do)
                // (profile = 1 brackets the trunk convolutions inside the plugin; the whole-forward pair is recorded only in the
                // per-launch mode 2: every event record is a packet in the hardware queue, and a timed run carries four per round -
                // tree kernel and convolution bracket - instead of six)
                void* n0 = e->cfg.profile == 2 ? be::mark() : nullptr;
                char err[512] = { 0 };
                if (e->torch.forward_dev(e->torch_model, P.nn_dense, &P.counters->leaf_total, max_batch_rows, e->g.planes, e->g.rows,
                                         e->g.cols, e->nn_logits, e->g.A, e->nn_value, e->stream, err, (int)sizeof(err)) != 0) {
                    e->running = false;
                    return fail(SPRL_E_MODEL, std::string("network forward failed: ") + err);
                }
                if (n0) {
                    e->marks.push_back(n0);
                    e->marks.push_back(be::mark());
                    e->mark_kind.push_back(1);
                }
                e->nn_batches++;
                if (split) be::event_record(e->ev_fwd, e->stream);
                if ((r & 7) == 7 || r == launches - 1) {
                    if (be::sync() != 0 || be::d2h(&c, P.counters, sizeof(c)) != 0) return fail(SPRL_E_DEVICE, be::last_error());
                    c.active_slots = c.active_last;
                    e->nn_rows += (int64_t)(c.leaf_rows - e->last_leaf_rows);
                    e->last_leaf_rows = c.leaf_rows;
                    if (c.error != ERR_NONE) break;
                    if (c.games_done >= (uint32_t)e->num_games && c.active_slots == 0 && c.leaf_total == 0) break;
                }
                if (e->marks.size() >= 4096) resolve_marks(e);
                continue;
            }
            if (be::sync() != 0 || be::d2h(&c, P.counters, sizeof(c)) != 0) return fail(SPRL_E_DEVICE, be::last_error());
            c.active_slots = c.active_last;
            if (c.error != ERR_NONE) break;
            const int bucket = e->nn_bucket;
            int batch = (int)((c.leaf_total + (uint32_t)bucket - 1) / (uint32_t)bucket) * bucket;
            if (batch > max_batch_rows) batch = max_batch_rows;
            if (c.leaf_total > 0) {
                void* n0 = e->cfg.profile ? be::mark() : nullptr;
                int rc;
                char err[512] = { 0 };
                if (e->forward_cb) {
                    rc = e->forward_cb(e->forward_user, P.nn_dense, batch, e->nn_logits, e->nn_value);
                    if (rc) snprintf(err, sizeof(err), "forward callback returned %d", rc);
                } else {
                    // on the engine's own stream when it has one (boards wider than 8, other architectures: batch size on the host)
                    rc = (e->stream && e->torch.forward_on)
                             ? e->torch.forward_on(e->torch_model, P.nn_dense, batch, e->g.planes, e->g.rows, e->g.cols, e->nn_logits,
                                                   e->g.A, e->nn_value, e->stream, err, (int)sizeof(err))
                             : e->torch.forward(e->torch_model, P.nn_dense, batch, e->g.planes, e->g.rows, e->g.cols, e->nn_logits,
                                                e->g.A, e->nn_value, err, (int)sizeof(err));
                }
                if (rc) {
                    e->running = false;
                    return fail(SPRL_E_MODEL, std::string("network forward failed: ") + err);
                }
                if (e->cfg.profile) {
                    e->marks.push_back(n0);
                    e->marks.push_back(be::mark());
                    e->mark_kind.push_back(1);
                }
                e->nn_batches++;
                if (split) be::event_record(e->ev_fwd, e->stream);
                e->nn_rows += batch;
            }
            if (c.games_done >= (uint32_t)e->num_games && c.active_slots == 0 && c.leaf_total == 0) break;
        }
        if (e->marks.size() >= 4096) resolve_marks(e);
    }
    e->rounds += rounds;
    if (!net && (be::sync() != 0 || be::d2h(&c, P.counters, sizeof(c)) != 0)) return fail(SPRL_E_DEVICE, be::last_error());
    e->seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (games_done) *games_done = (int32_t)c.games_done;
    if (active_slots) *active_slots = (int32_t)c.active_slots;
    return check_device_error(e, c);
}

// profile = 2: time (ms) and launches of the trunk-convolution launches by kind - 0 plain, 1 with residual, 2 with the stem in its
// prologue, 3 with the heads / FC layers behind it
int sprl_engine_conv_kinds(sprl_engine* e, double* ms4, int64_t* launches4) {
    if (!e || !ms4 || !launches4) return fail(SPRL_E_CONFIG, "null argument");
    for (int k = 0; k < 4; ++k) {
        ms4[k] = 0.0;
        launches4[k] = 0;
    }
    if (e->torch_model && e->torch.profile_read_kinds && e->cfg.profile) e->torch.profile_read_kinds(e->torch_model, ms4, launches4);
    return 0;
}

// network evaluations queued by each game of the current / last self-play run (game index = the order games were started in)
int sprl_engine_game_evals(sprl_engine* e, uint32_t* out, int32_t num_games) {
    if (!e || !out) return fail(SPRL_E_CONFIG, "null argument");
    if (!e->P.rec_evals || num_games < 0 || num_games > e->num_games) return fail(SPRL_E_STATE, "no self-play run with that many games");
    be::bind(e->cfg.device, e->stream);
    if (be::sync() != 0 || be::d2h(out, e->P.rec_evals, (size_t)num_games * sizeof(uint32_t)) != 0) return fail(SPRL_E_DEVICE, be::last_error());
    return 0;
}

int sprl_engine_collect(sprl_engine* e, sprl_records* out) {
    if (!e || !out) return fail(SPRL_E_CONFIG, "null argument");
    be::bind(e->cfg.device, e->stream);
    if (!e->running) return fail(SPRL_E_STATE, "no run in progress");
    EngineParams& P = e->P;
    const Geom& g = e->g;
    const int ng = e->num_games;
    const size_t np = (size_t)ng * (size_t)P.max_plies;
    std::vector<int32_t> nplies((size_t)ng);
    std::vector<int8_t> winners((size_t)ng);
    const size_t WDS = (size_t)g.words;
    std::vector<uint64_t> boards(np * 2 * WDS);
    std::vector<uint8_t> movers(np);
    std::vector<float> pdfs(np * (size_t)g.A);
    int rc = be::sync();
    rc |= be::d2h(nplies.data(), P.rec_nplies, nplies.size() * sizeof(int32_t));
    rc |= be::d2h(winners.data(), P.rec_winner, winners.size());
    rc |= be::d2h(boards.data(), P.rec_boards, boards.size() * sizeof(uint64_t));
    rc |= be::d2h(movers.data(), P.rec_movers, movers.size());
    rc |= be::d2h(pdfs.data(), P.rec_pdf, pdfs.size() * sizeof(float));
    if (rc) return fail(SPRL_E_DEVICE, be::last_error());
    RecordsOwner* o = new RecordsOwner();
    o->ply_offset.resize((size_t)ng + 1);
    int64_t total = 0;
    for (int i = 0; i < ng; ++i) {
        o->ply_offset[(size_t)i] = (int32_t)total;
        if (nplies[(size_t)i] <= 0) {
            delete o;
            return fail(SPRL_E_STATE, "collect called before every game has finished");
        }
        total += nplies[(size_t)i];
    }
    o->ply_offset[(size_t)ng] = (int32_t)total;
    o->boards.resize((size_t)total * (size_t)g.cells);
    o->movers.resize((size_t)total);
    o->pdfs.resize((size_t)total * (size_t)g.A);
    o->winners = winners;
    for (int i = 0; i < ng; ++i) {
        for (int p = 0; p < nplies[(size_t)i]; ++p) {
            const size_t src = (size_t)i * (size_t)P.max_plies + (size_t)p;
            const size_t dst = (size_t)o->ply_offset[(size_t)i] + (size_t)p;
            const uint64_t* p0 = &boards[src * 2 * WDS];
            const uint64_t* p1 = p0 + WDS;
            int8_t* b = &o->boards[dst * (size_t)g.cells];
            for (int c = 0; c < g.cells; ++c)
                b[c] = ((p0[c >> 6] >> (c & 63)) & 1) ? 0 : (((p1[c >> 6] >> (c & 63)) & 1) ? 1 : -1);
            o->movers[dst] = (int8_t)movers[src];
            memcpy(&o->pdfs[dst * (size_t)g.A], &pdfs[src * (size_t)g.A], (size_t)g.A * sizeof(float));
        }
    }
    memset(out, 0, sizeof(*out));
    out->game = e->cfg.game;
    out->num_games = ng;
    out->rows = g.rows;
    out->cols = g.cols;
    out->cells = g.cells;
    out->actions = g.A;
    out->nsym = g.nsym;
    out->use_symmetry = e->cfg.use_symmetry ? 1 : 0;
    out->history = g.hist;
    out->planes = g.planes;
    out->total_plies = total;
    out->ply_offset = o->ply_offset.data();
    out->boards = o->boards.data();
    out->movers = o->movers.data();
    out->pdfs = o->pdfs.data();
    out->winners = o->winners.data();
    out->owner_ = o;
    e->running = false;
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// finished-game records on the device (records_kernel.h)
// ---------------------------------------------------------------------------------------------------
namespace {
int64_t align16(int64_t v) { return (v + 15) & ~(int64_t)15; }
struct PackedLayout { int64_t offsets, winners, stones0, stones1, movers, pdfs, total; };
// byte offsets of the sections of a packed shard (the same rule as sprl_amd/distributed.py: section_offsets)
PackedLayout packed_layout(int64_t games, int64_t plies, int64_t actions, int64_t words) {
    PackedLayout L;
    int64_t p = 12 * 8;
    L.offsets = p; p = align16(p + 4 * (games + 1));
    L.winners = p; p = align16(p + games);
    L.stones0 = p; p = align16(p + 8 * words * plies);
    L.stones1 = p; p = align16(p + 8 * words * plies);
    L.movers = p; p = align16(p + plies);
    L.pdfs = p; p = align16(p + 4 * plies * actions);
    L.total = p;
    return L;
}
int records_ready(sprl_engine* e) {
    if (!e->running) return fail(SPRL_E_STATE, "no run in progress");
    if (e->rec_total >= 0) return 0;
    Counters c;
    if (be::sync() != 0 || be::d2h(&c, e->P.counters, sizeof(c)) != 0) return fail(SPRL_E_DEVICE, be::last_error());
    if (c.games_done < (uint32_t)e->num_games) return fail(SPRL_E_STATE, "records requested before every game has finished");
    int32_t total = 0;
    if (be::launch_records_scan(e->P) != 0 || be::sync() != 0 ||
        be::d2h(&total, e->P.rec_offsets + e->num_games, sizeof(total)) != 0)
        return fail(SPRL_E_DEVICE, be::last_error());
    e->rec_total = total;
    return 0;
}
}  // namespace

int sprl_engine_records_info(sprl_engine* e, int64_t* total_plies, int64_t* num_samples, int64_t* packed_bytes) {
    if (!e) return fail(SPRL_E_CONFIG, "null engine");
    be::bind(e->cfg.device, e->stream);
    int rc = records_ready(e);
    if (rc) return rc;
    if (total_plies) *total_plies = e->rec_total;
    if (num_samples) *num_samples = e->rec_total * (e->cfg.use_symmetry ? e->g.nsym : 1);
    if (packed_bytes) *packed_bytes = packed_layout(e->num_games, e->rec_total, e->g.A, e->g.words).total;
    return 0;
}

int sprl_engine_pack_records(sprl_engine* e, void* dst_device, int64_t capacity_bytes) {
    if (!e || !dst_device) return fail(SPRL_E_CONFIG, "null argument");
    be::bind(e->cfg.device, e->stream);
    int rc = records_ready(e);
    if (rc) return rc;
    const PackedLayout L = packed_layout(e->num_games, e->rec_total, e->g.A, e->g.words);
    if (capacity_bytes < L.total) return fail(SPRL_E_CONFIG, "destination too small for the packed records (sprl_engine_records_info)");
    if ((uintptr_t)dst_device & 15u) return fail(SPRL_E_CONFIG, "destination must be 16-byte aligned");
    uint8_t* d = (uint8_t*)dst_device;
    if (be::dmemset(d, 0, (size_t)L.total) != 0) return fail(SPRL_E_DEVICE, be::last_error());     // padding bytes are zero
    RecPacked o;
    o.head = (int64_t*)d;
    o.offsets = (int32_t*)(d + L.offsets);
    o.winners = (int8_t*)(d + L.winners);
    o.stones0 = (uint64_t*)(d + L.stones0);
    o.stones1 = (uint64_t*)(d + L.stones1);
    o.movers = d + L.movers;
    o.pdfs = (float*)(d + L.pdfs);
    if (be::launch_records_pack(e->cfg.game, e->P, o, e->cfg.use_symmetry ? 1 : 0) != 0 || be::sync() != 0)
        return fail(SPRL_E_DEVICE, be::last_error());
    return 0;
}

int sprl_engine_expand_records(sprl_engine* e, float* states_device, float* distributions_device, float* outcomes_device,
                               int64_t capacity_samples) {
    if (!e || !states_device || !distributions_device || !outcomes_device) return fail(SPRL_E_CONFIG, "null argument");
    be::bind(e->cfg.device, e->stream);
    int rc = records_ready(e);
    if (rc) return rc;
    const int ns = e->cfg.use_symmetry ? e->g.nsym : 1;
    if (capacity_samples < e->rec_total * ns) return fail(SPRL_E_CONFIG, "destination too small for the expanded samples");
    RecExpanded o;
    o.states = states_device;
    o.dists = distributions_device;
    o.outcomes = outcomes_device;
    o.nsym = ns;
    if (be::launch_records_expand(e->cfg.game, e->P, o) != 0 || be::sync() != 0) return fail(SPRL_E_DEVICE, be::last_error());
    return 0;
}

int sprl_engine_finish(sprl_engine* e) {
    if (!e) return fail(SPRL_E_CONFIG, "null engine");
    if (!e->running) return fail(SPRL_E_STATE, "no run in progress");
    e->running = false;
    return 0;
}

int sprl_engine_run(sprl_engine* e, int32_t num_games, sprl_records* out) {
    int rc = sprl_engine_begin(e, num_games);
    if (rc) return rc;
    int32_t done = 0, active = 0;
    const int chunk = e->eval_kind == SPRL_EVAL_NETWORK ? 32 : 64;
    do {
        rc = sprl_engine_step(e, chunk, &done, &active);
        if (rc) return rc;
    } while (done < num_games);
    return sprl_engine_collect(e, out);
}

// ---------------------------------------------------------------------------------------------------
// match play
// ---------------------------------------------------------------------------------------------------
int sprl_match_play(const sprl_config* cfg, const sprl_match_agent* agent0, const sprl_match_agent* agent1,
                    int32_t num_games, int8_t* winners, int32_t* nplies, int16_t* actions, int32_t max_plies) {
    if (!cfg || !agent0 || !agent1 || !winners || !nplies) return fail(SPRL_E_CONFIG, "null argument");
    if (num_games < 1) return fail(SPRL_E_CONFIG, "num_games must be >= 1");
    if (actions && max_plies < 1) return fail(SPRL_E_CONFIG, "max_plies must be >= 1 when actions are requested");
    sprl_config c = *cfg;
    const int pairs = c.concurrent_games < num_games ? c.concurrent_games : num_games;
    if (pairs < 1) return fail(SPRL_E_CONFIG, "concurrent_games must be >= 1");
    c.concurrent_games = 2 * pairs;                  // slot p: agent 0's tree, slot pairs + p: agent 1's tree
    if (actions) c.max_plies = max_plies;
    sprl_engine* e = nullptr;
    int rc = sprl_engine_create(&c, &e);
    if (rc) return rc;
    EngineParams& P = e->P;
    const sprl_match_agent* ag[2] = { agent0, agent1 };
    void* models[2] = { nullptr, nullptr };
    auto done = [&](int code) {
        for (void* m : models)
            if (m) e->torch.release(m);
        sprl_engine_destroy(e);
        return code;
    };
    bool any_net = false;
    for (int k = 0; k < 2; ++k) {
        P.m_use_sym[k] = ag[k]->use_symmetry ? 1 : 0;
        P.m_init_q_zero[k] = ag[k]->init_q == SPRL_INITQ_ZERO ? 1 : 0;
        if (ag[k]->forward) {
            P.m_eval_kind[k] = EVAL_NETWORK;
        } else if (!ag[k]->model) {
            return done(fail(SPRL_E_CONFIG, "agent has neither a model nor a forward hook"));
        } else if (strcmp(ag[k]->model, "random") == 0) {
            P.m_eval_kind[k] = EVAL_RANDOM;
        } else if (strcmp(ag[k]->model, "heuristic") == 0) {
            if (c.game != SPRL_OTHELLO) return done(fail(SPRL_E_CONFIG, "the heuristic evaluator exists for Othello only"));
            P.m_eval_kind[k] = EVAL_HEURISTIC;
        } else {
            if ((rc = load_network(e, ag[k]->model, &models[k])) != 0) return done(rc);
            P.m_eval_kind[k] = EVAL_NETWORK;
        }
        any_net = any_net || P.m_eval_kind[k] == EVAL_NETWORK;
    }
    const size_t na = (size_t)num_games * (size_t)P.max_plies;
    bool ok = true;
    ok = ok && (P.mailbox = (Mailbox*)dev_alloc(e, 2 * (size_t)P.num_slots * sizeof(Mailbox)));      // two entries per slot (step_match)
    ok = ok && (P.match_actions = (int16_t*)dev_alloc(e, na * sizeof(int16_t)));
    ok = ok && (P.rec_nplies = (int32_t*)dev_alloc(e, (size_t)num_games * sizeof(int32_t)));
    ok = ok && (P.rec_winner = (int8_t*)dev_alloc(e, (size_t)num_games));
    if (!ok) return done(fail(SPRL_E_NOMEM, std::string("match allocation failed (") + be::last_error() + ")"));
    P.num_games = num_games;
    std::vector<GameCtl> ctl((size_t)P.num_slots);
    memset(ctl.data(), 0, ctl.size() * sizeof(GameCtl));
    for (int s = 0; s < P.num_slots; ++s) {
        ctl[(size_t)s].status = ST_FRESH;
        ctl[(size_t)s].arena = (uint32_t)s;
    }
    std::vector<uint32_t> used((size_t)(P.num_slots + P.num_spare), 0u);
    for (int s = 0; s < P.num_slots; ++s) used[(size_t)s] = 1u;
    Counters cn;
    memset(&cn, 0, sizeof(cn));
    rc = 0;
    rc |= be::h2d(P.ctl, ctl.data(), ctl.size() * sizeof(GameCtl));
    rc |= be::h2d(P.arena_used, used.data(), used.size() * sizeof(uint32_t));
    rc |= be::h2d(P.counters, &cn, sizeof(cn));
    rc |= be::dmemset(P.mailbox, 0, 2 * (size_t)P.num_slots * sizeof(Mailbox));
    rc |= be::dmemset(P.match_actions, 0xff, na * sizeof(int16_t));
    rc |= be::dmemset(P.rec_nplies, 0, (size_t)num_games * sizeof(int32_t));
    rc |= be::sync();
    if (rc) return done(fail(SPRL_E_DEVICE, be::last_error()));

    const int floats_per_leaf = e->g.planes * e->g.cells;
    const int max_rows = P.num_slots * P.max_queue;
    P.rounds = any_net ? 1 : 16;
    // every launch either advances a search or hands a move over; bound the loop all the same
    const int64_t launch_cap = (int64_t)((num_games + pairs - 1) / pairs) * (int64_t)e->g.default_max_plies *
                               ((int64_t)P.num_traversals + 4) + 1024;
    for (int64_t it = 0;; ++it) {
        if (it > launch_cap) return done(fail(SPRL_E_STATE, "match did not finish within its launch bound"));
        P.launch_seq = (uint32_t)(it + 1);
        if (!any_net) be::dmemset(&P.counters->active_slots, 0, sizeof(uint32_t));      // (with a network: cleared by the leaf scan)
        if (be::launch_match(c.game, P) != 0) return done(fail(SPRL_E_DEVICE, be::last_error()));
        if (any_net && be::launch_compact(P, floats_per_leaf) != 0) return done(fail(SPRL_E_DEVICE, be::last_error()));
        if (be::sync() != 0 || be::d2h(&cn, P.counters, sizeof(cn)) != 0) return done(fail(SPRL_E_DEVICE, be::last_error()));
        if (any_net) cn.active_slots = cn.active_last;
        if (cn.error != ERR_NONE) return done(check_device_error(e, cn));
        if (any_net && cn.leaf_total > 0) {
            // rows [0, split) belong to agent 0's trees, [split, total) to agent 1's: one forward per agent
            uint32_t split = 0;
            if (be::d2h(&split, P.leaf_offset + pairs, sizeof(split)) != 0) return done(fail(SPRL_E_DEVICE, be::last_error()));
            const int lo[2] = { 0, (int)split }, hi[2] = { (int)split, (int)cn.leaf_total };
            for (int k = 0; k < 2; ++k) {
                const int rows = hi[k] - lo[k];
                if (rows <= 0 || P.m_eval_kind[k] != EVAL_NETWORK) continue;
                int batch = (rows + e->nn_bucket - 1) / e->nn_bucket * e->nn_bucket;   // padding rows land on rows that
                if (lo[k] + batch > max_rows) batch = max_rows - lo[k];                 // are rewritten or never read
                const float* in = P.nn_dense + (size_t)lo[k] * (size_t)floats_per_leaf;
                float* lg = e->nn_logits + (size_t)lo[k] * (size_t)e->g.A;
                float* va = e->nn_value + lo[k];
                char err[512] = { 0 };
                int frc;
                if (ag[k]->forward) {
                    frc = ag[k]->forward(ag[k]->forward_user, in, batch, lg, va);
                    if (frc) snprintf(err, sizeof(err), "forward callback returned %d", frc);
                } else {
                    frc = e->torch.forward(models[k], in, batch, e->g.planes, e->g.rows, e->g.cols, lg, e->g.A, va, err, (int)sizeof(err));
                }
                if (frc) return done(fail(SPRL_E_MODEL, std::string("network forward failed: ") + err));
                e->nn_batches++;
                e->nn_rows += batch;
            }
        }
        e->launches++;
        if (cn.games_done >= (uint32_t)num_games) break;
        if (cn.active_slots == 0) return done(fail(SPRL_E_STATE, "match stalled: no active tree but games remain"));
    }
    rc = be::sync();
    rc |= be::d2h(winners, P.rec_winner, (size_t)num_games);
    rc |= be::d2h(nplies, P.rec_nplies, (size_t)num_games * sizeof(int32_t));
    if (actions) rc |= be::d2h(actions, P.match_actions, na * sizeof(int16_t));
    if (rc) return done(fail(SPRL_E_DEVICE, be::last_error()));
    return done(0);
}

void sprl_records_free(sprl_records* r) {
    if (!r || !r->owner_) return;
    delete (RecordsOwner*)r->owner_;
    memset(r, 0, sizeof(*r));
}

int sprl_engine_stats(sprl_engine* e, sprl_stats* out) {
    if (!e || !out) return fail(SPRL_E_CONFIG, "null argument");
    be::bind(e->cfg.device, e->stream);
    memset(out, 0, sizeof(*out));
    resolve_marks(e);
    std::vector<GameCtl> ctl((size_t)e->P.num_slots);
    if (be::sync() != 0 || be::d2h(ctl.data(), e->P.ctl, ctl.size() * sizeof(GameCtl)) != 0)
        return fail(SPRL_E_DEVICE, be::last_error());
    *out = e->carried;                           // finished runs on this engine
    add_slot_stats(ctl, out);
    out->rounds = e->rounds;
    out->kernel_launches = e->launches;
    out->nn_batches = e->nn_batches;
    out->nn_rows = e->nn_rows;
    out->seconds_total = e->seconds;
    out->kernel_ms = e->kernel_ms;
    out->nn_ms = e->nn_ms;
    out->hbm_bytes = (int64_t)e->hbm_bytes;
    if (e->torch_model && e->torch.profile_read && e->cfg.profile)
    {
        e->torch.profile_read(e->torch_model, &out->conv_ms, &out->conv_launches, &out->conv_boards);
        // with the batch size left on the device the plugin only knows the capacity: the real rows are the engine's count
        if (e->dev_batch && e->nn_batches > 0)
            out->conv_boards = (int64_t)((double)e->nn_rows * (double)out->conv_launches / (double)e->nn_batches);
    }
    return 0;
}

int sprl_profile_busy(int kind, double* busy_ms, double* sum_ms) {
    double sum = 0.0, busy = 0.0;
    if (kind == 0) busy = be::busy_ms(&sum);
    else if (kind == 1 && g_conv_busy) busy = g_conv_busy(&sum);
    else return fail(SPRL_E_CONFIG, "sprl_profile_busy: kind 0 = tree kernel, 1 = trunk convolution (needs a loaded network)");
    if (busy_ms) *busy_ms = busy;
    if (sum_ms) *sum_ms = sum;
    return 0;
}

void sprl_profile_busy_reset(void) {
    be::busy_reset();
    if (g_conv_busy_reset) g_conv_busy_reset();
}

// ---------------------------------------------------------------------------------------------------------
// records -> training samples (SelfPlay.hpp:86-92,127-133,151-189; GridWorker.hpp:146-171)
// ---------------------------------------------------------------------------------------------------------
int64_t sprl_records_num_samples(const sprl_records* r) {
    if (!r) return 0;
    return r->total_plies * (r->use_symmetry ? r->nsym : 1);
}

int sprl_records_expand_boards(const sprl_records* r, int8_t* boards, int8_t* players) {
    // boards: [N][history][cells]; plies before the start of the game are written as -2 (undefined in the reference)
    if (!r || !boards || !players) return fail(SPRL_E_CONFIG, "null argument");
    const int ns = r->use_symmetry ? r->nsym : 1;
    const int H = r->history, cells = r->cells;
    for (int gi = 0; gi < r->num_games; ++gi)
        for (int p = r->ply_offset[gi]; p < r->ply_offset[gi + 1]; ++p)
            for (int s = 0; s < ns; ++s) {
                int8_t* out = boards + ((size_t)p * ns + s) * (size_t)H * (size_t)cells;
                memset(out, -2, (size_t)H * (size_t)cells);
                for (int t = 0; t < H && p - t >= r->ply_offset[gi]; ++t) {
                    const int8_t* in = r->boards + (size_t)(p - t) * (size_t)cells;
                    for (int c = 0; c < cells; ++c) out[(size_t)t * cells + map_cell(r->game, s, c)] = in[c];
                }
                players[(size_t)p * ns + s] = r->movers[p];
            }
    return 0;
}

int sprl_records_expand(const sprl_records* r, float* states, float* distributions, float* outcomes) {
    if (!r || !states || !distributions || !outcomes) return fail(SPRL_E_CONFIG, "null argument");
    const int ns = r->use_symmetry ? r->nsym : 1;
    const int cells = r->cells, A = r->actions, H = r->history, PL = r->planes;
    for (int gi = 0; gi < r->num_games; ++gi) {
        const int8_t w = r->winners[gi];
        for (int p = r->ply_offset[gi]; p < r->ply_offset[gi + 1]; ++p) {
            const int8_t mover = r->movers[p];
            const float* pdf = r->pdfs + (size_t)p * (size_t)A;
            const float reward = w < 0 ? 0.0f : (w == mover ? 1.0f : -1.0f);   // OthelloNode.cpp:94-100
            for (int s = 0; s < ns; ++s) {
                const size_t n = (size_t)p * (size_t)ns + (size_t)s;
                float* st = states + n * (size_t)PL * (size_t)cells;
                float* di = distributions + n * (size_t)A;
                memset(st, 0, (size_t)PL * (size_t)cells * sizeof(float));       // missing history = zero planes
                for (int t = 0; t < H && p - t >= r->ply_offset[gi]; ++t) {       // GridWorker.hpp:152-166
                    const int8_t* in = r->boards + (size_t)(p - t) * (size_t)cells;
                    for (int c = 0; c < cells; ++c) {
                        const int tc = map_cell(r->game, s, c);
                        if (in[c] == mover) st[(size_t)(2 * t) * cells + tc] = 1.0f;
                        else if (in[c] >= 0) st[(size_t)(2 * t + 1) * cells + tc] = 1.0f;
                    }
                }
                for (int c = 0; c < cells; ++c) st[(size_t)(2 * H) * cells + c] = mover == 0 ? 1.0f : 0.0f;   // colour plane
                for (int a = 0; a < A; ++a) di[map_action(r->game, s, a)] = pdf[a];
                outcomes[n] = reward;
            }
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// .npy v1.0 writer with the reference's exact header bytes (utils/npy.hpp:430-476)
// ---------------------------------------------------------------------------------------------------------
static int write_npy_f32(const std::string& path, const float* data, const std::vector<uint64_t>& shape) {
    std::string tuple;
    char num[32];
    if (shape.size() == 1) {
        snprintf(num, sizeof(num), "(%llu,)", (unsigned long long)shape[0]);
        tuple = num;
    } else {
        tuple = "(";
        for (size_t i = 0; i < shape.size(); ++i) {
            snprintf(num, sizeof(num), "%llu", (unsigned long long)shape[i]);
            tuple += num;
            tuple += i + 1 < shape.size() ? ", " : ")";
        }
    }
    std::string dict = "{'descr': '<f4', 'fortran_order': False, 'shape': " + tuple + ", }";
    size_t length = 6 + 2 + 2 + dict.size() + 1;
    size_t pad = 16 - length % 16;
    size_t count = 1;
    for (uint64_t s : shape) count *= (size_t)s;
    std::string tmp = path + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return -1;
    const unsigned char magic[8] = { 0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0 };
    uint16_t hl = (uint16_t)(dict.size() + pad + 1);
    unsigned char le[2] = { (unsigned char)(hl & 0xff), (unsigned char)(hl >> 8) };
    bool ok = fwrite(magic, 1, 8, f) == 8 && fwrite(le, 1, 2, f) == 2 && fwrite(dict.data(), 1, dict.size(), f) == dict.size();
    for (size_t i = 0; ok && i < pad; ++i) ok = fputc(' ', f) != EOF;
    ok = ok && fputc('\n', f) != EOF;
    ok = ok && fwrite(data, sizeof(float), count, f) == count;
    ok = (fclose(f) == 0) && ok;
    if (!ok) {
        remove(tmp.c_str());
        return -1;
    }
    return rename(tmp.c_str(), path.c_str());
}

int sprl_records_slice(const sprl_records* r, int32_t first_game, int32_t num_games, sprl_records* out) {
    if (!r || !out) return fail(SPRL_E_CONFIG, "null argument");
    if (first_game < 0 || num_games < 1 || first_game + num_games > r->num_games) return fail(SPRL_E_CONFIG, "game range outside the records");
    *out = *r;
    const int64_t p0 = r->ply_offset[first_game], p1 = r->ply_offset[first_game + num_games];
    RecordsOwner* o = new RecordsOwner();                      // only the rebased offsets are owned by the view
    o->ply_offset.resize((size_t)num_games + 1);
    for (int i = 0; i <= num_games; ++i) o->ply_offset[(size_t)i] = (int32_t)(r->ply_offset[first_game + i] - p0);
    out->num_games = num_games;
    out->total_plies = p1 - p0;
    out->ply_offset = o->ply_offset.data();
    out->boards = r->boards + (size_t)p0 * (size_t)r->cells;
    out->movers = r->movers + p0;
    out->pdfs = r->pdfs + (size_t)p0 * (size_t)r->actions;
    out->winners = r->winners + first_game;
    out->owner_ = o;
    return 0;
}

int sprl_write_v2(const char* path, const sprl_records* r) {
    if (!path || !r) return fail(SPRL_E_CONFIG, "null argument");
    const int64_t W = (r->cells + 63) / 64, n = r->total_plies, g = r->num_games;
    const PackedLayout L = packed_layout(g, n, r->actions, W);
    std::vector<uint8_t> buf((size_t)L.total, 0);
    const int64_t head[12] = { g, n, r->actions, r->cells, r->game, r->nsym, r->use_symmetry, r->rows, r->cols, W, r->history, 0 };
    memcpy(buf.data(), head, sizeof(head));
    memcpy(buf.data() + L.offsets, r->ply_offset, (size_t)(g + 1) * 4);
    memcpy(buf.data() + L.winners, r->winners, (size_t)g);
    uint64_t* s0 = (uint64_t*)(buf.data() + L.stones0);
    uint64_t* s1 = (uint64_t*)(buf.data() + L.stones1);
    for (int64_t p = 0; p < n; ++p)
        for (int c = 0; c < r->cells; ++c) {
            const int8_t v = r->boards[(size_t)p * (size_t)r->cells + (size_t)c];
            if (v == 0) s0[p * W + (c >> 6)] |= 1ull << (c & 63);
            else if (v == 1) s1[p * W + (c >> 6)] |= 1ull << (c & 63);
        }
    memcpy(buf.data() + L.movers, r->movers, (size_t)n);
    memcpy(buf.data() + L.pdfs, r->pdfs, (size_t)n * (size_t)r->actions * 4);
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return fail(SPRL_E_IO, "io error: failed to open a file.");
    const unsigned char magic[8] = { 'S', 'P', 'R', 'L', 'v', '2', 1, 0 };
    const int64_t size = L.total;
    bool ok = fwrite(magic, 1, 8, f) == 8 && fwrite(&size, 8, 1, f) == 1 && fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path) != 0) {
        remove(tmp.c_str());
        return fail(SPRL_E_IO, "io error: failed to write the record file.");
    }
    return 0;
}

int sprl_write_npy(const char* path_prefix, const sprl_records* r) {
    if (!path_prefix || !r) return fail(SPRL_E_CONFIG, "null argument");
    const int64_t n = sprl_records_num_samples(r);
    std::vector<float> states((size_t)n * (size_t)r->planes * (size_t)r->cells), dists((size_t)n * (size_t)r->actions), outs((size_t)n);
    int rc = sprl_records_expand(r, states.data(), dists.data(), outs.data());
    if (rc) return rc;
    std::string p(path_prefix);
    // controller polls for all three files (scripts/othello_controller.py:83-93): outcomes goes last, each
    // file appears atomically
    if (write_npy_f32(p + "_states.npy", states.data(), { (uint64_t)n, (uint64_t)r->planes, (uint64_t)r->rows, (uint64_t)r->cols }) != 0 ||
        write_npy_f32(p + "_distributions.npy", dists.data(), { (uint64_t)n, (uint64_t)r->actions }) != 0 ||
        write_npy_f32(p + "_outcomes.npy", outs.data(), { (uint64_t)n }) != 0)
        return fail(SPRL_E_IO, "io error: failed to open a file.");
    return 0;
}

}  // extern "C"
