// worker_main.cpp — the native self-play worker: the process + filesystem contract of the reference's OTHWorker / C4Worker /
// GoWorker (cpp/src/OTHWorker.cpp:31-70, C4Worker.cpp, GoWorker.cpp) and of runWorker (cpp/src/selfplay/GridWorker.hpp:84-198),
// over the C ABI of include/sprl_amd.h.
//
//   sprl_worker <game> <task_id> <num_tasks> [options]          game: othello | connect_four | go7 | go9 | go19
//
// * reads   data/models/<run>/traced_<run>_iteration_<i-1>.pt (GridWorker.hpp:36-55: polled every 30 s, +5 s to let the
//           writer finish; iteration 0 = the built-in initial evaluator, GridWorker.hpp:125-127)
// * writes  data/games/<run>/<group>/<task>/<run>_iteration_<i>_{states,distributions,outcomes}.npy
//           (GridWorker.hpp:116,173-196; group = task / (num_tasks / num_groups), OTHWorker.cpp:44), byte-identical headers,
//           each file complete before it appears, outcomes last
// * iteration 0 runs the init budgets, later iterations the steady-state ones (GridWorker.hpp:118-121; the reference's
//   parameter-shadowing bug, SURVEY Q2, is not reproduced: the steady-state values are the worker constants)
// * exit codes / messages: wrong argument count -> usage on stderr, 1 (OTHWorker.cpp:34-37); num_tasks other than the
//   constant the reference asserts on (OTHWorker.cpp:42) -> message, 1; engine errors -> message, 2.
//
// One MI355X stands in for many CPU tasks: --cover K plays the games of task ids [task_id, task_id + K) concurrently and
// writes each task's files into that task's own directory, so the unmodified Python controller
// (scripts/othello_controller.py:66-125) finds exactly the files it polls for.  One engine is kept across the
// steady-state iterations (only its model changes).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <filesystem>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sprl_amd.h"

namespace {

struct Constants {
    const char* name;
    int game;
    const char* run_name;
    int num_groups, num_worker_tasks, num_iters;
    int init_games, init_traversals, init_max_batch, init_max_queue;
    int games, traversals, max_batch, max_queue;
    float dir_eps, dir_alpha;
};

// OTHWorker.cpp:12-32, C4Worker.cpp:11-31, GoWorker.cpp:11-29.  go9 / go19: the Go worker's constants with BASELINE's
// 1600 iterations per move (configs 4 and 5); the reference compiles Go at 7x7 only.
const Constants WORKERS[] = {
    { "othello", SPRL_OTHELLO, "orangutan_alpha", 4, 384, 50, 3, 131072, 1, 1, 3, 8192, 8, 4, 0.25f, 0.3f },
    { "connect_four", SPRL_CONNECT_FOUR, "c4_test", 1, 1, 25, 10, 2048, 1, 1, 5, 512, 8, 4, 0.25f, 0.5f },
    { "go7", SPRL_GO7, "panda_alpha", 4, 384, 100, 3, 262144, 1, 1, 3, 32768, 16, 8, 0.25f, 0.2f },
    { "go9", SPRL_GO9, "panda_9x9", 4, 384, 100, 3, 1600, 16, 8, 3, 1600, 16, 8, 0.25f, 0.2f },
    { "go19", SPRL_GO19, "panda_19x19", 4, 384, 100, 3, 1600, 16, 8, 3, 1600, 16, 8, 0.25f, 0.2f },
};

int usage() {
    fprintf(stderr, "Usage: sprl_worker <game> <task_id> <num_tasks> [--cover K] [--run-name NAME] [--num-iters N] [--root DIR]\n"
                    "                   [--device D] [--seed S] [--concurrent GAMES] [--format v1|v2] [--poll-seconds S]\n"
                    "                   [--games N --traversals T --max-batch B --max-queue Q   (steady-state budgets)]\n"
                    "                   [--init-games N --init-traversals T --init-max-batch B --init-max-queue Q]\n"
                    "                   [--num-tasks-const N --num-groups G] [--model-iter0 random|heuristic|PATH] [--evaluator-override M]\n"
                    "                   [--resign-threshold T --resign-min-ply P]\n");
    return 1;
}

bool wait_model(const std::string& path, int iteration, double poll_seconds) {
    namespace fs = std::filesystem;
    while (!fs::exists(path)) {
        printf("Spinning on traced model from iteration %d...\n", iteration);      // GridWorker.hpp:46
        fflush(stdout);
        std::this_thread::sleep_for(std::chrono::duration<double>(poll_seconds));
    }
    std::this_thread::sleep_for(std::chrono::duration<double>(poll_seconds < 5.0 ? poll_seconds / 6.0 : 5.0));   // :52
    return true;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 4) return usage();
    const Constants* base = nullptr;
    for (const Constants& c : WORKERS)
        if (strcmp(argv[1], c.name) == 0) base = &c;
    if (!base) return usage();
    Constants k = *base;
    char* end = nullptr;
    const long task_id = strtol(argv[2], &end, 10);
    if (*end) return usage();
    const long num_tasks = strtol(argv[3], &end, 10);
    if (*end) return usage();
    int cover = 1, device = 0, concurrent = 0, format = 1, num_iters = -1, populations = 1;
    long long seed = 0;
    double poll = 30.0;                                           // MODEL_PATH_WAIT_INTERVAL, GridWorker.hpp:23
    float resign = 0.0f;
    int resign_min_ply = 0;
    std::string root = ".", run_name = k.run_name, model0 = "random", model_override;
    for (int i = 4; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> const char* { return i + 1 < argc ? argv[++i] : nullptr; };
        const char* v = nullptr;
        if (a == "--cover") { if (!(v = val())) return usage(); cover = atoi(v); }
        else if (a == "--run-name") { if (!(v = val())) return usage(); run_name = v; }
        else if (a == "--num-iters") { if (!(v = val())) return usage(); num_iters = atoi(v); }
        else if (a == "--root") { if (!(v = val())) return usage(); root = v; }
        else if (a == "--device") { if (!(v = val())) return usage(); device = atoi(v); }
        else if (a == "--seed") { if (!(v = val())) return usage(); seed = atoll(v); }
        else if (a == "--concurrent") { if (!(v = val())) return usage(); concurrent = atoi(v); }
        else if (a == "--format") { if (!(v = val())) return usage(); format = strcmp(v, "v2") == 0 ? 2 : 1; }
        else if (a == "--poll-seconds") { if (!(v = val())) return usage(); poll = atof(v); }
        else if (a == "--games") { if (!(v = val())) return usage(); k.games = atoi(v); }
        else if (a == "--traversals") { if (!(v = val())) return usage(); k.traversals = atoi(v); }
        else if (a == "--max-batch") { if (!(v = val())) return usage(); k.max_batch = atoi(v); }
        else if (a == "--max-queue") { if (!(v = val())) return usage(); k.max_queue = atoi(v); }
        else if (a == "--init-games") { if (!(v = val())) return usage(); k.init_games = atoi(v); }
        else if (a == "--init-traversals") { if (!(v = val())) return usage(); k.init_traversals = atoi(v); }
        else if (a == "--init-max-batch") { if (!(v = val())) return usage(); k.init_max_batch = atoi(v); }
        else if (a == "--init-max-queue") { if (!(v = val())) return usage(); k.init_max_queue = atoi(v); }
        else if (a == "--num-tasks-const") { if (!(v = val())) return usage(); k.num_worker_tasks = atoi(v); }
        else if (a == "--num-groups") { if (!(v = val())) return usage(); k.num_groups = atoi(v); }
        else if (a == "--model-iter0") { if (!(v = val())) return usage(); model0 = v; }
        else if (a == "--evaluator-override") { if (!(v = val())) return usage(); model_override = v; }   // tests: the model-file
                                                          // rendez-vous still happens, the file's content is not loaded
        else if (a == "--resign-threshold") { if (!(v = val())) return usage(); resign = (float)atof(v); }
        else if (a == "--resign-min-ply") { if (!(v = val())) return usage(); resign_min_ply = atoi(v); }
        else if (a == "--populations") { if (!(v = val())) return usage(); populations = atoi(v); }
        else return usage();
    }
    if (num_tasks != k.num_worker_tasks) {                        // the reference asserts (OTHWorker.cpp:42)
        fprintf(stderr, "num_tasks must be %d for %s (OTHWorker.cpp:42 / GoWorker.cpp:41)\n", k.num_worker_tasks, k.name);
        return 1;
    }
    if (cover < 1 || task_id < 0 || task_id + cover > num_tasks || k.num_groups < 1 || k.num_worker_tasks % k.num_groups) return usage();
    if (populations < 1 || populations > cover || populations > 8) return usage();
    if (num_iters < 0) num_iters = k.num_iters;
    if (seed == 0)                                                // the reference seeds from random_device
        seed = (long long)(std::chrono::steady_clock::now().time_since_epoch().count() & 0x7fffffffffffLL) | 1;
    namespace fs = std::filesystem;
    const long group0 = task_id / (k.num_worker_tasks / k.num_groups);
    printf("Task %ld of %ld, in group %ld", task_id, num_tasks, group0);
    if (cover > 1) printf(" (standing in for %d tasks on one GPU)", cover);
    printf(".\n");

    std::vector<std::string> dirs;
    for (int t = 0; t < cover; ++t) {
        const long tid = task_id + t, group = tid / (k.num_worker_tasks / k.num_groups);           // OTHWorker.cpp:44-49
        const std::string d = root + "/data/games/" + run_name + "/" + std::to_string(group) + "/" + std::to_string(tid);
        std::error_code ec;
        if (fs::is_directory(d)) printf("Directory already exists: %s\n", d.c_str());             // GridWorker.hpp:97-107
        else if (fs::create_directories(d, ec)) printf("Created directory: %s\n", d.c_str());
        else {
            fprintf(stderr, "Failed to create directory: %s\n", d.c_str());
            return 2;
        }
        dirs.push_back(d);
    }

    // The worker loop for the covered tasks [first, first + count): one engine, kept across iterations.  --populations P splits
    // the covered tasks into P such ranges, each with its own engine on a private HIP stream and its own host thread: the tree
    // kernel of one population overlaps the network forward of the other (+7 % games/s on Othello).
    auto run_range = [&](const int first, const int count, const int pop) -> int {
    sprl_engine* eng = nullptr;
    int eng_sig[4] = { -1, -1, -1, -1 };                            // traversals, batch, queue, concurrent of the live engine
    sprl_config eng_cfg;                                            // the configuration the live engine was created with
    memset(&eng_cfg, 0, sizeof(eng_cfg));
    int next_stream = 1 + pop * (1 << 27);                          // disjoint RNG stream ranges per population
    auto fail = [&](const char* what) {
        fprintf(stderr, "%s: %s\n", what, sprl_last_error());
        if (eng) sprl_engine_destroy(eng);
        return 2;
    };
    const int cover = count;                                        // (shadows the whole range inside this population)
    for (int it = 0; it < num_iters; ++it) {
        if (pop == 0) printf("Starting iteration %d...\n", it);                                    // GridWorker.hpp:112
        fflush(stdout);
        std::string model = model0;
        if (it > 0) {
            model = root + "/data/models/" + run_name + "/traced_" + run_name + "_iteration_" + std::to_string(it - 1) + ".pt";
            wait_model(model, it - 1, poll);
            if (!model_override.empty()) model = model_override;
        }
        const int games = it == 0 ? k.init_games : k.games, trav = it == 0 ? k.init_traversals : k.traversals;
        const int mb = it == 0 ? k.init_max_batch : k.max_batch, mq = it == 0 ? k.init_max_queue : k.max_queue;
        const int total = games * cover;
        const int conc_cap = concurrent > 0 ? (concurrent + populations - 1) / populations : 0;
        int conc = conc_cap > 0 && conc_cap < total ? conc_cap : total;
        if (!eng || eng_sig[0] != trav || eng_sig[1] != mb || eng_sig[2] != mq || eng_sig[3] != conc) {
            if (eng) sprl_engine_destroy(eng);
            eng = nullptr;
            sprl_config cfg;
            if (sprl_config_default(k.game, &cfg) != 0) return fail("config");
            cfg.device = device;
            cfg.num_traversals = trav;
            cfg.max_batch = mb;
            cfg.max_queue = mq;
            cfg.dir_eps = k.dir_eps;
            cfg.dir_alpha = k.dir_alpha;
            cfg.seed = (uint64_t)seed;
            cfg.stream_base = next_stream;
            cfg.own_stream = populations > 1 ? 1 : 0;
            cfg.resign_threshold = resign;
            cfg.resign_min_ply = resign_min_ply;
            // The iteration-0 budgets (131 072 / 262 144 traversals per move) need 0.26 - 0.5 GiB of node arena per resident game:
            // when the games of all covered tasks do not fit into HBM together, keep fewer of them resident (the engine starts
            // the next game in a slot as soon as one ends) instead of giving up.
            const int wanted = conc;
            int rc = 0;
            for (;;) {
                cfg.concurrent_games = conc;
                rc = sprl_engine_create(&cfg, &eng);
                if (rc != SPRL_E_NOMEM || conc <= 1) break;
                conc = (conc + 1) / 2;
            }
            if (rc != 0) return fail("engine");
            if (conc != wanted) printf("HBM holds %d of the %d games at once; the rest start as slots free up.\n", conc, wanted);
            eng_sig[0] = trav; eng_sig[1] = mb; eng_sig[2] = mq; eng_sig[3] = wanted;
            eng_cfg = cfg;
        }
        next_stream += total;                // (an engine kept from the previous iteration continues its stream numbering itself)
        printf(model == "random" || model == "heuristic" ? "Using initial network...\n" : "Using traced PyTorch network...\n");
        sprl_records rec;
        for (;;) {
            if (sprl_engine_set_model(eng, model.c_str()) != 0) return fail("model");
            const int rc = sprl_engine_run(eng, total, &rec);
            if (rc == 0) break;
            // the record buffers of `total` games are allocated when the run begins: an engine whose arenas just fitted can still
            // fail THERE - the same back-off, with a fresh engine of half the resident games
            if (rc != SPRL_E_NOMEM || eng_cfg.concurrent_games <= 1) return fail("self-play");
            sprl_engine_destroy(eng);
            eng = nullptr;
            eng_cfg.concurrent_games = (eng_cfg.concurrent_games + 1) / 2;
            if (sprl_engine_create(&eng_cfg, &eng) != 0) return fail("engine");
            printf("Record buffers did not fit beside the arenas: %d games resident at once.\n", eng_cfg.concurrent_games);
        }
        for (int t = 0; t < cover; ++t) {
            sprl_records part;
            if (sprl_records_slice(&rec, t * games, games, &part) != 0) {
                sprl_records_free(&rec);
                return fail("slice");
            }
            const std::string prefix = dirs[(size_t)(first + t)] + "/" + run_name + "_iteration_" + std::to_string(it);
            const int rc = format == 2 ? sprl_write_v2((prefix + ".sprl2").c_str(), &part) : sprl_write_npy(prefix.c_str(), &part);
            sprl_records_free(&part);                        // the view owns its rebased offsets (include/sprl_amd.h)
            if (rc != 0) {
                sprl_records_free(&rec);
                return fail("write");
            }
        }
        printf("%d games played, %lld states collected.\n", total, (long long)sprl_records_num_samples(&rec));   // SelfPlay.hpp:241
        fflush(stdout);
        sprl_records_free(&rec);
    }
    if (eng) sprl_engine_destroy(eng);
    return 0;
    };

    if (populations == 1) return run_range(0, cover, 0);
    std::vector<std::thread> threads;
    std::vector<int> rcs((size_t)populations, 0);
    for (int p = 0; p < populations; ++p) {
        const int first = (int)((long long)cover * p / populations), last = (int)((long long)cover * (p + 1) / populations);
        threads.emplace_back([&, p, first, last] { rcs[(size_t)p] = run_range(first, last - first, p); });
    }
    for (auto& t : threads) t.join();
    for (int rc : rcs)
        if (rc) return rc;
    return 0;
}
