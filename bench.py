#!/usr/bin/env python3
"""bench.py — self-play games/sec (+ UCT node-expansions/sec) on MI355X, Othello 8x8 @ 800 iters/move.

A *step* is one self-play iteration in the reference's sense (runIteration, cpp/src/selfplay/SelfPlay.hpp:204-248):
every GPU plays `--games` games (default = its 4096 resident game slots) from the start position to the end,
with the reference worker's search constants (OTHWorker.cpp:24-28, constants.hpp:6-10), a random-init
2-block x 64-channel policy/value CNN evaluated in fp32 through LibTorch-ROCm, D4 symmetrisation, Dirichlet
noise, parent-Q init and sub-tree reuse, and emits the compact self-play records (gathered to rank 0 over RCCL
when N > 1).  Inputs are synthetic by construction (all games start from the rules' start position, weights are
random-init); everything is resident in HBM when the timed region starts.

Contract: `python bench.py --gpus N --steps K --warmup W` (N > 1 under torch.distributed.run, one rank per GPU).
W untimed warm-up steps, then exactly K timed steps bracketed by barrier + synchronize; MAX over ranks; rank 0
prints ONE JSON line.  `value` = whole-job games/sec.  Weak scaling: per-GPU work is fixed as N grows.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TF = 157.3            # dense fp32 matrix peak, MI355X_MICROARCH.md (v_mfma_f32_16x16x4_f32)
WINO_FLOP_PER_BOARD = 2 * 4 * 36 * 64 * 64
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_traversal(st, A=65, mask_bytes=8, board_bytes=64, s_in=3 * 64 * 4):
    """SURVEY.md §8(d) per-traversal figure, recomputed with this run's measured search shape."""
    trav = max(1, st["traversals"])
    D = st["levels"] / trav
    f_nn = st["nn_evals"] / trav
    f_g = st["gray_hits"] / trav
    select = D * (3 * A * 4 + mask_bytes + 4 + 4) + (D + 1) * 16
    backup = (D + 1) * 8
    expand = (f_nn + f_g) * (A * 4 + A * 4)
    create = f_nn * (2 * A * 4 + board_bytes * 2 + mask_bytes)
    leaf_io = f_nn * (s_in * 2 + (A + 1) * 4 * 2 + A * 4)
    return select + backup + expand + create + leaf_io, dict(D=D, f_nn=f_nn, f_gray=f_g)


def _cpu_worker(args):
    """One single-threaded reference worker process: the reference's selfPlay + GridNetwork on LibTorch-CPU,
    `games` games one after the other (each timed), like a reference worker's iteration (OTHWorker.cpp:17: 3 games)."""
    model, games, trav, stream = args
    os.environ["OMP_NUM_THREADS"] = "1"
    os.environ["MKL_NUM_THREADS"] = "1"
    import torch
    torch.set_num_threads(1)
    from oracle import pyref
    times, evals, plies = [], 0, 0
    for g in range(games):
        t0 = time.time()
        r = pyref.selfplay("othello", 0, 1, trav, 8, 4, 0.25, 0.3, 4242, stream + g, True, model_path=model)
        times.append(time.time() - t0)
        evals += r["evals"]
        plies += len(r["players"]) // 8
    return times, evals, plies


def host_cores():
    """Cores this process may really use: the scheduler affinity, capped by the cgroup CPU quota (a GPU box hands a
    one-GPU job a share of the host, not the whole machine)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(model_path, traversals, games_per_process=3, max_processes=0):
    """Reference CPU path timed on this box's host cores, SURVEY section 8(d) protocol: one single-threaded process per
    usable host core (count stated), >= 3 games per process, median games/s per core + aggregate.  kind = "reference"
    when the prebuilt oracle/_ref library (the reference's own sources compiled in place) travelled with the repo, else
    the oracle ("port") with a LibTorch-CPU forward callback."""
    import multiprocessing as mp
    import statistics
    from oracle import pyref
    cores = host_cores()
    if max_processes > 0:
        cores = min(cores, max_processes)
    if pyref.available(True):
        ctx = mp.get_context("spawn")
        t0 = time.time()
        with ctx.Pool(cores) as pool:
            res = pool.map(_cpu_worker, [(model_path, games_per_process, traversals, 1000 + 16 * i) for i in range(cores)])
        wall = time.time() - t0
        games = cores * games_per_process
        evals = sum(r[1] for r in res)
        per_core = [len(r[0]) / sum(r[0]) for r in res]          # games/s of each process while it was playing
        med = statistics.median(per_core)
        busy = max(sum(r[0]) for r in res)                       # the slowest process bounds the iteration
        return {"value": games / busy, "unit": "games/s", "cores": cores, "kind": "reference",
                "sample": f"{cores} single-thread processes (all usable host cores) x {games_per_process} games, Othello {traversals} "
                          f"it/move, batch 8/queue 4, same traced 2x64 CNN on LibTorch-CPU, g++ -O3 -DNDEBUG; slowest process "
                          f"{busy:.1f}s (pool wall incl. start-up {wall:.1f}s), {evals} network evals",
                "games_per_sec_per_core_median": med, "games_per_sec_per_core_min": min(per_core),
                "games_per_sec_per_core_max": max(per_core), "aggregate_of_per_core_rates": sum(per_core),
                "evals_per_sec": evals / busy, "games_per_process": games_per_process,
                "reference_deployment_cores": 384,               # README.md:124-128, OTHWorker.cpp:13
                "reference_deployment_games_per_sec": 384 * med}
    # fallback: the oracle with a torch-CPU forward (scalar port, 1 core)
    import torch
    from oracle import pyoracle as po
    torch.set_num_threads(1)
    net = torch.jit.load(model_path, map_location="cpu").eval()

    def fwd(x):
        with torch.no_grad():
            lo, va = net(torch.from_numpy(x))
        return lo.numpy(), va.numpy()

    cb = po.make_forward(fwd, po.GAME_OTHELLO)
    cfg = po.make_config(po.GAME_OTHELLO, traversals, eval_kind=po.EVAL_CALLBACK, forward=cb)
    t0 = time.time()
    r = po.selfplay(cfg, 1, 4242, 1000, True)
    dt = time.time() - t0
    return {"value": 1.0 / dt, "unit": "games/s", "cores": 1, "kind": "port",
            "sample": f"1 game, Othello {traversals} it/move, oracle + torch-CPU forward, {r['stats']['nn_evals']} evals"}


GAMES = {
    # game: engine name, board, actions, planes, default concurrent games, traversals, CNN blocks (the reference controllers'
    # MODEL_NUM_BLOCKS: othello/connect_four 2, go 6), batch/queue (worker constants), label
    "othello": dict(engine="othello", rows=8, cols=8, A=65, planes=3, concurrent=4096, traversals=800, blocks=2, bq="8/4",
                    label="Othello 8x8", sym="D4", noise="Dirichlet(0.25,0.3)", mask_bytes=8, board_bytes=64, populations=2),
    "connect_four": dict(engine="connect_four", rows=6, cols=7, A=7, planes=3, concurrent=4096, traversals=100, blocks=2, bq="8/4",
                         label="Connect Four 6x7", sym="mirror", noise="Dirichlet(0.25,0.5)", mask_bytes=8, board_bytes=42),
    "go7": dict(engine="go7", rows=7, cols=7, A=50, planes=17, concurrent=2048, traversals=1600, blocks=6, bq="16/8",
                label="Go 7x7", sym="D4", noise="Dirichlet(0.25,0.2)", mask_bytes=8, board_bytes=49),
    # resident games (round 3): 4096 for 9x9 / 2048 for 19x19 (two populations of half that) - the wide tree kernel now keeps
    # 4-8 waves per CU resident (its position history left LDS), a population's tree launch then fills every SIMD, and larger
    # network batches fill the any-board convolution better (profiles/r03l_*, r03t_bench_go*.json: 9x9 14.6 games/s at 2048
    # resident games, 15.7 at 4096; 19x19 5.9 at 1024, 6.1 at 2048).  A step is 4-6 minutes: --steps defaults to 1 for Go.
    "go9": dict(engine="go9", rows=9, cols=9, A=82, planes=17, concurrent=4096, traversals=1600, blocks=6, bq="16/8", steps=1,
                label="Go 9x9", sym="D4", noise="Dirichlet(0.25,0.2)", mask_bytes=11, board_bytes=81, populations=2),
    # config 5 names a resign threshold: on by default for this game (with random-init weights the decision is noise and games
    # end after ~min-ply moves; without it every game runs to the 722-ply cap, ~20 min per step)
    "go19": dict(engine="go19", rows=19, cols=19, A=362, planes=17, concurrent=2048, traversals=1600, blocks=6, bq="16/8", steps=1,
                 label="Go 19x19", sym="D4", noise="Dirichlet(0.25,0.2)", mask_bytes=46, board_bytes=361,
                 resign_threshold=0.05, resign_min_ply=60, populations=2),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: 2; 1 for the Go configurations, whose steps take minutes)")
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--game", default="othello", choices=sorted(GAMES),
                    help="othello = the BASELINE metric (config 2/3); go9 / go19 = BASELINE configs 4 / 5")
    ap.add_argument("--games", type=int, default=0, help="games per GPU per step (default: = --concurrent)")
    ap.add_argument("--concurrent", type=int, default=0, help="resident game slots per GPU (default per game: othello 4096)")
    ap.add_argument("--traversals", type=int, default=0, help="UCT iterations per move (default per game: othello 800, go 1600)")
    ap.add_argument("--model", default="cnn", choices=["cnn", "random", "heuristic"],
                    help="cnn = BASELINE config (traced CNN via LibTorch-ROCm); random/heuristic = tree kernels only")
    ap.add_argument("--blocks", type=int, default=0, help="residual blocks of the CNN (default per game: othello 2, go 6)")
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--rounds-per-call", type=int, default=64)
    ap.add_argument("--populations", type=int, default=0,
                    help="split the resident games of a GPU into this many engines, each on its own HIP stream and host thread: "
                         "one population's tree kernel and convolution tails overlap the other's CNN work (default: othello, "
                         "go9, go19: 2, other games 1)")
    ap.add_argument("--no-alone-pass", action="store_true",
                    help="with several populations: skip the extra one-population step that measures the kernels running alone")
    ap.add_argument("--resign-threshold", type=float, default=-1.0,
                    help="extension (BASELINE config 5), 0 = off as in the reference (default: off, go19: 0.05)")
    ap.add_argument("--resign-min-ply", type=int, default=-1, help="default 0 (go19: 60)")
    ap.add_argument("--allow-lab", action="store_true",
                    help="A/B measurements only: accept SPRL_* lab switches (the JSON line then names them in config.evaluator)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="othello: skip the bounded Go 9x9 / Go 19x19 samples (BASELINE configs 4 / 5) that follow the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--node-cap", type=int, default=0, help="nodes per game arena (0 = engine default)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI) on a multi-GPU node; gloo only to rehearse the N > 1 code path")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses GPU 0")
    args = ap.parse_args()
    G = GAMES[args.game]
    args.steps = args.steps or G.get("steps", 2)
    args.concurrent = args.concurrent or G["concurrent"]
    args.traversals = args.traversals or G["traversals"]
    args.blocks = args.blocks or G["blocks"]
    if args.resign_threshold < 0:
        args.resign_threshold = G.get("resign_threshold", 0.0)
    if args.resign_min_ply < 0:
        args.resign_min_ply = G.get("resign_min_ply", 0)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py: no MI355X visible (the engine has no CPU fallback)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)
    comm_device = dev if args.backend == "nccl" else torch.device("cpu")

    from sprl_amd import engine as E
    from sprl_amd.distributed import gather_packed
    from sprl_amd.network import make_network, trace_to_file

    games = args.games or args.concurrent
    lib = E.load_library()
    tmpdir = tempfile.mkdtemp(prefix="sprl_bench_")
    model_path = None
    if args.model == "cnn":
        model_path = trace_to_file(make_network(G["engine"], args.blocks, args.channels, seed=0),
                                   os.path.join(tmpdir, f"traced_bench_r{rank}.pt"), G["engine"])
    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    stream_cursor = [1 + rank * (1 << 24)]      # unique RNG streams for every game of every step of every rank and pass

    def measure(pops, steps, warmup, profile_mode=1):
        """One timed region: `pops` engines (each on its own HIP stream and host thread when pops > 1) play `steps` steps of
        `games` games per GPU.  Returns the wall time, the counter deltas and the busy times of the two profiled kernels.
        profile_mode 1: one HIP-event pair around the trunk convolutions of a forward (the timed region); 2: one pair per
        convolution launch (kernel durations, for the one-population pass outside the timed region)."""
        if args.concurrent % pops or games % pops:
            raise SystemExit("--populations must divide --concurrent and --games")
        engines = []
        t_load = time.perf_counter()
        for p in range(pops):
            cfg = E.default_config(G["engine"], lib, device=local_rank, concurrent_games=args.concurrent // pops,
                                   num_traversals=args.traversals, seed=args.seed, node_cap=args.node_cap,
                                   stream_base=stream_cursor[0],
                                   profile=0 if args.no_profile else profile_mode, own_stream=1 if pops > 1 else 0,
                                   resign_threshold=args.resign_threshold, resign_min_ply=args.resign_min_ply)
            stream_cursor[0] += (games // pops) * (steps + warmup + 2)
            en = E.Engine(cfg, lib)
            en.set_model(model_path if model_path else args.model)
            engines.append(en)
        t_load = time.perf_counter() - t_load
        gather_s = [0.0]
        shard_bytes = [0]
        last_shards = []                            # rank 0: the packed shards of the last emit, host bytes

        def verify_last_shards():
            """Outside the timed region: decode the last step's shards and check them (every rank's games present, pdfs are
            distributions) - the proof that what was gathered is usable, without numpy bit-unpacking inside the bracket."""
            from sprl_amd.distributed import unpack_records
            if not last_shards:
                return None
            un = [unpack_records(t.numpy()) for t in last_shards]
            for sh in un:
                assert sh["ply_offset"][-1] == sh["total_plies"] and abs(float(sh["pdfs"].sum()) - sh["total_plies"]) < 1e-2 * sh["total_plies"]
            return {"shards": len(un), "games": [int(sh["num_games"]) for sh in un], "plies": [int(sh["total_plies"]) for sh in un]}

        copy_stream = torch.cuda.Stream(device=dev)
        pinned = {}                                 # (slot, bytes) -> pinned host buffer, reused from step to step
        in_flight = []                              # device tensors whose copy to the host may still be running

        def to_host_async(dev_tensors, slot0):
            """Device shards -> pinned host memory on a SIDE stream: the copy of step k runs under the games of
            step k + 1; the timed region ends with copy_stream.synchronize(), so every byte has arrived inside the bracket."""
            copy_stream.wait_stream(torch.cuda.current_stream(dev))
            outs = []
            with torch.cuda.stream(copy_stream):
                for i, t in enumerate(dev_tensors):
                    key = (slot0 + i, t.numel())
                    if key not in pinned:
                        pinned[key] = torch.empty(t.numel(), dtype=torch.uint8, pin_memory=True)
                    pinned[key].copy_(t, non_blocking=True)
                    outs.append(pinned[key])
            in_flight.append(dev_tensors)
            if len(in_flight) > 2 * max(1, pops):
                del in_flight[0]
            return outs

        def emit_records(en, slot):
            """The finished games' compact records, packed ON THE DEVICE from the engine's record buffers (records_kernel.h).
            N > 1: the packed shards are gathered to rank 0 by the collective backend on the tensors where they lie (RCCL: device
            memory, no host bounce - SURVEY section 8e).  Rank 0 (every rank at N = 1) then copies the bytes to the host - what a
            worker would write out - on a side stream; they are DECODED after the timed region (verify_last_shards)."""
            plies, _, nbytes = en.records_info()
            shard = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            en.pack_records_into(shard.data_ptr(), nbytes)
            en.finish()
            shard_bytes[0] = nbytes
            if dist is None:
                last_shards[:] = to_host_async([shard], slot)
                return plies
            if comm_device.type != "cuda":          # gloo rehearsal: host tensors all the way
                shards = gather_packed(shard.cpu(), nbytes, dist, unpack=False)
                if shards is not None:
                    last_shards[:] = shards
                return plies
            shards = gather_packed(shard, nbytes, dist, unpack=False, to_host=False)
            if shards is not None:
                last_shards[:] = to_host_async(shards, 64 * slot)
            return plies

        def play(en, n_games):
            en.begin(n_games)
            done = 0
            while done < n_games:
                done, _ = en.step(args.rounds_per_call)

        def one_step():
            if pops == 1:
                play(engines[0], games)
            else:                                       # one host thread per population (the C calls release the GIL)
                import threading
                ths = [threading.Thread(target=play, args=(engines[k], games // pops)) for k in range(pops)]
                for t in ths:
                    t.start()
                for t in ths:
                    t.join()
            plies = 0
            tg = time.perf_counter()
            for slot, en in enumerate(engines):
                plies += emit_records(en, slot)
            gather_s[0] += time.perf_counter() - tg
            return plies

        # untimed primer: first-touch of the arenas
        for en in engines:
            en.begin(games // pops)
            en.step(4)
        barrier()
        for _ in range(warmup):
            one_step()

        def all_stats():
            tot = {}
            for en in engines:
                for k, v in en.stats().items():
                    if isinstance(v, (int, float)):
                        tot[k] = max(tot.get(k, 0), v) if k in ("max_nodes_in_arena", "cyc_max_slot_launch") else tot.get(k, 0) + v
            return tot

        st0 = all_stats()
        lib.sprl_profile_busy_reset()
        gather_s[0] = 0.0
        barrier()
        t0 = time.perf_counter()
        plies = 0
        for _ in range(steps):
            plies += one_step()
        copy_stream.synchronize()                   # the last records have reached the host
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([elapsed], device=comm_device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        st1 = all_stats()
        # several populations: launches of one kernel overlap on different streams; busy = time with >= 1 launch executing
        tree_busy, tree_sum = E.profile_busy(lib, 0)
        conv_busy, conv_sum = E.profile_busy(lib, 1) if model_path else (None, None)
        d = {k: st1[k] - st0[k] for k in st1 if isinstance(st1[k], (int, float)) and k not in ("max_nodes_in_arena", "hbm_bytes")}
        verified = verify_last_shards()
        info = engines[0].evaluator_info() if model_path else args.model
        kinds = None
        if model_path and profile_mode == 2 and not args.no_profile:
            kinds = {}
            for en in engines:
                for k, (ms, n) in en.conv_kinds().items():
                    a = kinds.setdefault(k, [0.0, 0])
                    a[0] += ms
                    a[1] += n
        if model_path and not args.allow_lab:
            # the number is only valid on the default hand-written path: the evaluator string carries the plugin's resolved path and
            # every lab switch (SPRL_*) that was set when the model was loaded / the engine created
            for en in engines:
                ei = en.evaluator_info()
                if "hand-written gfx950 CNN" not in ei or ei.count("lab=[]") != 2:
                    raise SystemExit(f"bench.py: refusing to report a timed run on a non-default evaluator path: {ei!r} "
                                     "(unset the SPRL_* lab switches, or pass --allow-lab for an A/B measurement)")
        for en in engines:
            en.close()
        return dict(pops=pops, steps=steps, elapsed=elapsed, d=d, st1=st1, tree_busy=tree_busy, tree_sum=tree_sum,
                    conv_busy=conv_busy, conv_sum=conv_sum, t_load=t_load, gather_s=gather_s[0], shard_bytes=shard_bytes[0],
                    evaluator=info, verified=verified, profile_mode=profile_mode, kinds=kinds)

    def rooflines(M, G=G, game=args.game, blocks=None):
        """(roofline of the trunk convolution or None, roofline of the tree kernel or None) of one measure() result.
        achieved = algorithmic bytes / flops of all launches / time the kernel was executing.  One engine: that time is the sum
        of the launch durations (= launches x avg_launch_ms).  Several engines (--populations): their launches overlap on
        different HIP streams, the time is the union of the launch intervals (sprl_profile_busy), from the same HIP events."""
        d, mp = M["d"], M["pops"]
        if args.no_profile or d["kernel_ms"] <= 0:
            return None, None
        bpt, _ = algorithmic_bytes_per_traversal(d, A=G["A"], mask_bytes=G["mask_bytes"], board_bytes=G["board_bytes"],
                                                 s_in=G["planes"] * G["rows"] * G["cols"] * 4)
        tree_time_ms = M["tree_busy"] if (mp > 1 and M["tree_busy"]) else d["kernel_ms"]
        achieved = d["traversals"] * bpt / (tree_time_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "tree_kernel_traffic.json")
        if os.path.exists(tpath) and game == "othello":   # PMC passes are separate rocprofv3 runs (tools/profile_pmc.sh)
            with open(tpath) as tf:
                traffic = json.load(tf).get("hbm_bytes_per_launch")
        tree = {"bound": "hbm", "kernel": f"step_kernel ({G['label']}: select/expand/backup/re-root)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "bytes_per_traversal": bpt,
                "avg_launch_ms": d["kernel_ms"] / max(1, d["kernel_launches"]),
                "traversals_per_launch": d["traversals"] / max(1, d["kernel_launches"]),
                "launch_ms_sum": d["kernel_ms"], "busy_ms": M["tree_busy"],
                "overlap": (M["tree_sum"] / M["tree_busy"]) if M["tree_busy"] else None, "populations": mp}
        if d.get("conv_ms", 0) <= 0:
            return None, tree
        # the dominant kernel by time: the trunk convolution of the CNN (cnn_wino.hip), fp32 MFMA-bound.
        # Algorithmic work per board and launch = the Winograd-domain products the kernel must issue:
        # 4 tiles x 36 positions x 64 x 64 multiply-adds (4x fewer than the direct 3x3 convolution).
        # the tiling the kernel really uses (cnn_wino.hip: sprl_wino_nchw_tile): boards up to 8x8 F(4x4,3x3), wider boards whichever
        # of F(4x4,3x3) / F(3x3,3x3) needs fewer position-products (9x9: 3x3 tiles of 3x3 cells, 25 positions each)
        R, Cc = G["rows"], G["cols"]
        m_tile, npos = 4, 36
        if (R > 8 or Cc > 8) and ((R + 2) // 3) * ((Cc + 2) // 3) * 25 < ((R + 3) // 4) * ((Cc + 3) // 4) * 36:
            m_tile, npos = 3, 25
        tiles = ((R + m_tile - 1) // m_tile) * ((Cc + m_tile - 1) // m_tile)
        flop_per_board = 2 * tiles * npos * 64 * 64
        useful_cells = (R * Cc) / (tiles * m_tile * m_tile)     # MFMA work spent on cells of the board (the rest is tile padding)
        fill = (d["nn_evals"] / d["nn_rows"]) if d.get("nn_rows") else 1.0     # host-batched path: rows padded to a bucket are not work
        flop = d["conv_boards"] * min(1.0, fill) * flop_per_board
        conv_time_ms = M["conv_busy"] if (mp > 1 and M["conv_busy"]) else d["conv_ms"]
        tf = flop / (conv_time_ms * 1e-3) / 1e12
        ctraffic = None
        boards_per_launch = d["conv_boards"] / max(1, d["conv_launches"])
        cpath = os.path.join(ROOT, "profiles", "conv_kernel_traffic.json")
        if not os.path.exists(cpath) or game != "othello":
            cpath = os.path.join(ROOT, "profiles", f"conv_kernel_traffic_{game}.json")
        if os.path.exists(cpath):
            # PMC passes (separate rocprofv3 runs with one population, tools/profile_round.sh): HBM bytes per launch at the
            # profiled boards per launch; the kernel's traffic is proportional to the boards of a launch (0.99 x algorithmic)
            with open(cpath) as tf_:
                cj = json.load(tf_)
            ctraffic = cj.get("hbm_bytes_per_launch")
            if ctraffic and cj.get("boards_per_launch"):
                ctraffic = ctraffic * boards_per_launch / cj["boards_per_launch"]
        conv = {"bound": "mfma", "kernel": ("wino_conv64 (3x3 conv 64->64 + BN/residual/ReLU, Winograd F(4x4,3x3) on fp32 MFMA)"
                                            if G["rows"] <= 8 and G["cols"] <= 8 else
                                            "wino_conv64_nchw (any-board 3x3 conv 64->64 + BN/residual/ReLU, Winograd F(4x4,3x3) on fp32 MFMA)"),
                "achieved": tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TF,
                "traffic": ctraffic, "share_of_step_time": conv_time_ms * 1e-3 / M["elapsed"],
                "launch_ms_sum": d["conv_ms"], "busy_ms": M["conv_busy"],
                "overlap": (M["conv_sum"] / M["conv_busy"]) if M["conv_busy"] else None,
                "avg_launch_ms": d["conv_ms"] / max(1, d["conv_launches"]),
                "boards_per_launch": boards_per_launch,
                "timing": ("per_launch_events" if M.get("profile_mode", 1) == 2 else
                           f"bracket_{2 * (blocks or args.blocks)}_launches (one HIP-event pair around the trunk convolutions of a forward: "
                           "avg_launch_ms = bracket / launches, it includes the gaps between the launches)"),
                "flop_per_board": flop_per_board, "tile": f"F({m_tile}x{m_tile},3x3), {tiles} tiles x {npos} positions per board",
                "useful_cells_fraction": useful_cells, "frac_useful": tf / MFMA_F32_PEAK_TF * useful_cells,
                "direct_conv_equivalent_tflops": tf * (9.0 * m_tile * m_tile / npos) * useful_cells, "populations": mp}
        return conv, tree

    pops = args.populations if args.populations > 0 else (G.get("populations", 1) if args.model == "cnn" else 1)
    M = measure(pops, args.steps, args.warmup)
    elapsed, d, st1 = M["elapsed"], M["d"], M["st1"]
    t_load, gather_s, shard_bytes = M["t_load"], [M["gather_s"]], [M["shard_bytes"]]
    # the same kernels with nothing running beside them: a separate untimed-for-`value` pass with ONE population (one step)
    M1 = None
    if pops > 1 and not args.no_profile and not args.no_alone_pass and args.game == "othello":     # (a Go step is minutes long)
        M1 = measure(1, 1, 0, profile_mode=2)       # per-launch event pairs: kernel durations (comparable with rocprofv3's averages)

    def sample_rounds(game, pops_s, concurrent, rounds, warm, resign_threshold=0.0, resign_min_ply=0, model=None, profile_mode=1):
        """A BOUNDED sample of a configuration: `pops_s` engines (own streams and host threads when > 1) with `concurrent` resident
        games in all, `warm` search rounds untimed (first touch of the arenas, the opening rounds), then `rounds` rounds timed
        between barriers.  A move is num_traversals / max_queue rounds, so the games are between move warm / that and
        (warm + rounds) / that.  Same dict as measure()."""
        import threading
        Gs = GAMES[game]
        engines = []
        for p in range(pops_s):
            cfg = E.default_config(Gs["engine"], lib, device=local_rank, concurrent_games=concurrent // pops_s,
                                   num_traversals=Gs["traversals"] if game != args.game else args.traversals, seed=args.seed,
                                   node_cap=args.node_cap if game == args.game else 0, stream_base=stream_cursor[0],
                                   profile=profile_mode, own_stream=1 if pops_s > 1 else 0,
                                   resign_threshold=resign_threshold, resign_min_ply=resign_min_ply)
            stream_cursor[0] += 2 * (concurrent // pops_s)
            en = E.Engine(cfg, lib)
            en.set_model(model)
            en.begin(concurrent // pops_s)
            engines.append(en)

        def run(en, n):
            done = 0
            while done < n:
                k = min(args.rounds_per_call, n - done)
                en.step(k)
                done += k

        def run_all(n):
            if pops_s == 1:
                run(engines[0], n)
                return
            ths = [threading.Thread(target=run, args=(en, n)) for en in engines]
            for t in ths:
                t.start()
            for t in ths:
                t.join()

        def all_stats():
            tot = {}
            for en in engines:
                for k, v in en.stats().items():
                    if isinstance(v, (int, float)):
                        tot[k] = max(tot.get(k, 0), v) if k in ("max_nodes_in_arena", "cyc_max_slot_launch") else tot.get(k, 0) + v
            return tot

        run_all(warm)
        st0 = all_stats()
        lib.sprl_profile_busy_reset()
        barrier()
        t0 = time.perf_counter()
        run_all(rounds)
        barrier()
        elapsed = time.perf_counter() - t0
        st1 = all_stats()
        tree_busy, tree_sum = E.profile_busy(lib, 0)
        conv_busy, conv_sum = E.profile_busy(lib, 1)
        d = {k: st1[k] - st0[k] for k in st1 if k not in ("max_nodes_in_arena", "hbm_bytes")}
        info = engines[0].evaluator_info()
        for en in engines:
            ei = en.evaluator_info()
            if not args.allow_lab and ("hand-written gfx950 CNN" not in ei or ei.count("lab=[]") != 2):
                raise SystemExit(f"bench.py: refusing to report a sample on a non-default evaluator path: {ei!r}")
            en.close()
        return dict(pops=pops_s, steps=0, elapsed=elapsed, d=d, st1=st1, tree_busy=tree_busy if pops_s > 1 else None,
                    tree_sum=tree_sum if pops_s > 1 else None, conv_busy=conv_busy if pops_s > 1 else None,
                    conv_sum=conv_sum if pops_s > 1 else None, evaluator=info, profile_mode=profile_mode)

    MA = None
    if pops > 1 and not args.no_profile and not args.no_alone_pass and args.game != "othello" and model_path:
        n_alone = 1500 if args.game == "go9" else 600
        MA = sample_rounds(args.game, 1, args.concurrent // pops, n_alone, n_alone, args.resign_threshold, args.resign_min_ply,
                           model=model_path, profile_mode=2)

    # BASELINE configs 4 / 5 inside the DEFAULT run: after the Othello timed region, bounded two-population samples of
    # Go 9x9 and Go 19x19 at their stated budgets (1600 iterations/move, batch 16 / queue 8, the 6 x 64 network, 4096 / 2048
    # resident games), a fixed number of search rounds each - evaluations/s, traversals/s, plies/s and the rooflines of their
    # dominant kernels land in the driver-run record, not only in builder-run files
    secondary = None
    if args.game == "othello" and args.model == "cnn" and world == 1 and not args.no_secondary and not args.no_profile:
        secondary = {}
        for sg, warm_r, timed_r in (("go9", 200, 1000), ("go19", 100, 400)):
            Gs = GAMES[sg]
            t_s0 = time.perf_counter()
            mp_s = trace_to_file(make_network(Gs["engine"], Gs["blocks"], args.channels, seed=0),
                                 os.path.join(tmpdir, f"traced_{sg}.pt"), Gs["engine"])
            # resign: the extension cannot fire before ply 60 (go19's resign_min_ply) and a bounded sample from the start position
            # covers the first moves only, so this sample IS the resign-off regime (reference behaviour)
            Ms = sample_rounds(sg, 2, Gs["concurrent"], timed_r, warm_r, model=mp_s)
            cs, ts = rooflines(Ms, Gs, sg, Gs["blocks"])
            ds = Ms["d"]
            rounds_per_move = Gs["traversals"] / int(Gs["bq"].split("/")[1])
            secondary[sg] = {
                "workload": f"{Gs['label']}, {Gs['traversals']} UCT iters/move, {Gs['concurrent']} concurrent games, batch {Gs['bq'].split('/')[0]}/queue "
                            f"{Gs['bq'].split('/')[1]}, {Gs['sym']}, {Gs['noise']}, traced CNN {Gs['blocks']}x{args.channels} fp32, 2 populations, resign off",
                "sample": f"{warm_r} untimed + {timed_r} timed search rounds per population from the start position (moves "
                          f"{warm_r / rounds_per_move:.1f} .. {(warm_r + timed_r) / rounds_per_move:.1f} of every game)",
                "seconds": Ms["elapsed"], "wall_seconds_incl_setup": time.perf_counter() - t_s0,
                "nn_evals_per_sec": ds["nn_evals"] / Ms["elapsed"], "traversals_per_sec": ds["traversals"] / Ms["elapsed"],
                "expansions_per_sec": ds["expansions"] / Ms["elapsed"], "plies_per_sec": ds["plies"] / Ms["elapsed"],
                "moves_per_game_per_sec": ds["plies"] / Ms["elapsed"] / Gs["concurrent"],
                "hbm_gib": Ms["st1"]["hbm_bytes"] / 2**30, "max_nodes_in_arena": Ms["st1"]["max_nodes_in_arena"],
                "nodes_recycled_share": ds["nodes_recycled"] / max(1, ds["nodes_created"]), "compactions": ds["compactions"],
                "evaluator": Ms["evaluator"], "roofline": cs, "roofline_tree": ts}

    # what the collective backend really saw: world size, backend name and every rank's device, gathered from the ranks themselves
    me = {"rank": rank, "local_rank": local_rank, "device": f"cuda:{local_rank}", "name": torch.cuda.get_device_name(local_rank),
          "pid": os.getpid()}
    if dist is not None:
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, me)
        dist_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks": ranks_info}
    else:
        dist_info = {"world_size": 1, "backend": None, "ranks": [me]}
    if rank == 0:
        total_games = games * args.steps * world
        bpt, shape = algorithmic_bytes_per_traversal(d, A=G["A"], mask_bytes=G["mask_bytes"], board_bytes=G["board_bytes"],
                                                     s_in=G["planes"] * G["rows"] * G["cols"] * 4)
        backend_name = "RCCL over xGMI (nccl)" if args.backend == "nccl" else "gloo (rehearsal)"
        out = {
            "metric": f"self-play games/sec, {G['label']} @ {args.traversals} iters/move",
            "value": total_games / elapsed,
            "unit": "games/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / max(1, args.steps),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (start-position self-play, random-init weights)",
            "config": {"workload": f"{G['label']}, {args.traversals} UCT iters/move, {args.concurrent} concurrent games/GPU, "
                                   f"{games} games/GPU/step, batch {G['bq'].split('/')[0]}/queue {G['bq'].split('/')[1]}, {G['sym']}, {G['noise']}" +
                                   (f", resign threshold {args.resign_threshold}" if args.resign_threshold > 0 else "") +
                                   (f", {pops} populations on {pops} HIP streams" if pops > 1 else ""),
                       "evaluator": (f"traced CNN {args.blocks}x{args.channels} fp32, " + M["evaluator"] if model_path else args.model),
                       "parallelism": (f"game-sharded x{world}, records packed on the device, one gather per step over {backend_name}"
                                       if world > 1 else "1 GPU")},
            "distributed": dict(dist_info, gathered_last_step=M["verified"]),
            "expansions_per_sec": d["expansions"] * world / elapsed,
            "traversals_per_sec": d["traversals"] * world / elapsed,
            "nn_evals_per_sec": d["nn_evals"] * world / elapsed,
            "plies_per_game": d["plies"] / max(1, d["games"]),
            "search_shape": shape,
            "rank0": {"kernel_ms": d["kernel_ms"], "nn_ms": d["nn_ms"] or None, "kernel_launches": d["kernel_launches"],
                      "rounds": d["rounds"], "nn_batches": d["nn_batches"], "nn_rows": d["nn_rows"],
                      "nn_fill": (d["nn_evals"] / d["nn_rows"]) if d["nn_rows"] else None, "hbm_gib": st1["hbm_bytes"] / 2**30, "model_load_and_warmup_s": t_load,
                      "max_nodes_in_arena": st1["max_nodes_in_arena"], "compactions": d["compactions"],
                      "nodes_recycled_share": d["nodes_recycled"] / max(1, d["nodes_created"]),
                      "record_gather_ms": 1000.0 * gather_s[0] / max(1, args.steps), "record_shard_bytes": shard_bytes[0]},
        }
        if d.get("cyc_total", 0) > 0:
            out["phase_cycles_share"] = {k[4:]: d[k] / d["cyc_total"] for k in d if k.startswith("cyc_") and k not in ("cyc_total", "cyc_max_slot_launch")}
            out["phase_cycles_share"]["max_cycles_one_slot_launch"] = st1["cyc_max_slot_launch"]
            out["phase_cycles_share"]["noise_cycles_per_move"] = d["cyc_noise"] / max(1, d["plies"])
            out["phase_cycles_share"]["per_level_cycles"] = {k: d["cyc_lvl_" + k] / max(1, d["levels"]) for k in ("wait", "pick", "desc")}
            out["phase_cycles_share"]["total_cycles_per_slot_launch"] = d["cyc_total"] / max(1, d["kernel_launches"]) / args.concurrent
        conv_rl, tree_rl = rooflines(M)
        if conv_rl is not None:
            out["roofline"], out["roofline_tree"] = conv_rl, tree_rl
        elif tree_rl is not None:
            out["roofline"] = tree_rl
        if M1 is not None:
            c1, t1 = rooflines(M1)
            if conv_rl is not None and c1 is not None:
                # the kernel's own duration (one event pair per launch, nothing else on the GPU), measured after the timed region:
                # the figure that is comparable with the rocprofv3 --kernel-trace average under profiles/
                out["roofline"]["kernel_only"] = {"avg_launch_ms": c1["avg_launch_ms"], "boards_per_launch": c1["boards_per_launch"],
                                                  "achieved": c1["achieved"], "frac": c1["frac"], "timing": c1["timing"],
                                                  "from": "one_population_pass"}
                if M1.get("kinds"):
                    # the four launches of a forward are different kernels since round 4: the first carries the stem, the last the
                    # head convolutions and the FC layers.  Each does ONE trunk convolution's flops; `plain` and `residual` are the
                    # bare convolution (the figure comparable with earlier rounds' kernel-alone fraction)
                    bv = {}
                    for k, (ms, n) in M1["kinds"].items():
                        if n > 0:
                            avg = ms / n
                            tfk = c1["boards_per_launch"] * c1["flop_per_board"] / (avg * 1e-3) / 1e12
                            bv[k] = {"launches": n, "avg_launch_ms": avg, "achieved": tfk, "frac": tfk / MFMA_F32_PEAK_TF}
                    out["roofline"]["kernel_only"]["by_variant"] = bv
                    bare = [bv[k] for k in ("plain", "residual") if k in bv]
                    if bare:
                        nb_ = sum(b["launches"] for b in bare)
                        avg_b = sum(b["avg_launch_ms"] * b["launches"] for b in bare) / nb_
                        tfb = c1["boards_per_launch"] * c1["flop_per_board"] / (avg_b * 1e-3) / 1e12
                        out["roofline"]["kernel_only"]["bare_convolution"] = {"avg_launch_ms": avg_b, "achieved": tfb,
                                                                              "frac": tfb / MFMA_F32_PEAK_TF}
            if tree_rl is not None and t1 is not None:
                out["roofline_tree"]["kernel_only"] = {"avg_launch_ms": t1["avg_launch_ms"], "achieved": t1["achieved"], "frac": t1["frac"],
                                                       "from": "one_population_pass"}
            out["one_population_pass"] = {
                "what": "the same workload with ONE population (no kernel of another stream beside it), one step, after the timed "
                        "region: per-launch figures comparable to the rocprofv3 averages under profiles/",
                "games_per_sec": games * world / M1["elapsed"], "roofline": c1, "roofline_tree": t1}
        if secondary is not None:
            out["secondary"] = secondary
        if MA is not None:
            ca, ta = rooflines(MA)
            out["one_population_sample"] = {
                "what": "the same kernels with ONE population (no kernel of another stream beside them, so a launch does not wait for "
                        "CUs held by the other population's convolutions): a bounded sample after the timed region, one engine with "
                        "the resident games of one population, the rounds given below - a whole extra step would take minutes",
                "rounds": MA["d"]["rounds"], "rounds_before": MA["st1"]["rounds"] - MA["d"]["rounds"],
                "seconds": MA["elapsed"], "roofline": ca, "roofline_tree": ta}
        if world == 1 and not args.no_cpu_baseline and args.game == "othello":
            mp_model = model_path
            if mp_model is None:
                mp_model = trace_to_file(make_network("othello", args.blocks, args.channels, seed=0),
                                         os.path.join(tmpdir, "traced_cpu.pt"), "othello")
            try:
                cb = cpu_baseline(mp_model, args.traversals)
                if cb.get("value"):
                    cb["gpu_over_host_cores"] = out["value"] / cb["value"]           # 1 MI355X vs all usable host cores
                if cb.get("reference_deployment_games_per_sec"):
                    cb["gpu_over_reference_deployment"] = out["value"] / cb["reference_deployment_games_per_sec"]
                out["cpu_baseline"] = cb
            except Exception as exc:  # the baseline is reported, never the target: do not lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "games/s", "cores": 0, "kind": "reference",
                                       "sample": f"failed: {exc}"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
