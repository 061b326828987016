"""GPU self-play worker honouring the reference workers' process + filesystem contract - the Python face (ctypes) of the
native worker `sprl_amd/sprl_worker` (sprl_amd/csrc/worker_main.cpp), same options and file layout.

Mirrors `OTHWorker <task_id> <num_tasks>` / `C4Worker` (cpp/src/OTHWorker.cpp:31-70, cpp/src/C4Worker.cpp) and
`runWorker` (cpp/src/selfplay/GridWorker.hpp:84-198):

* reads   data/models/<run>/traced_<run>_iteration_<i-1>.pt          (GridWorker.hpp:36-55; iteration 0 = built-in
          initial evaluator, "random")
* writes  data/games/<run>/<group>/<task_id>/<run>_iteration_<i>_{states,distributions,outcomes}.npy
          (GridWorker.hpp:116,173-196; OTHWorker.cpp:44-49), group = task_id // (num_tasks // num_groups)
* iteration 0 uses the init budgets, later iterations the steady-state ones (GridWorker.hpp:118-121; the reference's
  shadowing bug Q2 is NOT reproduced: the steady-state values are the worker constants).

One MI355X replaces many CPU tasks: `--cover K` makes this process play the games of task ids
[task_id, task_id + K) concurrently on the GPU and write each task's files into that task's own directory, so the
unmodified Python controller (scripts/othello_controller.py:66-125) sees exactly the files it polls for.
Files appear atomically (temp + rename), outcomes last.
"""
import argparse
import os
import sys
import time
from dataclasses import dataclass

from . import engine as E

MODEL_PATH_WAIT_INTERVAL = 30  # GridWorker.hpp:23


@dataclass
class WorkerConstants:
    game: str
    run_name: str
    num_groups: int
    num_worker_tasks: int
    num_iters: int
    init_games: int
    init_traversals: int
    init_max_batch: int
    init_max_queue: int
    games: int
    traversals: int
    max_batch: int
    max_queue: int
    dir_eps: float
    dir_alpha: float


# OTHWorker.cpp:12-32, C4Worker.cpp:11-31
REFERENCE_WORKERS = {
    "othello": WorkerConstants("othello", "orangutan_alpha", 4, 384, 50, 3, 131072, 1, 1, 3, 8192, 8, 4, 0.25, 0.3),
    "connect_four": WorkerConstants("connect_four", "c4_test", 1, 1, 25, 10, 2048, 1, 1, 5, 512, 8, 4, 0.25, 0.5),
    # GoWorker.cpp:11-29 (Go as compiled by the reference: 7x7)
    "go7": WorkerConstants("go7", "panda_alpha", 4, 384, 100, 3, 262144, 1, 1, 3, 32768, 16, 8, 0.25, 0.2),
    # BASELINE configs 4 / 5: the Go worker's constants at 9x9 / 19x19 with 1600 iterations per move (the reference compiles 7x7 only)
    "go9": WorkerConstants("go9", "panda_9x9", 4, 384, 100, 3, 1600, 16, 8, 3, 1600, 16, 8, 0.25, 0.2),
    "go19": WorkerConstants("go19", "panda_19x19", 4, 384, 100, 3, 1600, 16, 8, 3, 1600, 16, 8, 0.25, 0.2),
}


def model_path_for(iteration, run_name, root="."):
    """waitModelPath's path scheme (GridWorker.hpp:35-43); iteration -1 -> "random"."""
    if iteration == -1:
        return "random"
    return os.path.join(root, "data", "models", run_name, f"traced_{run_name}_iteration_{iteration}.pt")


def wait_model_path(iteration, run_name, root=".", poll_seconds=MODEL_PATH_WAIT_INTERVAL, settle_seconds=5, log=print):
    path = model_path_for(iteration, run_name, root)
    if path == "random":
        return path
    while not os.path.exists(path):
        log(f"Spinning on traced model from iteration {iteration}...")
        time.sleep(poll_seconds)
    time.sleep(settle_seconds)          # GridWorker.hpp:52
    return path


def save_dir_for(consts, task_id, root="."):
    group = task_id // (consts.num_worker_tasks // consts.num_groups)       # OTHWorker.cpp:44
    return os.path.join(root, "data", "games", consts.run_name, str(group), str(task_id))


def write_task_files(lib, rec, games_per_task, dirs, run_name, iteration, fmt="v1"):
    """Each covered task's games go to that task's directory through the C ABI's own writers (sprl_records_slice +
    sprl_write_npy: the reference's header bytes, temp file + rename, outcomes last; or the compact v2 file)."""
    import ctypes as C
    for k, d in enumerate(dirs):
        part = E.Records()
        rc = lib.sprl_records_slice(C.byref(rec._rec), k * games_per_task, games_per_task, C.byref(part))
        if rc:
            raise E.SprlError(rc, lib.sprl_last_error().decode())
        prefix = os.path.join(d, f"{run_name}_iteration_{iteration}")
        rc = lib.sprl_write_v2(os.fsencode(prefix + ".sprl2"), C.byref(part)) if fmt == "v2" else \
            lib.sprl_write_npy(os.fsencode(prefix), C.byref(part))
        lib.sprl_records_free(C.byref(part))
        if rc:
            raise E.SprlError(rc, lib.sprl_last_error().decode())


def run_worker(consts, task_id, cover=1, num_iters=None, root=".", seed=None, concurrent_games=None, lib=None,
               model_for_iteration=None, log=print, device=0, resign_threshold=0.0, resign_min_ply=0, fmt="v1", populations=1):
    """The worker loop (GridWorker.hpp:111-197) for task ids [task_id, task_id + cover).  populations > 1 (as the native
    worker's --populations): the covered tasks are split into that many ranges, each with its own engine on a private HIP
    stream and its own host thread (the C calls release the GIL), RNG streams from disjoint ranges per population."""
    lib = lib or E.load_library()
    if populations > 1:
        import threading
        if populations > cover:
            raise ValueError("more populations than covered tasks")
        seed = int(time.time_ns() & 0x7FFFFFFFFFFF) | 1 if seed is None else seed
        errors = []

        def one(p):
            first, last = cover * p // populations, cover * (p + 1) // populations
            try:
                run_worker(consts, task_id + first, last - first, num_iters, root, seed,
                           None if concurrent_games is None else -(-concurrent_games // populations), lib, model_for_iteration,
                           log if p == 0 else (lambda *a, **k: None), device, resign_threshold, resign_min_ply, fmt,
                           populations=-(p + 1))
            except Exception as exc:  # noqa: BLE001 - reported to the caller below
                errors.append(exc)

        ths = [threading.Thread(target=one, args=(p,)) for p in range(populations)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errors:
            raise errors[0]
        return
    pop = -populations - 1 if populations < 0 else -1                 # (internal: this call is population `pop` of several)
    num_iters = consts.num_iters if num_iters is None else num_iters
    seed = int(time.time_ns() & 0x7FFFFFFFFFFF) | 1 if seed is None else seed   # reference: random_device (Q3)
    dirs = [save_dir_for(consts, task_id + k, root) for k in range(cover)]
    for d in dirs:
        existed = os.path.isdir(d)
        os.makedirs(d, exist_ok=True)
        log(("Directory already exists: " if existed else "Created directory: ") + d)
    next_stream = 1 + max(pop, 0) * (1 << 27)
    eng, eng_sig = None, None
    for it in range(num_iters):
        log(f"Starting iteration {it}...")
        model = (model_for_iteration(it) if model_for_iteration else wait_model_path(it - 1, consts.run_name, root, log=log))
        games = consts.init_games if it == 0 else consts.games
        trav = consts.init_traversals if it == 0 else consts.traversals
        mb = consts.init_max_batch if it == 0 else consts.max_batch
        mq = consts.init_max_queue if it == 0 else consts.max_queue
        total = games * cover
        sig = (trav, mb, mq, min(concurrent_games or total, total))
        if eng is None or sig != eng_sig:          # one engine serves every steady-state iteration; only its model changes
            if eng is not None:
                eng.close()
            cfg = E.default_config(consts.game, lib, device=device, concurrent_games=sig[3], num_traversals=trav, max_batch=mb,
                                   max_queue=mq, dir_eps=consts.dir_eps, dir_alpha=consts.dir_alpha, seed=seed,
                                   stream_base=next_stream, resign_threshold=resign_threshold, resign_min_ply=resign_min_ply,
                                   own_stream=1 if pop >= 0 else 0)
            eng, eng_sig = E.Engine(cfg, lib), sig
        next_stream += total                        # (a kept engine continues its stream numbering itself)
        log("Using initial network..." if model == "random" else "Using traced PyTorch network...")
        eng.set_model(model)
        rec = eng.run(total)
        write_task_files(lib, rec, games, dirs, consts.run_name, it, fmt)
        log(f"{total} games played, {rec.num_samples} states collected.")
        rec.close()
    if eng is not None:
        eng.close()


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("game", choices=sorted(REFERENCE_WORKERS))
    ap.add_argument("task_id", type=int)
    ap.add_argument("num_tasks", type=int)
    ap.add_argument("--cover", type=int, default=1, help="number of consecutive task ids this GPU stands in for")
    ap.add_argument("--run-name")
    ap.add_argument("--num-iters", type=int)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--seed", type=int)
    ap.add_argument("--resign-threshold", type=float, default=0.0,
                    help="extension, not in the reference (default off): the side to move resigns when the mean value of its "
                         "decision node after the search is below -threshold")
    ap.add_argument("--resign-min-ply", type=int, default=0)
    ap.add_argument("--format", default="v1", choices=["v1", "v2"], help="v1 = the reference's .npy triple; v2 = compact .sprl2")
    ap.add_argument("--populations", type=int, default=1,
                    help="split the covered tasks over this many engines on private HIP streams and host threads")
    try:
        args = ap.parse_args(argv)
    except SystemExit:
        print("Usage: python -m sprl_amd.worker <game> <task_id> <num_tasks>", file=sys.stderr)   # OTHWorker.cpp:34-37
        return 1
    consts = REFERENCE_WORKERS[args.game]
    if args.num_tasks != consts.num_worker_tasks:
        print(f"num_tasks must be {consts.num_worker_tasks} (OTHWorker.cpp:42)", file=sys.stderr)
        return 1
    if args.run_name:
        consts = WorkerConstants(**{**consts.__dict__, "run_name": args.run_name})
    print(f"Task {args.task_id} of {args.num_tasks}, covering {args.cover} task(s).")
    run_worker(consts, args.task_id, cover=args.cover, num_iters=args.num_iters, device=args.device, seed=args.seed,
               resign_threshold=args.resign_threshold, resign_min_ply=args.resign_min_ply, fmt=args.format,
               populations=args.populations)
    return 0


if __name__ == "__main__":
    sys.exit(main())
