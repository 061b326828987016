"""Host logic above the C ABI, on the CPU SIMT-emulator build: the worker's file contract and the
game-sharded multi-process path (gloo, world size 2)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import pyoracle as po
from sprl_amd import engine as E
from sprl_amd import worker as W
import parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    return E.load_library(os.path.join(EMU_DIR, "libsprl_emu.so"))


def test_reference_constants_and_paths():
    o = W.REFERENCE_WORKERS["othello"]
    assert (o.num_worker_tasks, o.num_groups, o.games, o.traversals, o.max_batch, o.max_queue) == (384, 4, 3, 8192, 8, 4)
    assert (o.init_traversals, o.init_max_batch, o.init_max_queue) == (131072, 1, 1)
    assert W.save_dir_for(o, 100, "r") == os.path.join("r", "data", "games", o.run_name, "1", "100")   # group = 100 // 96
    assert W.model_path_for(-1, "x") == "random"
    assert W.model_path_for(3, "x", "r") == os.path.join("r", "data", "models", "x", "traced_x_iteration_3.pt")
    assert W.main(["othello", "0", "12"]) == 1          # wrong num_tasks, like the reference's assert
    assert W.main(["othello"]) == 1                     # usage error -> exit code 1 (OTHWorker.cpp:34-37)


def test_worker_covers_several_tasks_with_reference_file_layout(emu, tmp_path):
    consts = W.WorkerConstants("connect_four", "tinyrun", 1, 4, 2, 2, 30, 8, 4, 1, 20, 8, 4, 0.25, 0.5)
    logs = []
    W.run_worker(consts, task_id=1, cover=2, root=str(tmp_path), seed=77, lib=emu,
                 model_for_iteration=lambda it: "random", log=logs.append)
    assert any(line.startswith("Starting iteration 1") for line in logs)
    for task in (1, 2):
        d = tmp_path / "data" / "games" / "tinyrun" / "0" / str(task)
        for it, games in ((0, 2), (1, 1)):
            s = np.load(d / f"tinyrun_iteration_{it}_states.npy")
            p = np.load(d / f"tinyrun_iteration_{it}_distributions.npy")
            o = np.load(d / f"tinyrun_iteration_{it}_outcomes.npy")
            assert s.dtype == np.float32 and s.shape[1:] == (3, 6, 7) and p.shape == (s.shape[0], 7)
            assert o.shape == (s.shape[0],) and s.shape[0] % 2 == 0          # nsym = 2 samples per ply
            raw = open(d / f"tinyrun_iteration_{it}_outcomes.npy", "rb").read()
            assert raw[:8] == b"\x93NUMPY\x01\x00" and (10 + raw[8] + 256 * raw[9]) % 16 == 0
        assert not list(d.glob("*.tmp"))
    # the two tasks' iteration-0 files are exactly games 0-1 and 2-3 of one run with streams 1..4
    cfg = po.make_config(po.GAME_C4, 30, math_mode=po.MATH_PORTABLE)
    ora = po.selfplay(cfg, 4, 77, 1, True)
    split = ora["offsets"][2]
    d1 = np.load(tmp_path / "data/games/tinyrun/0/1/tinyrun_iteration_0_distributions.npy")
    d2 = np.load(tmp_path / "data/games/tinyrun/0/2/tinyrun_iteration_0_distributions.npy")
    assert (d1.view(np.uint32) == ora["dists"][:split].view(np.uint32)).all()
    assert (d2.view(np.uint32) == ora["dists"][split:].view(np.uint32)).all()


def test_worker_populations_match_the_native_worker_scheme(emu, tmp_path):
    """run_worker(populations=2): tasks split over two engines / host threads, population p on RNG streams from 1 + p * 2^27
    (the same scheme as sprl_worker --populations, tests below)."""
    consts = W.WorkerConstants("connect_four", "poprun", 2, 4, 1, 2, 30, 8, 4, 1, 20, 8, 4, 0.25, 0.5)
    W.run_worker(consts, task_id=0, cover=4, root=str(tmp_path), seed=77, lib=emu, model_for_iteration=lambda it: "random",
                 log=lambda *a: None, populations=2)
    cfg = po.make_config(po.GAME_C4, 30, math_mode=po.MATH_PORTABLE)
    for pop, tasks in ((0, (0, 1)), (1, (2, 3))):
        ora = po.selfplay(cfg, 4, 77, 1 + pop * (1 << 27), True)
        split = ora["offsets"][2]
        for k, task in enumerate(tasks):
            got = np.load(tmp_path / f"data/games/poprun/{task // 2}/{task}/poprun_iteration_0_distributions.npy")
            want = ora["dists"][:split] if k == 0 else ora["dists"][split:]
            assert (got.view(np.uint32) == want.view(np.uint32)).all(), (pop, task)
    with pytest.raises(ValueError):
        W.run_worker(consts, task_id=0, cover=1, root=str(tmp_path), seed=1, lib=emu, populations=2)


@pytest.mark.parametrize("game,kw", [("othello", dict(num_traversals=16)), ("c4", dict(num_traversals=16)),
                                     ("go", dict(num_traversals=20)),
                                     ("go9", dict(num_traversals=20)),        # 81 cells: 2 words per bit set
                                     ("go19", dict(num_traversals=12))])      # 361 cells: 6 words
def test_pack_unpack_roundtrip(emu, game, kw):
    """The wire format of the record gather carries every board size (the 64-cell limit broke Go 9x9 / 19x19)."""
    from sprl_amd.distributed import pack_records, unpack_records, section_offsets
    ngames = 1 if game == "go19" else 2
    cfg = E.default_config(game, emu, concurrent_games=ngames, seed=9, **kw)
    eng = E.Engine(cfg, emu)
    eng.set_model("random")
    rec = eng.run(ngames)
    eng.close()
    buf = pack_records(rec)
    u = unpack_records(buf)
    assert u["cells"] == rec.cells and u["words"] == (rec.cells + 63) // 64 and u["history"] == rec.history
    assert u["boards"].shape == rec.boards.shape and (u["boards"] == rec.boards).all() and (u["movers"] == rec.movers).all()
    assert (u["pdfs"].view(np.uint32) == rec.pdfs.view(np.uint32)).all()
    assert (u["ply_offset"] == rec.ply_offset).all() and (u["winners"] == rec.winners).all()
    off = section_offsets(rec.num_games, rec.total_plies, rec.actions, u["words"])
    assert buf.size == off["total"] and all(v % 16 == 0 for v in off.values())
    with pytest.raises(ValueError):
        unpack_records(buf[:-16])


RANK_SCRIPT = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from sprl_amd import engine as E
from sprl_amd.distributed import gather_packed, gather_records, pack_records, unpack_records
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lib = E.load_library({emu!r})
games = 3
cfg = E.default_config({game!r}, lib, concurrent_games=2, num_traversals={trav}, seed=21, stream_base=1 + rank * games)
eng = E.Engine(cfg, lib); eng.set_model("random")
rec = eng.run(games)
shards = gather_records(rec, dist)
# the deferred-decoding form bench.py uses inside its timed bracket: raw host bytes now, unpack_records later
payload = pack_records(rec)
raw = gather_packed(torch.from_numpy(payload), payload.size, dist, unpack=False)
if rank == 0:
    assert len(raw) == world and all(r.dtype == torch.uint8 and r.device.type == "cpu" for r in raw)
    for r, sh in zip(raw, shards):
        late = unpack_records(r.numpy())
        assert int(r[:16].view(torch.int64)[1]) == sh["total_plies"] == late["total_plies"]
        assert (late["pdfs"].view(np.uint32) == sh["pdfs"].view(np.uint32)).all() and (late["boards"] == sh["boards"]).all()
else:
    assert raw is None
if rank == 0:
    np.savez({out!r}, **{{f"r{{i}}_{{k}}": v for i, sh in enumerate(shards) for k, v in sh.items() if isinstance(v, np.ndarray)}})
dist.barrier(); dist.destroy_process_group()
"""


@pytest.mark.parametrize("game,trav,port", [("c4", 24, 29533), ("othello", 16, 29534), ("go9", 20, 29535)])
def test_two_rank_game_sharding_gloo(emu, tmp_path, game, trav, port):
    """world_size 2 on CPU: each rank plays its own shard (disjoint RNG streams); rank 0 ends up with both
    shards, which together equal oracle games 1..6 in order."""
    out = str(tmp_path / "gathered.npz")
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT.format(root=ROOT, emu=os.path.join(EMU_DIR, "libsprl_emu.so"), out=out, game=game, trav=trav))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    assert [p.wait(timeout=240) for p in procs] == [0, 0]
    z = np.load(out)
    ogame = {"c4": po.GAME_C4, "othello": po.GAME_OTHELLO, "go9": po.GAME_GO9}[game]
    cfg = po.make_config(ogame, trav, math_mode=po.MATH_PORTABLE)
    ora = po.selfplay(cfg, 6, 21, 1, True)
    nsym = 2 if game == "c4" else 8
    pdfs = np.concatenate([z["r0_pdfs"], z["r1_pdfs"]])
    # oracle rows are symmetrised x nsym; symmetry 0 (identity) rows are the compact ones
    assert (pdfs.view(np.uint32) == ora["dists"][0::nsym].view(np.uint32)).all()
    cells = z["r0_boards"].shape[1]
    assert (np.concatenate([z["r0_boards"], z["r1_boards"]]) == ora["boards"][0::nsym][:, :cells]).all()
    assert (np.concatenate([z["r0_winners"], z["r1_winners"]]).size == 6)


@pytest.mark.parametrize("game,kw", [("othello", dict(num_traversals=16)), ("c4", dict(num_traversals=16)), ("go", dict(num_traversals=20)),
                                     ("go9", dict(num_traversals=20))])
def test_device_side_pack_and_expand_equal_host_paths(emu, game, kw):
    """The record kernels (records_kernel.h: same source on the device and here on the emulator) against the host paths:
    packed bytes == distributed.pack_records(collect()), expanded samples == sprl_records_expand (the reference's arrays)."""
    from sprl_amd.distributed import pack_records
    cfg = E.default_config(game, emu, concurrent_games=3, seed=4, **kw)
    eng = E.Engine(cfg, emu)
    eng.set_model("random")
    eng.begin(3)
    with pytest.raises(E.SprlError):
        eng.records_info()                                  # nothing has finished yet
    done = 0
    while done < 3:
        done, _ = eng.step(64)
    plies, samples, nbytes = eng.records_info()
    packed = np.full(nbytes + 16, 0xAB, np.uint8)
    base = (-packed.ctypes.data) % 16                       # a 16-byte aligned window inside the buffer
    eng.pack_records_into(packed.ctypes.data + base, nbytes)
    rec0 = eng  # keep the engine running for the expansion below
    A = emu.sprl_records_num_samples                       # noqa: F841 (symbol exists)
    states = np.full((samples, 2 * (8 if game.startswith("go") else 1) + 1, cfg_rows(game), cfg_rows(game, cols=True)), np.nan, np.float32)
    dists = np.full((samples, {"othello": 65, "c4": 7, "go": 50, "go9": 82}[game]), np.nan, np.float32)
    outs = np.full(samples, np.nan, np.float32)
    eng.expand_records_into(states.ctypes.data, dists.ctypes.data, outs.ctypes.data, samples)
    rec = eng.collect()                                     # the host path of the same run
    assert rec.total_plies == plies and rec.num_samples == samples
    want = pack_records(rec)
    assert want.size == nbytes and (packed[base:base + nbytes] == want).all()
    assert (packed[base + nbytes:] == 0xAB).all()
    s1, d1, o1 = rec.expand()
    assert (states == s1).all() and (dists.view(np.uint32) == d1.view(np.uint32)).all() and (outs == o1).all()
    eng.close()


def cfg_rows(game, cols=False):
    return {"othello": (8, 8), "c4": (6, 7), "go": (7, 7), "go9": (9, 9)}[game][1 if cols else 0]


def _native_worker():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    return os.path.join(EMU_DIR, "sprl_worker_emu")


def test_native_worker_process_contract(tmp_path):
    """The native worker (sprl_amd/csrc/worker_main.cpp, here linked against the emulator library): reference argv and exit
    codes (OTHWorker.cpp:34-42), directory scheme, init vs steady-state budgets, the model-file rendez-vous of
    GridWorker.hpp:35-55 (the file appears while the worker is polling), one engine across steady-state iterations,
    --cover with per-task files whose bytes equal the oracle's games, and the v2 compact format."""
    import threading
    import time
    exe = _native_worker()
    assert subprocess.run([exe, "othello"], capture_output=True).returncode == 1
    assert subprocess.run([exe, "othello", "0", "12"], capture_output=True).returncode == 1         # the reference asserts 384
    assert subprocess.run([exe, "pentago", "0", "1"], capture_output=True).returncode == 1
    common = ["connect_four", "1", "4", "--cover", "2", "--num-tasks-const", "4", "--num-groups", "2", "--num-iters", "3",
              "--games", "1", "--traversals", "20", "--init-games", "2", "--init-traversals", "30", "--init-max-batch", "8",
              "--init-max-queue", "4", "--seed", "77", "--root", str(tmp_path), "--run-name", "tinyrun", "--poll-seconds", "0.3",
              "--evaluator-override", "random"]
    models = tmp_path / "data" / "models" / "tinyrun"
    models.mkdir(parents=True)

    def controller():                       # the traced model of iteration i appears some time after iteration i's records
        for it in (0, 1):
            d = tmp_path / "data" / "games" / "tinyrun" / "0" / "1"
            while not (d / f"tinyrun_iteration_{it}_outcomes.npy").exists():
                time.sleep(0.05)
            time.sleep(0.5)
            (models / f"traced_tinyrun_iteration_{it}.pt").write_bytes(b"stand-in")

    th = threading.Thread(target=controller)
    th.start()
    out = subprocess.run([exe] + common, capture_output=True, text=True, timeout=240)
    th.join()
    assert out.returncode == 0, out.stderr
    log = out.stdout
    assert "Task 1 of 4, in group 0" in log and "Starting iteration 2..." in log and "Using initial network..." in log
    assert log.count("Spinning on traced model from iteration") >= 2           # it really waited for both files
    for task, group in ((1, 0), (2, 1)):    # group = task // (4 // 2)
        d = tmp_path / "data" / "games" / "tinyrun" / str(group) / str(task)
        for it, games in ((0, 2), (1, 1), (2, 1)):
            s = np.load(d / f"tinyrun_iteration_{it}_states.npy")
            p = np.load(d / f"tinyrun_iteration_{it}_distributions.npy")
            o = np.load(d / f"tinyrun_iteration_{it}_outcomes.npy")
            assert s.dtype == np.float32 and s.shape[1:] == (3, 6, 7) and p.shape == (s.shape[0], 7) and o.shape == (s.shape[0],)
        assert not list(d.glob("*.tmp"))
    # iteration 0 = games 0-1 (task 1) and 2-3 (task 2) of one oracle run with streams 1..4
    cfg = po.make_config(po.GAME_C4, 30, math_mode=po.MATH_PORTABLE)
    ora = po.selfplay(cfg, 4, 77, 1, True)
    split = ora["offsets"][2]
    d1 = np.load(tmp_path / "data/games/tinyrun/0/1/tinyrun_iteration_0_distributions.npy")
    d2 = np.load(tmp_path / "data/games/tinyrun/1/2/tinyrun_iteration_0_distributions.npy")
    assert (d1.view(np.uint32) == ora["dists"][:split].view(np.uint32)).all()
    assert (d2.view(np.uint32) == ora["dists"][split:].view(np.uint32)).all()
    # iterations 1 and 2 ran on ONE engine: its stream numbering continued (streams 5-6, then 7-8)
    ora1 = po.selfplay(po.make_config(po.GAME_C4, 20, math_mode=po.MATH_PORTABLE), 2, 77, 5, True)
    ora2 = po.selfplay(po.make_config(po.GAME_C4, 20, math_mode=po.MATH_PORTABLE), 2, 77, 7, True)
    for it, ora_it in ((1, ora1), (2, ora2)):
        a = np.load(tmp_path / f"data/games/tinyrun/0/1/tinyrun_iteration_{it}_distributions.npy")
        b = np.load(tmp_path / f"data/games/tinyrun/1/2/tinyrun_iteration_{it}_distributions.npy")
        assert (np.concatenate([a, b]).view(np.uint32) == ora_it["dists"].view(np.uint32)).all()
    # the worker's .npy bytes are the reference writer's (g5 fixture: C4, 3 games @100, one global stream is a different run;
    # here: header layout)
    raw = open(tmp_path / "data/games/tinyrun/0/1/tinyrun_iteration_0_states.npy", "rb").read()
    assert raw[:8] == b"\x93NUMPY\x01\x00" and (10 + raw[8] + 256 * raw[9]) % 16 == 0
    # compact format behind a flag
    from sprl_amd import records_v2
    out = subprocess.run([exe, "othello", "0", "1", "--num-tasks-const", "1", "--num-groups", "1", "--num-iters", "1", "--init-games", "2",
                          "--init-traversals", "24", "--init-max-batch", "8", "--init-max-queue", "4", "--seed", "5", "--root", str(tmp_path),
                          "--run-name", "v2run", "--format", "v2"], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr
    s2, d2_, o2 = records_v2.load_compact(str(tmp_path / "data/games/v2run/0/0/v2run_iteration_0.sprl2"))
    ora = po.selfplay(po.make_config(po.GAME_OTHELLO, 24, math_mode=po.MATH_PORTABLE), 2, 5, 1, True)
    assert (d2_.view(np.uint32) == ora["dists"].view(np.uint32)).all() and (o2 == ora["outcomes"]).all()
    assert s2.shape == (len(ora["players"]), 3, 8, 8)


def test_native_worker_populations(tmp_path):
    """--populations 2: the covered tasks are split over two engines driven from two host threads (on the GPU: two private HIP
    streams).  Population 0 plays its tasks on the RNG streams a single engine would have used, population 1 on its own
    disjoint range - every file is there and holds exactly the oracle's games for those streams."""
    exe = _native_worker()
    out = subprocess.run([exe, "connect_four", "0", "4", "--cover", "4", "--populations", "2", "--num-tasks-const", "4",
                          "--num-groups", "2", "--num-iters", "1", "--init-games", "2", "--init-traversals", "30",
                          "--init-max-batch", "8", "--init-max-queue", "4", "--seed", "77", "--root", str(tmp_path),
                          "--run-name", "poprun"], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr
    cfg = po.make_config(po.GAME_C4, 30, math_mode=po.MATH_PORTABLE)
    for pop, tasks in ((0, (0, 1)), (1, (2, 3))):
        ora = po.selfplay(cfg, 4, 77, 1 + pop * (1 << 27), True)          # this population's 2 tasks x 2 games
        split = ora["offsets"][2]
        for k, task in enumerate(tasks):
            d = tmp_path / "data" / "games" / "poprun" / str(task // 2) / str(task)
            got = np.load(d / "poprun_iteration_0_distributions.npy")
            want = ora["dists"][:split] if k == 0 else ora["dists"][split:]
            assert (got.view(np.uint32) == want.view(np.uint32)).all(), (pop, task)
            assert np.load(d / "poprun_iteration_0_outcomes.npy").shape[0] == got.shape[0]
    assert subprocess.run([exe, "connect_four", "0", "4", "--num-tasks-const", "4", "--cover", "2", "--populations", "3"],
                          capture_output=True).returncode == 1               # more populations than covered tasks


def test_native_worker_backs_off_when_device_memory_runs_out(tmp_path):
    """SPRL_E_NOMEM is its own code, and the worker halves the resident games both when the ARENAS do not fit
    (sprl_engine_create) and when the RECORD BUFFERS of the run do not fit beside them (sprl_engine_run allocates them) - the
    second case used to end the worker.  The emulator library bounds its "device" memory (SPRL_EMU_HBM_BYTES, a hook of the
    test build only).  Games do not depend on how many are resident, so every run writes the same bytes."""
    import re
    exe = _native_worker()
    base = ["connect_four", "0", "4", "--cover", "1", "--num-tasks-const", "4", "--num-groups", "1", "--num-iters", "1", "--init-games", "48",
            "--init-traversals", "12", "--init-max-batch", "8", "--init-max-queue", "4", "--seed", "5", "--concurrent", "4",
            "--run-name", "memrun", "--evaluator-override", "random"]

    def run(root, **env):
        out = subprocess.run([exe] + base + ["--root", str(root)], capture_output=True, text=True, timeout=240,
                             env=dict(os.environ, **env))
        assert out.returncode == 0, (out.stdout, out.stderr)
        d = root / "data" / "games" / "memrun" / "0" / "0"
        return out, b"".join(open(d / f"memrun_iteration_0_{part}.npy", "rb").read() for part in ("states", "distributions", "outcomes"))

    full, want = run(tmp_path / "a", SPRL_EMU_HBM_REPORT="1")
    peak = int(re.search(r"emu hbm peak bytes: (\d+)", full.stderr).group(1))
    assert "HBM holds" not in full.stdout and "Record buffers" not in full.stdout
    # 1 KB short of the peak: the arenas of 4 resident games fit, the record buffers of the 48 games then do not
    rec, got = run(tmp_path / "b", SPRL_EMU_HBM_BYTES=str(peak - 1024))
    assert "Record buffers did not fit beside the arenas: 2 games resident at once." in rec.stdout, rec.stdout
    assert got == want
    # one game's arena (1.1 MB at this budget) less: the arenas of 4 resident games themselves do not fit
    half, got = run(tmp_path / "c", SPRL_EMU_HBM_BYTES=str(peak - 1200 * 1024))
    assert "HBM holds 2 of the 4 games at once" in half.stdout or "HBM holds 1 of the 4 games at once" in half.stdout, half.stdout
    assert got == want


def test_timeline_tool_on_a_synthetic_trace(tmp_path):
    """tools/timeline.py (the overlap / idle / gap statistics of DESIGN.md section 5) on a hand-made two-queue kernel trace: queue 1
    runs a convolution 0-100 us and a tree kernel 110-150 us, queue 2 a convolution 50-130 us: busy 150 us of 150, two kernels
    together during 50-100 and 110-130, one 10 us gap on queue 1."""
    import csv
    import json
    import sys
    rows = [("1", "wino_conv64_kernel<8, 8, 0, 1, 0>(...)", 0, 100_000), ("1", "step_kernel<Othello>(EngineParams)", 110_000, 150_000),
            ("2", "wino_conv64_kernel<8, 8, 0, 0, 1>(...)", 50_000, 130_000)]
    path = tmp_path / "kernel_trace.csv"
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Agent_Id", "Queue_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        for q, name, s, e in rows:
            w.writerow(["KERNEL_DISPATCH", "0", q, name, 1_000_000 + s, 1_000_000 + e])
    out = tmp_path / "tl.json"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "timeline.py"), str(path), "--skip", "0", "--json", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    tl = json.load(open(out))
    assert abs(tl["wall_ms"] - 0.150) < 1e-9 and abs(tl["busy_any_share"] - 1.0) < 1e-9
    assert abs(tl["busy_two_or_more_share"] - 70 / 150) < 1e-9 and tl["queues"] == 2
    assert tl["classes"]["conv"]["launches"] == 2 and tl["classes"]["tree"]["launches"] == 1
    assert abs(tl["classes"]["conv"]["mean_us"] - 90.0) < 1e-9
    assert abs(tl["gaps"]["1"]["mean_us"] - 10.0) < 1e-9
