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


def test_pack_unpack_roundtrip(emu):
    from sprl_amd.distributed import pack_records, unpack_records
    _, rec, _ = parity.run_engine(emu, "othello", 2, concurrent_games=2, num_traversals=16, seed=9)
    u = unpack_records(pack_records(rec))
    assert (u["boards"] == rec.boards).all() and (u["movers"] == rec.movers).all()
    assert (u["pdfs"].view(np.uint32) == rec.pdfs.view(np.uint32)).all()
    assert (u["ply_offset"] == rec.ply_offset).all() and (u["winners"] == rec.winners).all()


RANK_SCRIPT = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from sprl_amd import engine as E
from sprl_amd.distributed import gather_records
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lib = E.load_library({emu!r})
games = 3
cfg = E.default_config("c4", lib, concurrent_games=2, num_traversals=24, seed=21, stream_base=1 + rank * games)
eng = E.Engine(cfg, lib); eng.set_model("random")
rec = eng.run(games)
shards = gather_records(rec, dist)
if rank == 0:
    np.savez({out!r}, **{{f"r{{i}}_{{k}}": v for i, sh in enumerate(shards) for k, v in sh.items() if isinstance(v, np.ndarray)}})
dist.barrier(); dist.destroy_process_group()
"""


def test_two_rank_game_sharding_gloo(emu, tmp_path):
    """world_size 2 on CPU: each rank plays its own shard (disjoint RNG streams); rank 0 ends up with both
    shards, which together equal oracle games 1..6 in order."""
    out = str(tmp_path / "gathered.npz")
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT.format(root=ROOT, emu=os.path.join(EMU_DIR, "libsprl_emu.so"), out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    assert [p.wait(timeout=240) for p in procs] == [0, 0]
    z = np.load(out)
    cfg = po.make_config(po.GAME_C4, 24, math_mode=po.MATH_PORTABLE)
    ora = po.selfplay(cfg, 6, 21, 1, True)
    pdfs = np.concatenate([z["r0_pdfs"], z["r1_pdfs"]])
    # oracle pdf rows are symmetrised x2; symmetry 0 (identity) rows are the compact ones
    assert (pdfs.view(np.uint32) == ora["dists"][0::2].view(np.uint32)).all()
    assert (np.concatenate([z["r0_boards"], z["r1_boards"]]) == ora["boards"][0::2]).all()
