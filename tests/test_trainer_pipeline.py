"""SURVEY §8(f) ranks 1-2 on the CPU: the trainer step's loss/weighting semantics against a literal restatement of
the reference formula, the replay-window weights, and the in-process self-play -> train -> hot-swap loop (engine
on the SIMT emulator, network forward as a host callback)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from sprl_amd import engine as E
from sprl_amd import trainer as T
from sprl_amd.network import GridResNet
from sprl_amd.pipeline import LoopConfig, SelfPlayTrainLoop

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")


def test_weighted_losses_match_reference_formula():
    torch.manual_seed(0)
    lo, va = torch.randn(7, 5), torch.tanh(torch.randn(7, 1))
    pdf = torch.softmax(torch.randn(7, 5), 1)
    z = torch.sign(torch.randn(7, 1))
    w = torch.tensor([1., 1, 2, 2, 3, 3, 3]).reshape(-1, 1)
    p, v = T.weighted_losses(lo, va, pdf, z, w)
    sm = torch.softmax(lo, dim=1)                              # othello_controller.py:160-170, literally
    ref_p = torch.sum(-torch.sum(pdf * torch.log(sm + 1e-8), dim=1, keepdim=True) * w) / torch.sum(w)
    ref_v = torch.sum((z - va) ** 2 * w) / torch.sum(w)
    assert torch.equal(p, ref_p) and torch.equal(v, ref_v)


def test_replay_window_and_lr_schedule():
    cfg = T.TrainerConfig(num_past_iters_to_train=3)
    win = T.ReplayWindow(cfg, "cpu")
    for it in range(5):
        win.add(it, np.zeros((2, 3, 6, 7), np.float32), np.full((2, 7), 1 / 7, np.float32), np.ones(2, np.float32))
    s, d, o, t = win.training_tensors(4)
    assert s.shape[0] == 6 and o.shape == (6, 1)
    assert t.reshape(-1).tolist() == [1, 1, 2, 2, 3, 3]        # iterations 2,3,4 -> timestamps 3,4,5 minus (5 - 3)
    assert [T.learning_rate_for(T.TrainerConfig(), i) for i in (0, 4, 5, 10, 20)] == \
        pytest.approx([0.01, 0.01, 0.001, 0.0001, 0.00001])


def test_training_reduces_loss_and_returns_best_snapshot():
    torch.manual_seed(1)
    net = GridResNet(6, 7, 7, 1, 1, 8)
    n = 256
    s = (torch.rand(n, 3, 6, 7) > 0.5).float()
    d = torch.zeros(n, 7)
    d[torch.arange(n), s[:, 0, 5].argmax(1)] = 1.0             # learnable target: a function of the bottom row
    o = torch.where(s[:, 2, 0, 0:1] > 0, 1.0, -1.0)
    t = torch.ones(n, 1)
    cfg = T.TrainerConfig(batch_size=64, max_groups=2, epochs_per_group=4)
    best, hist = T.train_network(net, 0.01, (s, d, o, t), cfg, generator=torch.Generator().manual_seed(0))
    first, last = hist["epochs"][0], hist["epochs"][-1]
    assert last["train_policy"] < first["train_policy"]
    assert set(best.keys()) == set(net.state_dict().keys())
    assert hist["best_val"] <= min(e["val_policy"] + e["val_value"] for e in hist["epochs"]) + 1e-9


def test_in_process_loop_hot_swaps_the_model(tmp_path):
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    emu = E.load_library(os.path.join(EMU_DIR, "libsprl_emu.so"))
    calls = {"n": 0}

    def forward_factory(net):
        def fwd(planes_ptr, batch, logits_ptr, value_ptr):
            x = np.ctypeslib.as_array(C.cast(planes_ptr, C.POINTER(C.c_float)), shape=(batch, 3, 6, 7))
            with torch.no_grad():
                lo, va = net.cpu().eval()(torch.from_numpy(x.copy()))
            np.ctypeslib.as_array(C.cast(logits_ptr, C.POINTER(C.c_float)), shape=(batch, 7))[:] = lo.numpy()
            np.ctypeslib.as_array(C.cast(value_ptr, C.POINTER(C.c_float)), shape=(batch,))[:] = va.numpy().reshape(-1)
            calls["n"] += 1
            return 0
        return fwd

    cfg = LoopConfig(game="connect_four", num_iters=2, init_games=4, init_traversals=16, init_max_batch=8,
                     init_max_queue=4, games=3, traversals=12, num_blocks=1, num_channels=8, root=str(tmp_path),
                     run_name="loop", write_files=True)
    tcfg = T.TrainerConfig(batch_size=32, max_groups=1, epochs_per_group=2)
    loop = SelfPlayTrainLoop(cfg, tcfg, lib=emu, forward_factory=forward_factory, train_device="cpu", log=lambda *_: None)
    hist = loop.run()
    assert [h["iteration"] for h in hist] == [0, 1] and hist[0]["games"] == 4 and hist[1]["games"] == 3
    assert calls["n"] > 0                                      # iteration 1 searched with the freshly trained network
    for it in (0, 1):
        assert os.path.exists(tmp_path / "data" / "models" / "loop" / f"traced_loop_iteration_{it}.pt")
        s = np.load(tmp_path / "data" / "games" / "loop" / "0" / "0" / f"loop_iteration_{it}_states.npy")
        assert s.shape[0] == hist[it]["samples"]
    m = torch.jit.load(str(tmp_path / "data" / "models" / "loop" / "traced_loop_iteration_1.pt"))
    lo, va = m(torch.zeros(2, 3, 6, 7))
    assert lo.shape == (2, 7) and va.shape == (2, 1)
    # without write_files nothing touches the file system (no .pt rendez-vous, no .npy round trip)
    quiet = tmp_path / "quiet"
    quiet.mkdir()
    cfg2 = LoopConfig(game="connect_four", num_iters=2, init_games=2, init_traversals=8, init_max_batch=8, init_max_queue=4,
                      games=2, traversals=8, num_blocks=1, num_channels=8, root=str(quiet), run_name="q")
    loop2 = SelfPlayTrainLoop(cfg2, tcfg, lib=emu, forward_factory=forward_factory, train_device="cpu", log=lambda *_: None)
    assert len(loop2.run()) == 2 and not list(quiet.rglob("*"))


def test_trace_to_bytes_is_the_file_archive(tmp_path):
    """The in-memory archive handed to sprl_engine_set_model_buffer loads to the same module as the reference-style file."""
    import io
    from sprl_amd.network import make_network, trace_to_bytes, trace_to_file
    net = make_network("connect_four", 1, 8, seed=3)
    raw = trace_to_bytes(net, "connect_four")
    a = torch.jit.load(io.BytesIO(raw))
    b = torch.jit.load(trace_to_file(net, str(tmp_path / "t.pt"), "connect_four"))
    x = torch.rand(3, 3, 6, 7)
    for u, v in zip(a(x), b(x)):
        assert torch.equal(u, v)


@pytest.mark.parametrize("game", ["c4", "othello", "go", "go9"])
def test_compact_record_format_v2_roundtrip(game, tmp_path):
    """the compact on-disk form expands to exactly the reference arrays."""
    from sprl_amd import records_v2
    import parity
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    emu = E.load_library(os.path.join(EMU_DIR, "libsprl_emu.so"))
    _, rec, _ = parity.run_engine(emu, game, 2, concurrent_games=2, num_traversals=20, seed=13)
    path = str(tmp_path / "run_iteration_0.sprl2")
    records_v2.write_compact(path, rec)
    s2, d2, o2 = records_v2.load_compact(path)
    s1, d1, o1 = rec.expand()
    assert s1.shape == s2.shape and (s1 == s2).all()
    assert (d1.view(np.uint32) == d2.view(np.uint32)).all() and (o1 == o2).all()
    v1_bytes = s1.nbytes + d1.nbytes + o1.nbytes
    assert os.path.getsize(path) * 8 < v1_bytes


def test_trainer_against_reference_controller_fixture(golden):
    """f-2: tests/golden/g_trainer.npz was produced by the REFERENCE's train_network
    (scripts/othello_controller.py:128-241) on a synthetic window, recording the batches it drew.  Replayed on the same
    batches from the same initial weights, our trainer must select the same best epoch, stop after the same number of epochs
    and end with the same weights: the saved best model and the live network agree on a probe batch to 1e-5.
    (The same replay runs on the MI355X under -m gpu: tests/test_gpu_parity.py::test_trainer_fixture_on_the_gpu.)"""
    import parity
    parity.replay_trainer_fixture(golden("g_trainer.npz"), "cpu", atol=1e-5)


DDP_SCRIPT = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from sprl_amd import trainer as T
from sprl_amd.network import GridResNet
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2)
torch.manual_seed(3)
net = GridResNet(6, 7, 7, 1, 1, 8)                      # same initial weights on both ranks
g = torch.Generator().manual_seed(100 + rank)
n = 200 if rank == 0 else 331                           # game lengths differ: the shards are unequal
s = (torch.rand(n, 3, 6, 7, generator=g) > 0.5).float()
d = torch.softmax(torch.randn(n, 7, generator=g), 1)
o = torch.sign(torch.randn(n, 1, generator=g))
t = torch.ones(n, 1)
cfg = T.TrainerConfig(batch_size=64, max_groups=3, epochs_per_group=2)
best, hist = T.train_network(net, 0.01, (s, d, o, t), cfg, generator=torch.Generator().manual_seed(7 + rank), ddp=True)
flat = torch.cat([v.reshape(-1).float() for v in best.values()])
live = torch.cat([v.reshape(-1).float() for v in net.state_dict().values()])
np.savez({out!r} + str(rank) + ".npz", best=flat.numpy(), live=live.numpy(), best_epoch=hist["best_epoch"], epochs=len(hist["epochs"]),
         val=np.array([e["val_policy"] + e["val_value"] for e in hist["epochs"]]))
dist.barrier(); dist.destroy_process_group()
"""


def test_ddp_training_with_unequal_shards_stays_in_step(tmp_path):
    """ranks hold shards of different sizes.  The step count per epoch is agreed across ranks and the
    validation sums are all-reduced, so both ranks run the same number of epochs, pick the same best epoch and export
    identical weights (world size 2, gloo)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "ddp_rank.py"
    script.write_text(DDP_SCRIPT.format(root=root, out=str(tmp_path / "r")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    assert [p.wait(timeout=300) for p in procs] == [0, 0]
    a, b = np.load(str(tmp_path / "r0.npz")), np.load(str(tmp_path / "r1.npz"))
    assert int(a["best_epoch"]) == int(b["best_epoch"]) and int(a["epochs"]) == int(b["epochs"])
    assert np.array_equal(a["val"], b["val"])
    assert np.array_equal(a["best"], b["best"]) and np.array_equal(a["live"], b["live"])


DDP_EMPTY_SCRIPT = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from sprl_amd import trainer as T
from sprl_amd.network import GridResNet
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2)
torch.manual_seed(3)
net = GridResNet(6, 7, 7, 1, 1, 8)
n = 200 if rank == 0 else 1                             # rank 1: one sample, nothing left for training after the split
s = (torch.rand(n, 3, 6, 7) > 0.5).float()
d = torch.softmax(torch.randn(n, 7), 1)
o = torch.sign(torch.randn(n, 1))
t = torch.ones(n, 1)
cfg = T.TrainerConfig(batch_size=64, max_groups=1, epochs_per_group=1)
try:
    T.train_network(net, 0.01, (s, d, o, t), cfg, ddp=True)
    code = 1
except ValueError:
    code = 0                                            # EVERY rank raises: nobody is left waiting in a collective
dist.barrier(); dist.destroy_process_group()
sys.exit(code)
"""


def test_ddp_empty_shard_is_an_error_on_every_rank(tmp_path):
    """one rank's shard leaves no training samples.  The error is agreed in the step-count all-reduce, so BOTH ranks
    raise - the rank with the full shard does not hang in the collective until the watchdog fires (world size 2, gloo)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "ddp_empty.py"
    script.write_text(DDP_EMPTY_SCRIPT.format(root=root))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    assert [p.wait(timeout=120) for p in procs] == [0, 0]
