"""GPU parity tests proper (run with -m gpu on the MI355X box): the hand-written gfx950 kernels, called through
the C ABI, against the CPU oracle on the same seeds.  Bit-exact for boards, movers, pdf bits, outcomes and
search counters; the CNN path is compared within the tolerance stated in each test."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from sprl_amd import engine as E
import parity

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parity_ref_available():
    from oracle import pyref
    return pyref.available(True)


@pytest.fixture(scope="module")
def lib():
    L = E.load_library()          # raises if the gfx950 build is missing: no fallback
    assert L.sprl_device_available() == 1, "no MI355X visible"
    return L


@pytest.fixture(scope="module")
def traced_model(tmp_path_factory):
    from sprl_amd.network import make_network, trace_to_file
    d = tmp_path_factory.mktemp("model")
    return trace_to_file(make_network("othello", 2, 64, seed=0), str(d / "traced_oth.pt"), "othello")


def test_othello_random_whole_games(lib):
    parity.check_case(lib, "othello", 6, concurrent_games=4, num_traversals=100)


def test_othello_heuristic_whole_games(lib):
    parity.check_case(lib, "othello", 4, model="heuristic", concurrent_games=4, num_traversals=100)


def test_othello_800_traversals_reference_budget(lib):
    """BASELINE config: 800 traversals/move, batch 8 / queue 4, Dirichlet (0.25, 0.3), D4."""
    rec, st = parity.check_case(lib, "othello", 3, concurrent_games=3, num_traversals=800, seed=2024)
    assert st["compactions"] == 0


def test_c4_reference_config(lib):
    """BASELINE configs[0]: Connect Four, 100 traversals/move (reference CPU-runnable case)."""
    parity.check_case(lib, "c4", 16, concurrent_games=8, num_traversals=100)


def test_go7_whole_games(lib):
    """Go 7x7 (reference size), GoWorker constants (batch 16 / queue 8, alpha 0.2), 400 traversals/move."""
    rec, st = parity.check_case(lib, "go", 6, concurrent_games=6, num_traversals=400)
    assert rec.planes == 17


def test_go7_many_games_short_budget(lib):
    parity.check_case(lib, "go", 128, concurrent_games=128, num_traversals=32, seed=77)


def test_go7_compaction(lib):
    rec, st = parity.check_case(lib, "go", 4, concurrent_games=4, num_traversals=64, node_cap=200, spare_arenas=4, no_recycle=1)
    assert st["compactions"] > 0


def test_wide_kernel_go7_and_go9(lib):
    """Multi-strip kernel: at the reference-pinned 7x7 size and at 9x9 (BASELINE config 4 geometry: 1600 traversals,
    batch 16 / queue 8, D4)."""
    parity.check_case(lib, "go7_wide", 4, concurrent_games=4, num_traversals=200)
    rec, st = parity.check_case(lib, "go9", 4, concurrent_games=4, num_traversals=1600, seed=12)
    assert rec.cells == 81
    parity.check_case(lib, "go9", 64, concurrent_games=64, num_traversals=48, seed=3)
    rec, st = parity.check_case(lib, "go9", 4, concurrent_games=4, num_traversals=64, node_cap=300, spare_arenas=4, no_recycle=1)
    assert st["compactions"] > 0
    rec, st = parity.check_case(lib, "go9", 4, concurrent_games=4, num_traversals=64, node_cap=300, spare_arenas=0)
    assert st["compactions"] == 0 and st["nodes_recycled"] > 0.5 * st["nodes_created"] and st["max_nodes_in_arena"] <= 300


def test_go19(lib):
    """Go 19x19 geometry (BASELINE config 5): 6-strip rows; short budget, whole games."""
    rec, st = parity.check_case(lib, "go19", 2, concurrent_games=2, num_traversals=64, seed=8)
    assert rec.cells == 361 and rec.planes == 17


def test_compaction_tiny_arena(lib):
    """Bump allocation + Cheney compaction alone (node recycling off): arenas change owner across XCDs."""
    rec, st = parity.check_case(lib, "othello", 4, concurrent_games=4, num_traversals=60, node_cap=160, spare_arenas=4, no_recycle=1)
    assert st["compactions"] > 0 and st["nodes_recycled"] == 0


def test_node_recycling_small_arenas(lib):
    """The nodes of pruned siblings are reused (reference: UCTNode::pruneChildrenExcept frees them, uct/UCTNode.hpp:356-366):
    whole games stay bit-identical to the oracle, the same tiny arenas now need no compaction, and at the BASELINE budget
    (800 traversals/move) a game's arena high-water mark stays a few thousand nodes instead of ~36 000."""
    rec, st = parity.check_case(lib, "othello", 4, concurrent_games=4, num_traversals=60, node_cap=160, spare_arenas=4)
    assert st["compactions"] == 0 and st["nodes_recycled"] > 0.5 * st["nodes_created"] and st["max_nodes_in_arena"] <= 160
    rec, st = parity.check_case(lib, "go", 4, concurrent_games=4, num_traversals=64, node_cap=260, spare_arenas=4, seed=17)
    assert st["compactions"] == 0 and st["nodes_recycled"] > 0
    rec, st = parity.check_case(lib, "othello", 8, concurrent_games=8, num_traversals=800, seed=21)
    assert st["compactions"] == 0 and st["max_nodes_in_arena"] < 4224 and st["nodes_recycled"] > 0.9 * st["nodes_created"]


def test_child_indices_beyond_16_bits(lib):
    """Arenas above 65535 nodes: 24-bit child indices (the reference's iteration-0 budget of 131072 traversals per
    move does not fit 16-bit indices).  The allocator is started next to the boundary (test hook), then a real large budget."""
    rec, st = parity.check_case(lib, "othello", 2, concurrent_games=2, num_traversals=40, node_cap=70000, no_recycle=1, seed=3,
                                alloc_base=65400)
    assert st["max_nodes_in_arena"] > 65535 + 500 and st["compactions"] == 0
    # the reference's real iteration-0 constants (OTHWorker.cpp:17-20: 3 games x 131072 traversals, batch 1 / queue 1), eight
    # tasks covered by one engine: the default arena (2 x traversals + 5120 nodes = 0.26 GiB per game) must fit comfortably
    cfg = E.default_config("othello", lib, concurrent_games=24, num_traversals=131072, max_batch=1, max_queue=1)
    eng = E.Engine(cfg, lib)
    assert eng.stats()["hbm_bytes"] < 24 * 0.3 * 2 ** 30 + 2 ** 30
    eng.close()
    # a whole game at 65536 traversals/move, batch 1 / queue 1 (the shape of the reference's iteration 0), bit-exact vs the oracle
    rec, st = parity.check_case(lib, "othello", 1, concurrent_games=1, num_traversals=65536, max_batch=1, max_queue=1, seed=9)
    assert st["max_nodes_in_arena"] > 65535 and st["compactions"] == 0


def test_batch1_queue1_nosym_nonoise(lib):
    parity.check_case(lib, "othello", 2, concurrent_games=2, num_traversals=50, use_symmetry=0, add_noise=0,
                      max_batch=1, max_queue=1)


def test_symmetrised_mask_option(lib):
    parity.check_case(lib, "othello", 2, concurrent_games=2, num_traversals=50, mask_frame=E.MASK_SYMMETRISED)


def test_many_concurrent_games_are_independent(lib):
    """Game g must not depend on how many games run beside it: 256 slots vs the oracle one game at a time."""
    parity.check_case(lib, "othello", 256, concurrent_games=256, num_traversals=24, seed=5)


def test_network_toy_forward_callback(lib):
    """Encode -> forward -> decode through DEVICE buffers with a deterministic stand-in network."""
    import torch
    A = 65

    def fwd(planes_ptr, batch, logits_ptr, value_ptr):
        class _Arr:
            def __init__(self, ptr, shape):
                self.__cuda_array_interface__ = dict(shape=shape, typestr="<f4", data=(ptr, False), version=2)
        x = torch.as_tensor(_Arr(planes_ptr, (batch, 3, 8, 8)), device="cuda").cpu().numpy()
        lo, va = parity.toy_forward_numpy(x, A)
        torch.as_tensor(_Arr(logits_ptr, (batch, A)), device="cuda").copy_(torch.from_numpy(lo))
        torch.as_tensor(_Arr(value_ptr, (batch,)), device="cuda").copy_(torch.from_numpy(va))
        torch.cuda.synchronize()
        return 0

    cfg, rec, st = parity.run_engine(lib, "othello", 2, forward=fwd, concurrent_games=2, num_traversals=24, seed=5)
    cb = po.make_forward(lambda x: parity.toy_forward_numpy(x, A), po.GAME_OTHELLO)
    ora = po.selfplay(parity.oracle_config("othello", cfg, po.EVAL_CALLBACK, forward=cb), 2, 5, 1, True)
    parity.assert_same_games(rec, ora)
    parity.assert_same_counters(st, ora["stats"])


def _plugin():
    import ctypes as C
    plug = C.CDLL(os.path.join(os.path.dirname(E.DEFAULT_LIB), "libsprl_amd_torch.so"))
    plug.sprl_torch_load.restype = C.c_void_p
    plug.sprl_torch_load.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    plug.sprl_torch_forward.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p, C.c_char_p, C.c_int]
    plug.sprl_torch_is_native.argtypes = [C.c_void_p]
    plug.sprl_torch_free.argtypes = [C.c_void_p]
    return plug


def _plugin_forward(plug, h, x, actions):
    """The evaluator entry point engine.cpp calls for every search round (sprl_torch_forward), on device tensors."""
    import ctypes as C
    import torch
    err = C.create_string_buffer(512)
    b = x.shape[0]
    lo = torch.zeros(b, actions, device="cuda")
    va = torch.zeros(b, device="cuda")
    assert plug.sprl_torch_forward(h, x.data_ptr(), b, x.shape[1], x.shape[2], x.shape[3], lo.data_ptr(), actions, va.data_ptr(), err, 512) == 0, err.value
    torch.cuda.synchronize()
    return lo.cpu().numpy(), va.cpu().numpy()


def test_torchscript_cnn_through_libtorch(lib, traced_model):
    """The engine's evaluator on positions the engine itself recorded: self-play with the traced CNN, then the recorded
    states go through the SAME plugin entry point the engine calls every round (sprl_torch_forward -> hand-written gfx950
    path) and are compared with a CPU fp32 forward of the same TorchScript file - what the reference worker computes
    (networks/GridNetwork.hpp:99-102, torch::kCPU :157)."""
    import ctypes as C
    import torch
    cfg, rec, st = parity.run_engine(lib, "othello", 4, model=traced_model, concurrent_games=4, num_traversals=64)
    states, dists, outcomes = rec.expand()
    assert st["games"] == 4 and st["nn_batches"] > 0 and st["nn_evals"] > 0
    assert np.allclose(dists.sum(1), 1.0, atol=1e-4)
    assert set(np.unique(outcomes).tolist()) <= {-1.0, 0.0, 1.0}
    plug = _plugin()
    err = C.create_string_buffer(512)
    h = plug.sprl_torch_load(traced_model.encode(), 0, err, 512)
    assert h, err.value
    x = torch.from_numpy(states[:512]).contiguous()
    lg, vg = _plugin_forward(plug, h, x.cuda(), 65)
    plug.sprl_torch_free(h)
    m_cpu = torch.jit.load(traced_model, map_location="cpu").eval()
    with torch.no_grad():
        lc, vc = m_cpu(x)
    e_l, e_v = np.abs(lg - lc.numpy()).max(), np.abs(vg - vc.numpy().reshape(-1)).max()
    print(f"engine evaluator vs LibTorch-CPU on {x.shape[0]} recorded positions: max|dlogit| {e_l:.3e}, max|dvalue| {e_v:.3e}")
    assert e_l < CNN_ATOL and e_v < CNN_ATOL


# Measured on MI355X (profiles/r02_cnn_error.txt): on the BASELINE-shape network the hand-written fp32 path (Winograd
# F(4x4,3x3) trunk on fp32 MFMA) differs from the reference's fp32 outputs by 1.5e-8 (default-init scale, |logits| <= 0.1) and
# 3.6e-6 (logits of size 3; the reference's own fp32 is 1.5e-6 from float64 there); on positions the engine recorded 7.6e-8.
# Tolerance = SURVEY section 8(c)'s 1e-5, 2.8x the largest error seen - at those magnitudes.  An absolute bound cannot hold at
# every scale in fp32 (the reference's OWN CPU fp32 forward is 1.1e-5 from float64 at |logits| = 16, g9b gain 3), so the
# tolerance is stated as  max(CNN_ATOL, CNN_RTOL x the largest |logit| of the batch's network output): CNN_RTOL is relative to
# the output scale, not to each element (a logit near zero that is the difference of large terms carries their error).
# Measured on MI355X (round 3, gpurun_out/r03a_gpu_tests.log -> profiles/r03_cnn_error.txt): gain 3, |logits| <= 16: 4.2e-5 from
# the reference's fp32 outputs = 2.6e-6 of the scale (gain 2: 1.1e-6 of the scale; the reference's own fp32: 0.7e-6).
# CNN_RTOL = 5e-6 is 1.9x the largest relative error seen.
CNN_ATOL = 1e-5
CNN_RTOL = 5e-6


def cnn_tol(reference_logits):
    return max(CNN_ATOL, CNN_RTOL * float(np.abs(reference_logits).max()))


def test_handwritten_cnn_against_reference_golden_baseline_shape(lib, golden, tmp_path):
    """VERDICT r1 #2b: reference-generated golden at the BASELINE shape.  tests/golden/g9b_baseline_network.npz holds the
    outputs of the REFERENCE's BasicGridNetwork(8,8,65,1,2,64) (CPU fp32 and float64) for seed-reproducible weights; the same
    weights go into our module, are traced like the controller does, and run through the hand-written gfx950 path."""
    import ctypes as C
    import sys
    import torch
    from sprl_amd.network import GridResNet, trace_to_file
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import netfill
    g = golden("g9b_baseline_network.npz")
    plug = _plugin()
    err = C.create_string_buffer(512)
    x = torch.from_numpy(g["input"]).cuda().contiguous()
    report = []
    for gi, gain in enumerate(g["gains"]):
        net = netfill.fill_state_dict(GridResNet(8, 8, 65, 1, 2, 64), int(g["seed"][0]) + gi, float(gain)).eval()
        path = trace_to_file(net, str(tmp_path / f"g9b_{gi}.pt"), "othello")
        h = plug.sprl_torch_load(path.encode(), 0, err, 512)
        assert h and plug.sprl_torch_is_native(h) == 2, err.value
        lo, va = _plugin_forward(plug, h, x, 65)
        assert _path_info(plug, h) == "kind=2 stem=1 tail=2 lab=[]"  # four launches: stem inside conv 1, heads + FC inside conv 4; no lab switch
        plug.sprl_torch_free(h)
        e_ref = max(np.abs(lo - g[f"logits{gi}"]).max(), np.abs(va - g[f"value{gi}"].reshape(-1)).max())
        e_f64 = max(np.abs(lo - g[f"logits_f64_{gi}"]).max(), np.abs(va - g[f"value_f64_{gi}"].reshape(-1)).max())
        cpu_f64 = max(np.abs(g[f"logits{gi}"] - g[f"logits_f64_{gi}"]).max(), np.abs(g[f"value{gi}"] - g[f"value_f64_{gi}"]).max())
        scale = float(np.abs(g[f"logits{gi}"]).max())
        report.append(f"gain {gain}: |logits| <= {scale:.2f}; hand-written vs reference fp32 {e_ref:.3e} (= {e_ref / scale:.2e} of the "
                      f"scale), vs float64 {e_f64:.3e}; reference fp32 vs float64 {cpu_f64:.3e}; tolerance {cnn_tol(g[f'logits{gi}']):.1e}")
        assert e_ref < cnn_tol(g[f"logits{gi}"]), report[-1]
    print("\n".join(report))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "cnn_error.txt"), "w").write("\n".join(report) + "\n")


def _path_info(plug, h):
    import ctypes as C
    plug.sprl_torch_path_info.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    buf = C.create_string_buffer(384)
    assert plug.sprl_torch_path_info(h, buf, 384) == 0
    return buf.value.decode()


@pytest.mark.parametrize("game,tail", [("othello", 2), ("connect_four", 2), ("go7", 2), ("othello", 1), ("go7", 1)])
def test_native_cnn_path_matches_torchscript(lib, game, tail, tmp_path, monkeypatch):
    """The plugin's recognised-architecture path — stem kernel, Winograd F(4x4,3x3) trunk on fp32 MFMA with fused
    BN/residual/ReLU, the 1x1 heads and (round 4) the FC layers fused behind the last convolution (cnn_wino.hip,
    cnn_epilogue.hip) — against the plain TorchScript fp32 forward of the same file: 1e-5 absolute on logits and value
    (CNN_ATOL below; measured errors in profiles/r02_cnn_error.txt).  tail = 2: the forward ends in the last convolution's
    launch (the default); tail = 1: the two-kernel form (heads fused, FC kernel), kept for shapes the fused form does not
    cover and selected here by the lab switch SPRL_TORCH_NO_CONV_FC - which is read when the model is LOADED and must show
    up in sprl_torch_path_info."""
    import ctypes as C
    import torch
    if tail == 1:
        monkeypatch.setenv("SPRL_TORCH_NO_CONV_FC", "1")
        monkeypatch.setenv("SPRL_TORCH_NO_CONV_STEM", "1")
    from sprl_amd.network import GAME_SHAPES, make_network, trace_to_file
    model = trace_to_file(make_network(game, 2, 64, seed=1), str(tmp_path / f"traced_{game}.pt"), game)
    rows, cols, actions, hist = GAME_SHAPES[game]
    planes = 2 * hist + 1
    plug = C.CDLL(os.path.join(os.path.dirname(E.DEFAULT_LIB), "libsprl_amd_torch.so"))
    plug.sprl_torch_load.restype = C.c_void_p
    plug.sprl_torch_load.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    plug.sprl_torch_forward.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p,
                                                                                 C.c_char_p, C.c_int]
    plug.sprl_torch_is_native.argtypes = [C.c_void_p]
    plug.sprl_torch_free.argtypes = [C.c_void_p]
    err = C.create_string_buffer(512)
    h = plug.sprl_torch_load(model.encode(), 0, err, 512)
    assert h, err.value
    assert plug.sprl_torch_is_native(h) == 2
    ref = torch.jit.load(model, map_location="cuda").eval()
    for batch in (3, 1024, 4099):
        x = (torch.rand(batch, planes, rows, cols, device="cuda") > 0.6).float().contiguous()
        lo = torch.zeros(batch, actions, device="cuda")
        va = torch.zeros(batch, device="cuda")
        assert plug.sprl_torch_forward(h, x.data_ptr(), batch, planes, rows, cols, lo.data_ptr(), actions, va.data_ptr(), err, 512) == 0, err.value
        torch.cuda.synchronize()
        with torch.no_grad():
            rl, rv = ref(x)
        np.testing.assert_allclose(lo.cpu().numpy(), rl.cpu().numpy(), atol=CNN_ATOL, rtol=0)
        np.testing.assert_allclose(va.cpu().numpy(), rv.cpu().numpy().reshape(-1), atol=CNN_ATOL, rtol=0)
    stem = 1 if (planes == 3 and tail == 2) else 0          # (go7: 17 planes keep the stem launch; the tail = 1 leg also switches the stem fold off)
    assert _path_info(plug, h) == f"kind=2 stem={stem} tail={tail} lab=[{'SPRL_TORCH_NO_CONV_FC SPRL_TORCH_NO_CONV_STEM' if tail == 1 else ''}]"
    monkeypatch.delenv("SPRL_TORCH_NO_CONV_STEM", raising=False)
    monkeypatch.delenv("SPRL_TORCH_NO_CONV_FC", raising=False)
    h2 = plug.sprl_torch_load(model.encode(), 0, err, 512)           # the switch was read at load: a model loaded now has none
    assert _path_info(plug, h2).endswith("lab=[]")
    plug.sprl_torch_free(h2)
    plug.sprl_torch_free(h)


def test_full_size_properties(lib):
    """BASELINE-sized launch geometry (4096 resident games) on a short budget: size-independent properties —
    pdfs are distributions supported on legal moves, outcomes antisymmetric per game, plies consistent."""
    cfg, rec, st = parity.run_engine(lib, "othello", 4096, concurrent_games=4096, num_traversals=16, seed=3)
    assert rec.num_games == 4096 and st["games"] == 4096 and st["compactions"] == 0
    assert np.allclose(rec.pdfs.sum(1), 1.0, atol=1e-5)
    assert (np.diff(rec.ply_offset) >= 9).all() and (np.diff(rec.ply_offset) <= 128).all()
    occupied = rec.boards[:, :64] >= 0
    assert (rec.pdfs[:, :64][occupied] == 0).all()        # no probability on occupied squares
    # spot-check 8 games against the oracle
    ora = po.selfplay(parity.oracle_config("othello", cfg, po.EVAL_RANDOM), 8, 3, 1, True)
    b, p = rec.expand_boards()
    n = len(ora["players"])
    assert (b[:n] == ora["boards"]).all() and (p[:n] == ora["players"]).all()


def test_npy_roundtrip(lib, tmp_path):
    cfg, rec, _ = parity.run_engine(lib, "c4", 3, concurrent_games=3, num_traversals=50, seed=3)
    rec.write_npy(str(tmp_path / "run_iteration_0"))
    s = np.load(tmp_path / "run_iteration_0_states.npy")
    d = np.load(tmp_path / "run_iteration_0_distributions.npy")
    o = np.load(tmp_path / "run_iteration_0_outcomes.npy")
    assert s.shape[0] == d.shape[0] == o.shape[0] == rec.num_samples and s.shape[1:] == (3, 6, 7)
    ocfg = parity.oracle_config("c4", cfg, po.EVAL_RANDOM)
    ora = po.selfplay(ocfg, 3, 3, 1, True)
    po.write_records(ocfg, str(tmp_path / "ora_iteration_0"), ora)
    for part in ("states", "distributions", "outcomes"):
        assert open(tmp_path / f"run_iteration_0_{part}.npy", "rb").read() == \
            open(tmp_path / f"ora_iteration_0_{part}.npy", "rb").read()


def test_in_process_selfplay_training_loop(lib, tmp_path):
    """engine -> records -> HBM tensors -> AdamW steps on the GPU -> traced model -> set_model,
    three iterations of Connect Four in one process, no file polling."""
    import torch
    from sprl_amd import trainer as T
    from sprl_amd.pipeline import LoopConfig, SelfPlayTrainLoop
    cfg = LoopConfig(game="connect_four", num_iters=3, init_games=64, init_traversals=64, init_max_batch=8,
                     init_max_queue=4, games=64, traversals=48, num_blocks=1, num_channels=16, root=str(tmp_path),
                     run_name="gpu_loop")
    tcfg = T.TrainerConfig(batch_size=256, max_groups=1, epochs_per_group=3)
    loop = SelfPlayTrainLoop(cfg, tcfg, lib=lib, log=lambda *_: None)
    hist = loop.run()
    assert len(hist) == 3 and all(h["games"] == 64 for h in hist)
    assert str(loop.window.training_tensors(2)[0].device).startswith("cuda")
    assert hist[2]["best_val"] < hist[0]["best_val"] + 1.0          # training ran and produced finite losses
    assert all(np.isfinite(h["best_val"]) for h in hist)
    # f-1: the samples of iteration 0 (built-in evaluator) were expanded ON THE DEVICE into the window's tensors: they equal
    # the oracle's games bit for bit - planes, tempered pdfs, outcomes, in the reference's sample order
    s0, d0, o0, _ = loop.window.items[0]
    ora = po.selfplay(po.make_config(po.GAME_C4, 64, math_mode=po.MATH_PORTABLE), 64, 1, 1, True)
    assert s0.shape[0] == len(ora["players"]) and str(s0.device).startswith("cuda")
    assert (d0.cpu().numpy().view(np.uint32) == ora["dists"].view(np.uint32)).all()
    assert (o0.cpu().numpy().reshape(-1) == ora["outcomes"]).all()
    own = (ora["boards"] == ora["players"][:, None]).reshape(-1, 6, 7)
    assert (s0[:, 0].cpu().numpy() == own).all() and (s0[:, 2, 0, 0].cpu().numpy() == (ora["players"] == 0)).all()
    # the model went back to the engine through memory (sprl_engine_set_model_buffer) and was used: evaluations happened
    assert hist[1]["samples"] > 0 and isinstance(loop.traced, bytes)
    assert not list(tmp_path.rglob("*.pt")) and not list(tmp_path.rglob("*.npy"))     # nothing went through the file system
    # every iteration carries its stage times (whole-iteration turnaround; the Othello numbers come from
    # tools/iteration_turnaround.py): the stages add up to the total, and the swap of iteration i+1 is the model of iteration i
    for h in hist:
        parts = h["t_swap"] + h["t_selfplay"] + h["t_ingest"] + h["t_window"] + h["t_train"] + h["t_export"]
        assert h["t_selfplay"] > 0 and h["t_train"] > 0 and h["epochs"] == 3 and h["optimiser_steps"] >= 3
        assert parts <= h["t_total"] * 1.001 and parts >= 0.9 * h["t_total"], h


# ---- match play (Evaluate.cpp) on the device ----

def test_match_othello_random_vs_heuristic(lib):
    agents = [dict(model="random", use_symmetry=True, parent_q=False), dict(model="heuristic", use_symmetry=True, parent_q=True)]
    parity.check_match(lib, "othello", agents, 12, concurrent_games=8, num_traversals=100, seed=777)


def test_match_c4_and_go7(lib):
    a = [dict(model="random", use_symmetry=True, parent_q=True), dict(model="random", use_symmetry=False, parent_q=False)]
    parity.check_match(lib, "c4", a, 16, concurrent_games=16, num_traversals=100)
    parity.check_match(lib, "go", a, 4, concurrent_games=4, num_traversals=64, node_cap=300, spare_arenas=4)


def test_match_wide_boards(lib):
    """Match play on boards wider than 8 (VERDICT r2 missing #5; interface/play.hpp:24-70 through step_kernel_wide.h: step_match):
    Go 9x9 with agents of different symmetrisation / InitQ, more games than resident pairs, and one 19x19 game - move lists, lengths
    and winners equal the oracle's restatement of playGame."""
    agents = [dict(model="random", use_symmetry=True, parent_q=True), dict(model="random", use_symmetry=False, parent_q=False)]
    w, a, n = parity.check_match(lib, "go9", agents, 6, concurrent_games=4, num_traversals=64, max_plies=200, seed=21)
    assert (n > 20).all()
    parity.check_match(lib, "go9", agents[::-1], 3, concurrent_games=2, num_traversals=8, max_batch=4, max_queue=2, max_plies=200, seed=22)
    parity.check_match(lib, "go19", agents, 1, concurrent_games=1, num_traversals=32, max_plies=740, seed=23)


def test_match_two_network_agents_toy_forward(lib):
    """Both agents evaluate through forward hooks on DEVICE buffers; the dense batch is split per agent."""
    import torch
    A = 65

    class _Arr:
        def __init__(self, ptr, shape):
            self.__cuda_array_interface__ = dict(shape=shape, typestr="<f4", data=(ptr, False), version=2)

    def make(scale):
        def fwd(planes_ptr, batch, logits_ptr, value_ptr):
            x = torch.as_tensor(_Arr(planes_ptr, (batch, 3, 8, 8)), device="cuda").cpu().numpy()
            lo, va = parity.toy_forward_numpy(x, A)
            torch.as_tensor(_Arr(logits_ptr, (batch, A)), device="cuda").copy_(torch.from_numpy(lo * np.float32(scale)))
            torch.as_tensor(_Arr(value_ptr, (batch,)), device="cuda").copy_(torch.from_numpy(va))
            torch.cuda.synchronize()
            return 0

        def ofwd(x):
            lo, va = parity.toy_forward_numpy(x, A)
            return lo * np.float32(scale), va
        return fwd, po.make_forward(ofwd, po.GAME_OTHELLO)

    e0, o0 = make(1.0)
    e1, o1 = make(0.25)
    agents = [dict(model="net", use_symmetry=True, parent_q=True), dict(model="net", use_symmetry=False, parent_q=True)]
    parity.check_match(lib, "othello", agents, 3, concurrent_games=3, num_traversals=24, forwards=(e0, e1),
                       oracle_forwards=(o0, o1))


def test_match_traced_models(lib, traced_model, tmp_path):
    """Two traced CNNs through the LibTorch-ROCm plugin: games are legal and complete, colours alternate."""
    from sprl_amd.network import make_network, trace_to_file
    other = trace_to_file(make_network("othello", 1, 32, seed=5), str(tmp_path / "traced_b.pt"), "othello")
    cfg = parity.match_config(lib, "othello", concurrent_games=32, num_traversals=48, seed=3)
    w, actions, n = E.play_match(cfg, dict(model=traced_model), dict(model=other, parent_q=False), 48, lib=lib)
    assert (n >= 9).all() and (n <= 128).all()
    assert sum(E.match_score(w)) == 48
    for g in range(48):                      # replay every game with the oracle's rules: every move legal, same end
        assert po.replay_winner(po.GAME_OTHELLO, actions[g, :n[g]]) == w[g]


# ---- engines on private HIP streams, driven from several host threads ----

def test_two_engines_on_private_streams_random_evaluator(lib):
    """Two engines, own_stream=1, one host thread each, running at the same time: both bit-exact against the oracle."""
    import threading
    out = {}

    def work(k, seed):
        cfg = E.default_config("othello", lib, concurrent_games=64, num_traversals=48, seed=seed, stream_base=1 + 1000 * k,
                               own_stream=1)
        eng = E.Engine(cfg, lib)
        eng.set_model("random")
        out[k] = (cfg, eng.run(64), eng.stats())
        eng.close()

    ths = [threading.Thread(target=work, args=(k, 11 + k)) for k in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for k in range(2):
        cfg, rec, st = out[k]
        ora = po.selfplay(parity.oracle_config("othello", cfg, po.EVAL_RANDOM), 64, 11 + k, cfg.stream_base, True)
        parity.assert_same_games(rec, ora)
        parity.assert_same_counters(st, ora["stats"])


def test_private_stream_cnn_games_equal_null_stream_games(lib, traced_model):
    """The hand-written CNN gives every board the same bits whatever batch it sits in, so an engine on a private stream
    (rounds enqueued without host synchronisation) plays exactly the games of an engine on the null stream."""
    recs = []
    for own in (0, 1):
        cfg = E.default_config("othello", lib, concurrent_games=32, num_traversals=40, seed=5, own_stream=own)
        eng = E.Engine(cfg, lib)
        eng.set_model(traced_model)
        rec = eng.run(32)
        recs.append((rec.expand_boards(), rec.expand()))
        eng.close()
    (b0, p0), (s0, d0, o0) = recs[0]
    (b1, p1), (s1, d1, o1) = recs[1]
    assert (b0 == b1).all() and (p0 == p1).all() and (o0 == o1).all()
    assert (d0.view(np.uint32) == d1.view(np.uint32)).all()


def test_private_stream_go9(lib, tmp_path):
    """Boards wider than 8 on a private stream: the whole hand-written forward (batch size read on the device) runs on the
    engine's own stream.  Same configuration, same rounds, same batches: the games of a private-stream engine equal those of a
    null-stream engine, also when two of them run from two threads."""
    import threading
    from sprl_amd.network import make_network, trace_to_file
    model = trace_to_file(make_network("go9", 2, 64, seed=6), str(tmp_path / "traced_go9s.pt"), "go9")

    def play(own, out, key, seed):
        cfg = E.default_config("go9", lib, concurrent_games=8, num_traversals=32, seed=seed, own_stream=own)
        eng = E.Engine(cfg, lib)
        eng.set_model(model)
        rec = eng.run(8)
        out[key] = (rec.expand_boards(), rec.expand())
        eng.close()

    res = {}
    play(0, res, "null5", 5)
    play(0, res, "null6", 6)
    play(1, res, "own5", 5)
    ths = [threading.Thread(target=play, args=(1, res, f"thr{sd}", sd)) for sd in (5, 6)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for a_key, b_key in (("null5", "own5"), ("null5", "thr5"), ("null6", "thr6")):
        (b0, p0), (s0, d0, o0) = res[a_key]
        (b1, p1), (s1, d1, o1) = res[b_key]
        assert b0.shape == b1.shape and (b0 == b1).all() and (p0 == p1).all() and (o0 == o1).all(), (a_key, b_key)
        assert (d0.view(np.uint32) == d1.view(np.uint32)).all(), (a_key, b_key)


@pytest.mark.parametrize("game,games,trav", [("go9", 8, 32), ("go19", 4, 24)])
def test_wide_board_forward_on_device_count_equals_host_count(lib, tmp_path, monkeypatch, game, games, trav):
    """boards wider than 8 no longer need the batch size on the host.  The same games with the round loop
    synchronising every round (SPRL_SYNC_ROUNDS=1: count read back, exact grids) and with the count left on the device
    (capacity-sized grids, workgroups past the count leave) - same kernels, same rows: bit-identical records."""
    from sprl_amd.network import make_network, trace_to_file
    model = trace_to_file(make_network(game, 2, 64, seed=8), str(tmp_path / f"traced_{game}_dev.pt"), game)

    def play(own_stream=0):
        cfg = E.default_config(game, lib, concurrent_games=games, num_traversals=trav, seed=4, own_stream=own_stream)
        eng = E.Engine(cfg, lib)
        eng.set_model(model)
        info = eng.evaluator_info()
        rec = eng.run(games)
        out = (rec.expand_boards(), rec.pdfs.copy(), rec.winners.copy(), info, eng.stats())
        eng.close()
        return out

    dev = play()
    monkeypatch.setenv("SPRL_SYNC_ROUNDS", "1")
    host = play()
    monkeypatch.delenv("SPRL_SYNC_ROUNDS")
    assert "read on the device" in dev[3] and "on the host" in host[3], (dev[3], host[3])
    assert (dev[0][0] == host[0][0]).all() and (dev[0][1] == host[0][1]).all() and (dev[2] == host[2]).all()
    assert (dev[1].view(np.uint32) == host[1].view(np.uint32)).all()
    assert dev[4]["nn_evals"] == host[4]["nn_evals"] and dev[4]["nn_evals"] > 0
    if game == "go9":
        # lab switch SPRL_TREE_STREAM (measured and left off): tree / scan / gather launches on a
        # second, high-priority stream of the engine, ordered against the forward by two events - the same games, bit for bit
        monkeypatch.setenv("SPRL_TREE_STREAM", "2")
        split = play(own_stream=1)                          # (the switch needs an engine with a stream of its own)
        monkeypatch.delenv("SPRL_TREE_STREAM")
        assert "SPRL_TREE_STREAM=2" in split[3], split[3]
        assert (dev[0][0] == split[0][0]).all() and (dev[0][1] == split[0][1]).all() and (dev[2] == split[2]).all()
        assert (dev[1].view(np.uint32) == split[1].view(np.uint32)).all() and dev[4]["nn_evals"] == split[4]["nn_evals"]


def test_go9_plugin_forward_matches_torchscript(lib, tmp_path):
    """9x9 and 19x19 boards, 64-channel trunk: the whole forward in hand-written kernels (plugin kind 2, asserted - a silent
    fall-back to the library convolutions would not count): NCHW stem for 17 planes on the matrix cores, any-board
    Winograd/MFMA trunk, 1x1 heads, FC tail; against the plain TorchScript fp32 forward of the same file."""
    import ctypes as C
    import torch
    from sprl_amd.network import make_network, trace_to_file
    model = trace_to_file(make_network("go9", 6, 64, seed=4), str(tmp_path / "traced_go9b.pt"), "go9")   # 6 blocks: go_controller.py:44
    plug = C.CDLL(os.path.join(os.path.dirname(E.DEFAULT_LIB), "libsprl_amd_torch.so"))
    plug.sprl_torch_load.restype = C.c_void_p
    plug.sprl_torch_load.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    plug.sprl_torch_forward.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p,
                                                                                 C.c_char_p, C.c_int]
    plug.sprl_torch_is_native.argtypes = [C.c_void_p]
    err = C.create_string_buffer(512)
    h = plug.sprl_torch_load(model.encode(), 0, err, 512)
    assert h, err.value
    assert plug.sprl_torch_is_native(h) == 2
    ref = torch.jit.load(model, map_location="cuda").eval()
    for batch in (5, 1024):
        x = (torch.rand(batch, 17, 9, 9, device="cuda") > 0.6).float().contiguous()
        lo = torch.zeros(batch, 82, device="cuda")
        va = torch.zeros(batch, device="cuda")
        assert plug.sprl_torch_forward(h, x.data_ptr(), batch, 17, 9, 9, lo.data_ptr(), 82, va.data_ptr(), err, 512) == 0, err.value
        torch.cuda.synchronize()
        with torch.no_grad():
            rl, rv = ref(x)
        np.testing.assert_allclose(lo.cpu().numpy(), rl.cpu().numpy(), atol=CNN_ATOL, rtol=0)
        np.testing.assert_allclose(va.cpu().numpy(), rv.cpu().numpy().reshape(-1), atol=CNN_ATOL, rtol=0)
    model19 = trace_to_file(make_network("go19", 6, 64, seed=6), str(tmp_path / "traced_go19.pt"), "go19")
    h19 = plug.sprl_torch_load(model19.encode(), 0, err, 512)
    assert h19, err.value
    assert plug.sprl_torch_is_native(h19) == 2
    ref19 = torch.jit.load(model19, map_location="cuda").eval()
    x = (torch.rand(37, 17, 19, 19, device="cuda") > 0.6).float().contiguous()
    lo = torch.zeros(37, 362, device="cuda")
    va = torch.zeros(37, device="cuda")
    assert plug.sprl_torch_forward(h19, x.data_ptr(), 37, 17, 19, 19, lo.data_ptr(), 362, va.data_ptr(), err, 512) == 0, err.value
    torch.cuda.synchronize()
    with torch.no_grad():
        rl, rv = ref19(x)
    np.testing.assert_allclose(lo.cpu().numpy(), rl.cpu().numpy(), atol=CNN_ATOL, rtol=0)
    np.testing.assert_allclose(va.cpu().numpy(), rv.cpu().numpy().reshape(-1), atol=CNN_ATOL, rtol=0)


def test_go_shape_cnn_against_reference_golden(lib, golden, tmp_path):
    """the networks BASELINE configs 4 / 5 run - BasicGridNetwork(9, 9, 82, 8, 6, 64) and (19, 19, 362, 8, 6, 64)
    (scripts/go_controller.py:44-45, src/networks/grid_networks.py:30-79): 17 planes, SIX residual blocks = twelve Winograd
    convolutions deep - pinned to outputs of the REFERENCE's own module (CPU fp32; tests/golden/g9c_go_networks.npz from
    gen_golden.py g9c).  The same seed-reproducible weights go into our module, are traced as the controller does, and run
    through the hand-written path (plugin kind 2: NCHW MFMA stem, F(3x3,3x3) trunk at 9x9 / F(4x4,3x3) at 19x19, tail kernels)
    at two weight gains (|logits| <= 0.35 / 15.7 at 9x9, 0.1 / 4.4 at 19x19); tolerance cnn_tol as for the 8x8 golden."""
    import ctypes as C
    import sys
    import torch
    from sprl_amd.network import GridResNet, trace_to_file
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import netfill
    g = golden("g9c_go_networks.npz")
    seed = int(g["seed"][0])
    plug = _plugin()
    err = C.create_string_buffer(512)
    report = []
    for width, actions in ((9, 82), (19, 362)):
        n = int(g[f"n{width}"][0])
        x = torch.from_numpy(netfill.go_like_inputs(n, width, 8, seed + width)).cuda().contiguous()
        for gi, gain in enumerate(g["gains"]):
            net = netfill.fill_state_dict(GridResNet(width, width, actions, 8, 6, 64), seed + 100 * width + gi, float(gain)).eval()
            path = trace_to_file(net, str(tmp_path / f"g9c_{width}_{gi}.pt"), "go9" if width == 9 else "go19")
            h = plug.sprl_torch_load(path.encode(), 0, err, 512)
            assert h and plug.sprl_torch_is_native(h) == 2, err.value
            lo, va = _plugin_forward(plug, h, x, actions)
            assert _path_info(plug, h).endswith("lab=[]")
            plug.sprl_torch_free(h)
            rl, rv = g[f"logits{width}_{gi}"], g[f"value{width}_{gi}"].reshape(-1)
            e_ref = max(np.abs(lo - rl).max(), np.abs(va - rv).max())
            e_f64 = max(np.abs(lo - g[f"logits{width}_f64_{gi}"]).max(), np.abs(va - g[f"value{width}_f64_{gi}"].reshape(-1)).max())
            cpu_f64 = max(np.abs(rl - g[f"logits{width}_f64_{gi}"]).max(), np.abs(rv - g[f"value{width}_f64_{gi}"].reshape(-1)).max())
            scale = float(np.abs(rl).max())
            report.append(f"{width}x{width}, 6 blocks, gain {gain}: |logits| <= {scale:.2f}; hand-written vs reference fp32 {e_ref:.3e} "
                          f"(= {e_ref / scale:.2e} of the scale), vs float64 {e_f64:.3e}; reference fp32 vs float64 {cpu_f64:.3e}; "
                          f"tolerance {cnn_tol(rl):.1e}")
            assert e_ref < cnn_tol(rl), report[-1]
    print("\n".join(report))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "cnn_error_go.txt"), "w").write("\n".join(report) + "\n")


def test_go9_with_traced_cnn_generic_path(lib, tmp_path):
    """A trunk that is not 64 channels wide (here 1 block x 32) has no hand-written Winograd kernel: the recognised architecture
    runs as library convolutions + the hand-written fused epilogue (plugin kind 1).  Games must be complete and legal-looking,
    policies normalised."""
    from sprl_amd.network import make_network, trace_to_file
    model = trace_to_file(make_network("go9", 1, 32, seed=2), str(tmp_path / "traced_go9.pt"), "go9")
    cfg, rec, st = parity.run_engine(lib, "go9", 6, model=model, concurrent_games=6, num_traversals=40)
    states, dists, outcomes = rec.expand()
    assert st["games"] == 6 and st["nn_evals"] > 0
    assert states.shape[1:] == (17, 9, 9) and dists.shape[1] == 82
    assert np.allclose(dists.sum(1), 1.0, atol=1e-4)
    assert set(np.unique(outcomes).tolist()) <= {-1.0, 0.0, 1.0}


def test_resign_threshold_on_device(lib):
    """The resign extension (default off; BASELINE config 5 names it): device == oracle, single-strip and wide kernels."""
    parity.check_case(lib, "othello", 8, concurrent_games=8, num_traversals=64, seed=31, resign_threshold=0.05, resign_min_ply=4)
    parity.check_case(lib, "go9", 4, concurrent_games=4, num_traversals=48, seed=31, resign_threshold=0.05, resign_min_ply=4)
    rec, _ = parity.check_case(lib, "go19", 2, concurrent_games=2, num_traversals=24, seed=31, resign_threshold=0.02,
                               resign_min_ply=6)
    assert (rec.ply_offset[1:] - rec.ply_offset[:-1] < 722).all()


def test_device_side_records_pack_and_expand(lib):
    """records_kernel.h on the device: the packed wire format (what ranks hand RCCL) and the expanded training samples,
    written into torch CUDA tensors, equal the host paths of the same run - Othello and Go 9x9 (two-word bit sets)."""
    import torch
    from sprl_amd.distributed import pack_records, unpack_records
    for game, trav in (("othello", 48), ("go9", 40)):
        cfg = E.default_config(game, lib, concurrent_games=8, num_traversals=trav, seed=4)
        eng = E.Engine(cfg, lib)
        eng.set_model("random")
        eng.begin(8)
        done = 0
        while done < 8:
            done, _ = eng.step(64)
        plies, samples, nbytes = eng.records_info()
        shard = torch.full((nbytes,), 0xAB, dtype=torch.uint8, device="cuda")
        eng.pack_records_into(shard.data_ptr(), nbytes)
        planes = 17 if game == "go9" else 3
        side = 9 if game == "go9" else 8
        A = 82 if game == "go9" else 65
        states = torch.full((samples, planes, side, side), float("nan"), device="cuda")
        dists = torch.full((samples, A), float("nan"), device="cuda")
        outs = torch.full((samples,), float("nan"), device="cuda")
        eng.expand_records_into(states.data_ptr(), dists.data_ptr(), outs.data_ptr(), samples)
        torch.cuda.synchronize()
        rec = eng.collect()
        want = pack_records(rec)
        assert want.size == nbytes and (shard.cpu().numpy() == want).all()
        u = unpack_records(shard.cpu().numpy())
        assert (u["boards"] == rec.boards).all() and u["words"] == (rec.cells + 63) // 64
        s1, d1, o1 = rec.expand()
        assert (states.cpu().numpy() == s1).all() and (outs.cpu().numpy() == o1).all()
        assert (dists.cpu().numpy().view(np.uint32) == d1.view(np.uint32)).all()
        eng.close()


def test_native_worker_two_iterations_with_model_handover(lib, tmp_path):
    """The native worker process (sprl_amd/sprl_worker) on the GPU through the reference's file protocol
    (GridWorker.hpp:35-55,111-197): iteration 0 with the built-in evaluator, then it polls for
    data/models/<run>/traced_<run>_iteration_0.pt, which a stand-in controller traces and drops while the worker is
    spinning; iteration 1 and 2 search with that CNN (one engine kept).  Files: reference names, shapes, header bytes;
    iteration 0 equals the oracle's games bit for bit."""
    import subprocess
    import threading
    import time
    from sprl_amd.network import make_network, trace_to_file
    exe = os.path.join(ROOT, "sprl_amd", "sprl_worker")
    assert os.path.exists(exe), "sprl_amd/sprl_worker not built (make -C sprl_amd/csrc)"
    run = "gpurun"
    models = tmp_path / "data" / "models" / run
    models.mkdir(parents=True)
    staged = trace_to_file(make_network("othello", 2, 64, seed=5), str(tmp_path / "staged.pt"), "othello")

    def controller():
        d = tmp_path / "data" / "games" / run / "0" / "0"
        for it in (0, 1):
            while not (d / f"{run}_iteration_{it}_outcomes.npy").exists():
                time.sleep(0.05)
            time.sleep(0.6)                                   # the worker is polling by now
            tmp = models / f"tmp_{it}.pt"
            tmp.write_bytes(open(staged, "rb").read())
            os.replace(tmp, models / f"traced_{run}_iteration_{it}.pt")

    th = threading.Thread(target=controller)
    th.start()
    out = subprocess.run([exe, "othello", "0", "2", "--cover", "2", "--num-tasks-const", "2", "--num-groups", "1", "--num-iters", "3",
                          "--init-games", "2", "--init-traversals", "64", "--init-max-batch", "8", "--init-max-queue", "4",
                          "--games", "3", "--traversals", "48", "--seed", "31", "--root", str(tmp_path), "--run-name", run,
                          "--poll-seconds", "0.3"], capture_output=True, text=True, timeout=600)
    th.join()
    assert out.returncode == 0, out.stderr + out.stdout
    assert out.stdout.count("Spinning on traced model") >= 2 and "Using traced PyTorch network..." in out.stdout
    for task in (0, 1):
        d = tmp_path / "data" / "games" / run / "0" / str(task)
        for it in (0, 1, 2):
            s = np.load(d / f"{run}_iteration_{it}_states.npy")
            p = np.load(d / f"{run}_iteration_{it}_distributions.npy")
            o = np.load(d / f"{run}_iteration_{it}_outcomes.npy")
            assert s.dtype == np.float32 and s.shape[1:] == (3, 8, 8) and p.shape == (s.shape[0], 65) and o.shape == (s.shape[0],)
            assert np.allclose(p.sum(1), 1.0, atol=1e-4) and set(np.unique(o).tolist()) <= {-1.0, 0.0, 1.0}
    ora = po.selfplay(po.make_config(po.GAME_OTHELLO, 64, math_mode=po.MATH_PORTABLE), 4, 31, 1, True)
    split = ora["offsets"][2]
    d0 = np.load(tmp_path / f"data/games/{run}/0/0/{run}_iteration_0_distributions.npy")
    d1 = np.load(tmp_path / f"data/games/{run}/0/1/{run}_iteration_0_distributions.npy")
    assert (d0.view(np.uint32) == ora["dists"][:split].view(np.uint32)).all()
    assert (d1.view(np.uint32) == ora["dists"][split:].view(np.uint32)).all()


def test_native_worker_two_populations_with_cnn(lib, tmp_path):
    """sprl_worker --populations 2 on the GPU with the traced CNN from iteration 0 on: two engines on private streams driven
    from two host threads.  Population 0's task files equal those of a single-engine worker covering the same tasks with the
    same seed (the hand-written CNN gives a board the same bits in any batch), population 1's files are complete and valid."""
    import subprocess
    from sprl_amd.network import make_network, trace_to_file
    exe = os.path.join(ROOT, "sprl_amd", "sprl_worker")
    model = trace_to_file(make_network("othello", 2, 64, seed=5), str(tmp_path / "m.pt"), "othello")
    common = ["othello", "0", "4", "--num-tasks-const", "4", "--num-groups", "2", "--num-iters", "1", "--init-games", "3",
              "--init-traversals", "48", "--init-max-batch", "8", "--init-max-queue", "4", "--seed", "41", "--model-iter0", model]
    a = subprocess.run([exe] + common + ["--cover", "2", "--root", str(tmp_path / "one"), "--run-name", "r"],
                       capture_output=True, text=True, timeout=300)
    assert a.returncode == 0, a.stderr + a.stdout
    b = subprocess.run([exe] + common + ["--cover", "4", "--populations", "2", "--root", str(tmp_path / "two"), "--run-name", "r"],
                       capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr + b.stdout
    for task in (0, 1):
        for part in ("states", "distributions", "outcomes"):
            fa = open(tmp_path / f"one/data/games/r/0/{task}/r_iteration_0_{part}.npy", "rb").read()
            fb = open(tmp_path / f"two/data/games/r/0/{task}/r_iteration_0_{part}.npy", "rb").read()
            assert fa == fb, (task, part)
    for task in (2, 3):
        d = tmp_path / f"two/data/games/r/1/{task}"
        s = np.load(d / "r_iteration_0_states.npy")
        p = np.load(d / "r_iteration_0_distributions.npy")
        o = np.load(d / "r_iteration_0_outcomes.npy")
        assert s.shape[1:] == (3, 8, 8) and p.shape == (s.shape[0], 65) and o.shape == (s.shape[0],) and s.shape[0] >= 3 * 8 * 8
        assert np.allclose(p.sum(1), 1.0, atol=1e-4)


def test_go19_full_budget_with_compaction_and_resign(lib):
    """BASELINE config 5 at its stated budget: Go 19x19, 1600 iterations/move, batch 16 / queue 8, whole games against the
    oracle.  First with node recycling off and small arenas (8000 nodes of 5.5 KiB), so the large-tree fallback - Cheney
    compaction into a spare arena - fires many times per game (games run to the double pass / the 722-ply cap, as in the
    reference); then as the engine runs by default (recycling, default arena of 4 x 1600 + 1024 nodes, no spare
    arenas needed) with the resign extension on."""
    # (one game in this leg: the oracle plays it too, on one host core, and the driver's GPU-test budget is shared with 86 others)
    rec, st = parity.check_case(lib, "go19", 1, concurrent_games=1, num_traversals=1600, node_cap=8000, spare_arenas=2, seed=5,
                                no_recycle=1)
    assert rec.cells == 361 and st["compactions"] >= 10 and st["max_nodes_in_arena"] <= 8000
    assert st["traversals"] >= 1600 * st["plies"]
    rec2, st2 = parity.check_case(lib, "go19", 2, concurrent_games=2, num_traversals=1600, spare_arenas=0, seed=6,
                                  resign_threshold=0.02, resign_min_ply=30)
    assert st2["compactions"] == 0 and st2["nodes_recycled"] > 0.9 * st2["nodes_created"] and st2["plies"] < 2 * st["plies"]
    assert st2["max_nodes_in_arena"] <= 4 * 1600 + 1024


# Thresholds of the distributional test: two-sample Kolmogorov-Smirnov p > KS_P_MIN on game length, on the
# per-game mean entropy of the root-visit pdfs and on the evaluations per game; means of the outcome (Player ZERO's reward), of
# the plies and of the network evaluations per game within MEAN_SIGMAS standard errors of the difference, every sample with
# its OWN variance.  Seeds are fixed, so the test is deterministic: the thresholds say how unlikely a failure would be for two
# samples of ONE distribution (about 1e-3 per statistic).
KS_P_MIN = 1e-3
MEAN_SIGMAS = 3.5


def test_cnn_games_distribution_matches_reference_fixture(lib, golden, tmp_path):
    """SURVEY section 7 (iii) / VERDICT r3 #4: with the hand-written fp32 forward the CNN games are not bit-comparable with the
    reference (last-bit differences of the logits move single visits), so the headline path is checked as a DISTRIBUTION - and
    with enough games to see a small bias.  tests/golden/g_cnn_dist.npz holds per-game statistics of 1024 games of the
    REFERENCE's own selfPlay + GridNetwork (LibTorch-CPU, oracle/_ref; selfplay/SelfPlay.hpp:51-192, networks/GridNetwork.hpp:62-145)
    generated once in the build container (gen_golden.py g_cnn_dist, 13 core-minutes): a traced 2x64 network with netfill
    weights at gain 2 (|logits| of a few units: the policy head shapes the search), Othello, 200 traversals/move, batch 8 /
    queue 4, D4, Dirichlet(0.25, 0.3).  The gfx950 engine plays 1024 games with the same network (other RNG streams) and reports
    evaluations PER GAME (sprl_engine_game_evals), so every statistic uses its own sample's variance.
    Detectable effect at these sizes (3.5 standard errors of the difference, both samples 1024 games): plies 0.26 (0.4 %),
    evaluations per game 44 (0.5 %), outcome 0.15, entropy 0.0086 (1.5 %)."""
    import sys
    from scipy import stats
    from sprl_amd.network import GridResNet, trace_to_file
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import netfill
    g = golden("g_cnn_dist.npz")
    ref = g["stats"]                                  # [games][plies, outcome of Player ZERO, evaluations, mean pdf entropy]
    trav, games = int(g["traversals"][0]), len(ref)
    net = netfill.fill_state_dict(GridResNet(8, 8, 65, 1, 2, 64), int(g["net_seed"][0]), float(g["gain"][0])).eval()
    model = trace_to_file(net, str(tmp_path / "dist.pt"), "othello")
    cfg = E.default_config("othello", lib, concurrent_games=games, num_traversals=trav, seed=int(g["rng_seed"][0]), stream_base=1)
    eng = E.Engine(cfg, lib)
    eng.set_model(model)
    info = eng.evaluator_info()
    rec = eng.run(games)
    st = eng.stats()
    evals = eng.game_evals(games).astype(np.float64)
    eng.close()
    assert "hand-written gfx950 CNN" in info and info.count("lab=[]") == 2, info
    assert evals.sum() == st["nn_evals"], (evals.sum(), st["nn_evals"])          # the per-game view adds up to the engine's total
    z0 = np.where(rec.winners == 0, 1.0, np.where(rec.winners == 1, -1.0, 0.0))
    gpu = parity.game_statistics(rec.ply_offset, rec.pdfs, z0, evals)
    cols = {"plies": 0, "outcome": 1, "evals": 2, "entropy": 3}
    lines = [f"reference fixture: {len(ref)} games; GPU: {games} games; {trav} traversals/move; network: netfill gain {float(g['gain'][0])}"]
    ok = True
    for name, c in cols.items():
        a, b = gpu[name], ref[:, c]
        se = np.sqrt(a.var(ddof=1) / len(a) + b.var(ddof=1) / len(b))
        d = abs(a.mean() - b.mean())
        ks = stats.ks_2samp(a, b).pvalue if name != "outcome" else 1.0
        lines.append(f"{name:8s} gpu {a.mean():10.4f} +- {a.std():8.4f}   ref {b.mean():10.4f} +- {b.std():8.4f}   |diff| = {d:.4f} = {d / se:.2f} "
                     f"standard errors (detectable at {MEAN_SIGMAS} s.e.: {MEAN_SIGMAS * se:.4f} = {100 * MEAN_SIGMAS * se / max(1e-9, abs(b.mean())):.2f} %)"
                     + (f"   KS p = {ks:.3f}" if name != "outcome" else ""))
        ok = ok and d < MEAN_SIGMAS * se and ks > KS_P_MIN
    print("\n".join(lines))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "cnn_distribution.txt"), "w").write("\n".join(lines) + "\n")
    assert ok, lines


def test_trainer_fixture_on_the_gpu(golden):
    """f-2 on the MI355X: the replay of the REFERENCE controller's training call (g_trainer.npz: its batches, its
    best epoch, its exported weights) with window, network and optimiser on cuda:0.  The library's GPU convolution / reduction
    order differs from the CPU's, and 12 epochs of AdamW carry that along: best epoch and stopping epoch must be the reference's
    exactly, the outputs of the best / live network are compared at 1e-2 (measured on MI355X: 3.0e-3, gpurun_out/trainer_step.txt;
    the CPU replay of the same fixture agrees to 1e-5).  Also times one optimiser step of the BASELINE-shape
    network (2 x 64, batch 1024 = othello_controller.py:52) with the window resident in HBM."""
    import time
    import torch
    from sprl_amd import trainer as T
    from sprl_amd.network import GridResNet
    dev = parity.replay_trainer_fixture(golden("g_trainer.npz"), "cuda:0", atol=1e-2)
    # training-step time, BASELINE shape: losses stay on the device, one host sync per epoch
    torch.manual_seed(0)
    n, bs = 64 * 1024, 1024
    s = (torch.rand(n, 3, 8, 8, device="cuda") > 0.5).float()
    d = torch.softmax(torch.randn(n, 65, device="cuda"), 1)
    o = torch.sign(torch.randn(n, 1, device="cuda"))
    t = torch.ones(n, 1, device="cuda")
    net = GridResNet(8, 8, 65, 1, 2, 64)
    cfg = T.TrainerConfig(batch_size=bs, max_groups=1, epochs_per_group=2)
    T.train_network(net, 0.01, (s[:4 * bs], d[:4 * bs], o[:4 * bs], t[:4 * bs]), cfg)           # warm-up (kernel selection)
    torch.cuda.synchronize()
    t0 = time.time()
    best, hist = T.train_network(net, 0.01, (s, d, o, t), cfg)
    torch.cuda.synchronize()
    dt = time.time() - t0
    steps = 2 * ((int(0.9 * n) + bs - 1) // bs)
    line = (f"trainer fixture on cuda:0: max deviation from the reference controller's outputs {dev:.2e}; BASELINE-shape training: "
            f"{steps} optimiser steps of batch {bs} + 2 validation passes in {dt:.2f} s = {1e3 * dt / steps:.2f} ms/step, "
            f"{steps * bs / dt:.0f} samples/s")
    print(line)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "trainer_step.txt"), "w").write(line + "\n")
    assert all(np.isfinite(e["train_policy"]) for e in hist["epochs"])
