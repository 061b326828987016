"""The LibTorch evaluator plugin's load-time graph rewrite (conv bias hoisted into a fusible add) keeps the
network function: plugin forward on host tensors (device -1, test-only mode) vs the plain TorchScript forward."""
import ctypes as C
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from sprl_amd import engine as E
from sprl_amd.network import make_network, trace_to_file


def test_rewritten_graph_matches_original(tmp_path):
    path = trace_to_file(make_network("othello", 2, 16, seed=3), str(tmp_path / "m.pt"), "othello")
    plug = C.CDLL(os.path.join(os.path.dirname(E.DEFAULT_LIB), "libsprl_amd_torch.so"))
    plug.sprl_torch_load.restype = C.c_void_p
    plug.sprl_torch_load.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    plug.sprl_torch_forward.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p,
                                                                                 C.c_char_p, C.c_int]
    plug.sprl_torch_free.argtypes = [C.c_void_p]
    err = C.create_string_buffer(512)
    h = plug.sprl_torch_load(path.encode(), -1, err, 512)
    assert h, err.value
    rng = np.random.default_rng(0)
    for batch in (1, 7, 64):
        x = (rng.random((batch, 3, 8, 8)) > 0.6).astype(np.float32)
        lo = np.zeros((batch, 65), np.float32)
        va = np.zeros(batch, np.float32)
        for _ in range(3):          # profiling executor: the optimised/fused graph kicks in after warm-up runs
            assert plug.sprl_torch_forward(h, x.ctypes.data, batch, 3, 8, 8, lo.ctypes.data, 65, va.ctypes.data, err, 512) == 0, err.value
        ref = torch.jit.load(path).eval()
        with torch.no_grad():
            rl, rv = ref(torch.from_numpy(x))
        np.testing.assert_allclose(lo, rl.numpy(), atol=2e-5, rtol=0)
        np.testing.assert_allclose(va, rv.numpy().reshape(-1), atol=2e-5, rtol=0)
    plug.sprl_torch_free(h)


def test_winograd_weight_transform_and_algebra_on_cpu():
    """The host half of the hand-written trunk convolution, without a GPU: the plugin's weight transform (U = G g G^T,
    stored in the kernel's A-operand lane order) fed through a numpy restatement of what the kernel computes
    (V = B^T d B per 6x6 patch, 36 position-wise [64 x 64] products, Y = A^T M A) must equal the direct 3x3 convolution."""
    plug = C.CDLL(os.path.join(os.path.dirname(E.DEFAULT_LIB), "libsprl_amd_torch.so"))
    plug.sprl_wino_transform_weights.argtypes = [C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5)
    w = (rng.standard_normal((64, 64, 3, 3)) * 0.1).astype(np.float32)
    x = rng.standard_normal((2, 64, 8, 8)).astype(np.float32)
    u = np.zeros(36 * 64 * 64, np.float32)
    plug.sprl_wino_transform_weights(w.ctypes.data, u.ctypes.data)
    # undo the lane order: U4[p / 4][s][kb][lane][p % 4], lane = slot * 16 + k % 16, input channel = 16 (s >> 2) + 4 slot + (s & 3)
    U = np.zeros((36, 64, 64), np.float64)                     # [p][k][c]
    u4 = u.reshape(9, 16, 4, 64, 4)
    for p in range(36):
        for s in range(16):
            for kb in range(4):
                for lane in range(64):
                    k = 16 * kb + lane % 16
                    c = 16 * (s >> 2) + 4 * (lane // 16) + (s & 3)
                    U[p, k, c] = u4[p // 4, s, kb, lane, p % 4]
    BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                   [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64)
    AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)
    xp = np.zeros((2, 64, 10, 10))
    xp[:, :, 1:9, 1:9] = x
    y = np.zeros((2, 64, 8, 8))
    for n in range(2):
        for ty in range(2):
            for tx in range(2):
                d = xp[n, :, 4 * ty:4 * ty + 6, 4 * tx:4 * tx + 6]                  # [c][6][6]
                V = np.einsum("ai,cij,bj->abc", BT, d, BT).reshape(36, 64)        # [p][c]
                M = np.einsum("pkc,pc->pk", U, V).reshape(6, 6, 64)               # [xi][nu][k]
                y[n, :, 4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4] = np.einsum("ia,abk,jb->kij", AT, M, AT)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), padding=1).numpy()
    np.testing.assert_allclose(y, ref, atol=2e-5, rtol=0)     # fp32-rounded U against exact weights
