"""The hand-written trunk convolution (sprl_amd/csrc/cnn_wino.hip: Winograd F(4x4,3x3) on fp32 MFMA with fused
scale/shift/residual/ReLU) against a float64 conv2d on the same inputs.  Tolerance: 8e-5 absolute on outputs of magnitude
O(1-10) with N(0,1) inputs in all 64 channels - twice the 3.7e-5 measured by tools/wino_lab.hip on 14 400 such boards
(F(4x4,3x3) in fp32 carries ~1e-5 relative error through its transforms); real network activations: tests/test_gpu_parity.py."""
import ctypes as C
import os

import numpy as np
import pytest

from sprl_amd import engine as E

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def plug():
    E.load_library()
    p = C.CDLL(os.path.join(os.path.dirname(E.DEFAULT_LIB), "libsprl_amd_torch.so"))
    p.sprl_wino_conv64.argtypes = [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p]
    p.sprl_wino_transform_weights.argtypes = [C.c_void_p, C.c_void_p]
    p.sprl_wino_conv64_nchw.argtypes = [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p]
    p.sprl_wino_conv64_nchw_tiled.argtypes = [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]
    p.sprl_wino_transform_weights_f3.argtypes = [C.c_void_p, C.c_void_p]
    p.sprl_wino_conv64_t.argtypes = [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]
    p.sprl_wino_transform_weights_t.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    p.sprl_stem_conv3x3_t.argtypes = [C.c_void_p] * 5 + [C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return p


def to_w(x):
    """NCHW [n][64][H][W] -> layout W [n][g][i][cs][tile][j] (4096 floats per board, zeros off the board)."""
    import torch
    n, _, H, W = x.shape
    p = torch.zeros(n, 64, 8, 8, device=x.device, dtype=x.dtype)
    p[:, :, :H, :W] = x
    p = p.reshape(n, 4, 4, 4, 2, 4, 2, 4)            # n, kb, cs, kr, ty, i, tx, j   (k = 16 kb + 4 cs + kr)
    return p.permute(0, 1, 3, 5, 2, 4, 6, 7).contiguous().reshape(n, 4096)


def from_w(y, H, W):
    n = y.shape[0]
    p = y.reshape(n, 4, 4, 4, 4, 2, 2, 4)            # n, kb, kr, i, cs, ty, tx, j
    p = p.permute(0, 1, 4, 2, 5, 3, 6, 7).contiguous().reshape(n, 64, 8, 8)
    assert (p[:, :, H:, :] == 0).all() and (p[:, :, :, W:] == 0).all(), "cells off the board must be zero"
    return p[:, :, :H, :W]


def run_conv(plug, x, w, scale, shift, res, relu):
    import torch
    u = np.zeros(36 * 64 * 64, np.float32)
    wc = np.ascontiguousarray(w.cpu().numpy())
    plug.sprl_wino_transform_weights(wc.ctypes.data, u.ctypes.data)
    ud = torch.from_numpy(u).cuda()
    B, _, H, W = x.shape
    xw = to_w(x)
    rw = to_w(res) if res is not None else None
    yw = torch.full_like(xw, float("nan"))
    rc = plug.sprl_wino_conv64(xw.data_ptr(), ud.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                               rw.data_ptr() if rw is not None else None, yw.data_ptr(), B, H, W, int(relu), None)
    assert rc == 0
    torch.cuda.synchronize()
    return from_w(yw, H, W)


@pytest.mark.parametrize("H,W,B", [(8, 8, 8), (8, 8, 13), (8, 8, 1024), (6, 7, 37), (7, 7, 64), (8, 8, 1)])
def test_wino_conv_matches_conv2d(plug, H, W, B):
    import torch
    torch.manual_seed(H * 100 + W * 10 + B)
    x = torch.randn(B, 64, H, W, device="cuda")
    w = torch.randn(64, 64, 3, 3, device="cuda") * 0.06
    scale = torch.rand(64, device="cuda") + 0.5
    shift = torch.randn(64, device="cuda") * 0.3
    res = torch.randn(B, 64, H, W, device="cuda")
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    for use_res, relu in ((False, True), (True, True), (False, False)):
        want = ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
        if use_res:
            want = want + res.double()
        if relu:
            want = torch.relu(want)
        got = run_conv(plug, x, w, scale, shift, res if use_res else None, relu)
        err = (got.double() - want).abs().max().item()
        assert err < 8e-5, (H, W, B, use_res, relu, err)


def test_wino_structured_inputs(plug):
    """One-hot inputs and one-hot filters: every (tap, border) combination lands where conv2d puts it."""
    import torch
    x = torch.zeros(9, 64, 8, 8, device="cuda")
    w = torch.zeros(64, 64, 3, 3, device="cuda")
    for n in range(9):
        x[n, (7 * n) % 64, (3 * n) % 8, (5 * n + 1) % 8] = 1.0 + n
    for k in range(64):
        w[k, (k * 5) % 64, k % 3, (k // 3) % 3] = 1.0
        w[k, (k * 11 + 1) % 64, (k + 1) % 3, (k // 5) % 3] = -0.5
    one, zero = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
    got = run_conv(plug, x, w, one, zero, None, False)
    want = torch.nn.functional.conv2d(x, w, padding=1)
    assert (got - want).abs().max().item() < 1e-5


@pytest.mark.parametrize("tile", [4, 3])
@pytest.mark.parametrize("H,W,B", [(9, 9, 7), (19, 19, 3), (8, 8, 5), (5, 7, 9), (9, 9, 300), (19, 19, 40), (13, 6, 11)])
def test_wino_conv_general_boards_nchw(plug, H, W, B, tile):
    """The any-board variant (workgroup = 16 tiles, patches gathered from NCHW) against conv2d in float64, in both tilings
    (F(4x4,3x3): 36 positions per 4x4 tile; F(3x3,3x3): 25 positions per 3x3 tile - the 9x9 choice) on every board, with the
    board count given by the host and read from device memory (capacity-sized grid)."""
    import torch
    torch.manual_seed(H * 1000 + W * 10 + B)

    def act():
        """[B][64][H][W] with the readable bytes around it that the kernel asks for (patch rows are fetched 16 + 8 bytes at a
        time); the slack holds NaN: nothing read from it may reach a result."""
        n = B * 64 * H * W
        flat = torch.full((n + 12,), float("nan"), device="cuda")      # 16 bytes in front as well (left neighbour of column 0)
        flat[4:n + 4] = torch.randn(n, device="cuda")
        return flat[4:n + 4].view(B, 64, H, W)

    x = act()
    w = torch.randn(64, 64, 3, 3, device="cuda") * 0.06
    scale = torch.rand(64, device="cuda") + 0.5
    shift = torch.randn(64, device="cuda") * 0.3
    res = act()
    u = np.zeros((36 if tile == 4 else 28) * 64 * 64, np.float32)
    wc = np.ascontiguousarray(w.cpu().numpy())          # keep the host copy alive across the call
    (plug.sprl_wino_transform_weights if tile == 4 else plug.sprl_wino_transform_weights_f3)(wc.ctypes.data, u.ctypes.data)
    ud = torch.from_numpy(u).cuda()
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    for use_res, relu in ((True, True), (False, False)):
        want = ref + res.double() if use_res else ref
        if relu:
            want = torch.relu(want)
        y = torch.full_like(x, float("nan"))
        rc = plug.sprl_wino_conv64_nchw_tiled(x.data_ptr(), ud.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                              res.data_ptr() if use_res else None, y.data_ptr(), B, H, W, int(relu), tile, None, None)
        assert rc == 0
        torch.cuda.synchronize()
        err = (y.double() - want).abs().max().item()
        assert err < 8e-5, (H, W, B, tile, use_res, relu, err)
        # the count on the device: only the first `live` boards are computed, nothing behind them is written
        live = max(1, B // 2)
        cnt = torch.tensor([live], dtype=torch.int32, device="cuda")
        y2 = torch.full_like(x, float("nan"))
        rc = plug.sprl_wino_conv64_nchw_tiled(x.data_ptr(), ud.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                              res.data_ptr() if use_res else None, y2.data_ptr(), B, H, W, int(relu), tile,
                                              cnt.data_ptr(), None)
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(y2[:live], y[:live]) and torch.isnan(y2[live:]).all()


def to_layout_t(x, M):
    """[B][64][H][W] -> layout T of the any-board trunk kernel: [q = 16][cell = M*M][n * tiles + tile][4 channels] (cnn_wino.hip;
    the tile index runs over the batch)."""
    import torch
    B, _, H, W = x.shape
    TY, TX = (H + M - 1) // M, (W + M - 1) // M
    xp = torch.zeros(B, 64, TY * M, TX * M, device=x.device, dtype=x.dtype)
    xp[:, :, :H, :W] = x
    # [B][q][e][ty][i][tx][j] -> [q][i][j][B][ty][tx][e]
    return xp.view(B, 16, 4, TY, M, TX, M).permute(1, 4, 6, 0, 3, 5, 2).reshape(16, M * M, B * TY * TX, 4).contiguous()


def from_layout_t(t, M, H, W):
    TY, TX = (H + M - 1) // M, (W + M - 1) // M
    B = t.shape[2] // (TY * TX)
    # [q][i][j][B][ty][tx][e] -> [B][q][e][ty][i][tx][j]
    return t.view(16, M, M, B, TY, TX, 4).permute(3, 0, 6, 4, 1, 5, 2).reshape(B, 64, TY * M, TX * M)[:, :, :H, :W]


@pytest.mark.parametrize("tile", [4, 3])
@pytest.mark.parametrize("H,W,B", [(9, 9, 7), (19, 19, 3), (8, 8, 5), (5, 7, 9), (9, 9, 300), (19, 19, 40), (13, 6, 11)])
def test_wino_conv_general_boards_layout_t(plug, H, W, B, tile):
    """The any-board kernel on layout T (aligned 16-byte vectors of four channels, the product path for Go 9x9 / 19x19): both
    tilings on every board against conv2d in float64; cells of the tiles that lie off the board hold NaN in the inputs (they must
    never be read); board count from the host and from device memory."""
    import torch
    torch.manual_seed(H * 1000 + W * 10 + B + tile)

    def act():
        x = torch.randn(B, 64, H, W, device="cuda")
        t = to_layout_t(x, tile)
        mask = to_layout_t(torch.ones_like(x), tile)
        return x, torch.where(mask > 0, t, torch.full_like(t, float("nan")))

    x, xt = act()
    res, rest = act()
    w = torch.randn(64, 64, 3, 3, device="cuda") * 0.06
    scale = torch.rand(64, device="cuda") + 0.5
    shift = torch.randn(64, device="cuda") * 0.3
    u = np.zeros((36 if tile == 4 else 28) * 64 * 64, np.float32)
    wc = np.ascontiguousarray(w.cpu().numpy())
    plug.sprl_wino_transform_weights_t(wc.ctypes.data, u.ctypes.data, tile)
    ud = torch.from_numpy(u).cuda()
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    for use_res, relu in ((True, True), (False, False)):
        want = ref + res.double() if use_res else ref
        if relu:
            want = torch.relu(want)
        yt = torch.full_like(xt, float("nan"))
        rc = plug.sprl_wino_conv64_t(xt.data_ptr(), ud.data_ptr(), scale.data_ptr(), shift.data_ptr(), rest.data_ptr() if use_res else None,
                                     yt.data_ptr(), B, H, W, int(relu), tile, None, None)
        assert rc == 0
        torch.cuda.synchronize()
        y = from_layout_t(yt, tile, H, W)
        err = (y.double() - want).abs().max().item()
        assert err < 8e-5, (H, W, B, tile, use_res, relu, err)
        live = max(1, B // 2)
        cnt = torch.tensor([live], dtype=torch.int32, device="cuda")
        y2t = torch.full_like(xt, float("nan"))
        rc = plug.sprl_wino_conv64_t(xt.data_ptr(), ud.data_ptr(), scale.data_ptr(), shift.data_ptr(), rest.data_ptr() if use_res else None,
                                     y2t.data_ptr(), B, H, W, int(relu), tile, cnt.data_ptr(), None)
        assert rc == 0
        torch.cuda.synchronize()
        y2 = from_layout_t(y2t, tile, H, W)
        assert torch.equal(y2[:live], y[:live]) and torch.isnan(y2[live:]).all()


@pytest.mark.parametrize("P,H,W,B,tile", [(17, 9, 9, 7, 3), (17, 19, 19, 3, 4), (3, 13, 6, 11, 4), (17, 9, 9, 513, 3), (3, 9, 9, 1, 4)])
def test_stem_any_board_layout_t(plug, P, H, W, B, tile):
    """The MFMA stem writing layout T: same values as conv2d in float64 once the layout is undone."""
    import torch
    torch.manual_seed(P * 100 + H + B)
    x = (torch.rand(B, P, H, W, device="cuda") < 0.4).float() + 0.25 * torch.randn(B, P, H, W, device="cuda")
    w = torch.randn(64, P, 3, 3, device="cuda") * 0.2
    scale = torch.rand(64, device="cuda") + 0.5
    shift = torch.randn(64, device="cuda") * 0.3
    want = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), padding=1) * scale.double().view(1, -1, 1, 1)
                      + shift.double().view(1, -1, 1, 1))
    TY, TX = (H + tile - 1) // tile, (W + tile - 1) // tile
    yt = torch.full((16, tile * tile, B * TY * TX, 4), float("nan"), device="cuda")
    rc = plug.sprl_stem_conv3x3_t(x.data_ptr(), w.data_ptr(), scale.data_ptr(), shift.data_ptr(), yt.data_ptr(), B, P, H, W, tile, None, None)
    assert rc == 0
    torch.cuda.synchronize()
    err = (from_layout_t(yt, tile, H, W).double() - want).abs().max().item()
    assert err < 1e-5, (P, H, W, B, tile, err)


@pytest.mark.parametrize("P,H,W,B", [(17, 9, 9, 7), (17, 19, 19, 3), (3, 13, 6, 11), (17, 9, 9, 513), (3, 9, 9, 1)])
def test_stem_any_board_mfma_and_valu_forms(plug, P, H, W, B):
    """Stem for boards wider than 8 (cnn_epilogue.hip: stem_mfma_nchw_kernel on the matrix cores, and the VALU form,
    sprl_stem_conv3x3_nchw_valu) against conv2d in float64: conv3x3 P -> 64, folded scale/shift, ReLU, NCHW in and out."""
    import torch
    plug.sprl_stem_conv3x3_nchw.argtypes = [C.c_void_p] * 5 + [C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p]
    plug.sprl_stem_conv3x3_nchw_valu.argtypes = plug.sprl_stem_conv3x3_nchw.argtypes
    torch.manual_seed(P * 100 + H + B)
    x = (torch.rand(B, P, H, W, device="cuda") < 0.4).float() + 0.25 * torch.randn(B, P, H, W, device="cuda")
    w = torch.randn(64, P, 3, 3, device="cuda") * 0.2
    scale = torch.rand(64, device="cuda") + 0.5
    shift = torch.randn(64, device="cuda") * 0.3
    want = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), padding=1) * scale.double().view(1, -1, 1, 1)
                      + shift.double().view(1, -1, 1, 1))
    for valu in (False, True):
        fn = plug.sprl_stem_conv3x3_nchw_valu if valu else plug.sprl_stem_conv3x3_nchw      # (an entry point, not an environment switch)
        y = torch.full((B, 64, H, W), float("nan"), device="cuda")
        rc = fn(x.data_ptr(), w.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), B, P, H, W, None)
        assert rc == 0
        torch.cuda.synchronize()
        err = (y.double() - want).abs().max().item()
        assert err < 1e-5, (P, H, W, B, valu, err)


@pytest.mark.parametrize("H,W,B", [(8, 8, 37), (6, 7, 5), (7, 7, 9), (8, 8, 1030)])
def test_trainer_conv_op_forward_and_gradients(plug, H, W, B):
    """sprl_amd/trainer_ops.py: WinoConv3x3 (the trunk convolutions of a TRAINING step on the hand-written kernel) against
    conv2d + autograd in float64: the output, the input gradient (the same kernel with the filters transposed over the channels
    and rotated by 180 degrees, brought to the Winograd domain on the device), the weight and bias gradients (the framework's)."""
    import torch
    from sprl_amd import trainer_ops as TO
    assert TO.available()
    torch.manual_seed(H * 100 + W * 10 + B)
    x = torch.randn(B, 64, H, W, device="cuda", requires_grad=True)
    w = (torch.randn(64, 64, 3, 3, device="cuda") * 0.06).requires_grad_()
    b = (torch.randn(64, device="cuda") * 0.3).requires_grad_()
    gy = torch.randn(B, 64, H, W, device="cuda")
    assert TO.supported(x, w)
    y = TO.WinoConv3x3.apply(x, w, b)
    y.backward(gy)
    xd, wd, bd = (t.detach().double().requires_grad_() for t in (x, w, b))
    yd = torch.nn.functional.conv2d(xd, wd, bd, padding=1)
    yd.backward(gy.double())
    scale = lambda t: float(t.abs().max())
    assert (y.double() - yd).abs().max().item() < 2e-5 * max(1.0, scale(yd))
    assert (x.grad.double() - xd.grad).abs().max().item() < 2e-5 * max(1.0, scale(xd.grad))
    assert (w.grad.double() - wd.grad).abs().max().item() < 1e-4 * max(1.0, scale(wd.grad))
    assert (b.grad.double() - bd.grad).abs().max().item() < 1e-4 * max(1.0, scale(bd.grad))
    # the device-side filter transform is the host's (torch_eval.cpp: wino_transform) up to the last bit of the final rounding
    # (double arithmetic on both sides; the device compiler contracts multiply-adds)
    import ctypes as C
    L = TO._lib()
    L.sprl_wino_transform_weights.argtypes = [C.c_void_p, C.c_void_p]
    uh = torch.zeros(36 * 64 * 64)
    wc = w.detach().cpu().contiguous()
    L.sprl_wino_transform_weights(wc.data_ptr(), uh.data_ptr())
    ud = torch.empty(36 * 64 * 64, device="cuda")
    assert L.sprl_train_wino_weights(w.detach().contiguous().data_ptr(), ud.data_ptr(), 0, None) == 0
    torch.cuda.synchronize()
    assert torch.allclose(ud.cpu(), uh, rtol=1e-6, atol=1e-9)


def test_trainer_fast_trunk_trains_like_the_library(plug):
    """train_network with the trunk convolutions on the hand-written kernel and the step replayed from a HIP graph (the defaults)
    against the plain eager loop on the library's kernels: same data, same seeds - the per-epoch losses agree to 1e-3 relative
    (different fp32 summation orders through 2 x 3 epochs of AdamW), the best epoch is the same."""
    import torch
    from sprl_amd import trainer as T
    from sprl_amd.network import GridResNet
    torch.manual_seed(5)
    n, bs = 8 * 1024, 1024
    s = (torch.rand(n, 3, 8, 8, device="cuda") > 0.5).float()
    d = torch.softmax(torch.randn(n, 65, device="cuda"), 1)
    o = torch.sign(torch.randn(n, 1, device="cuda"))
    t = torch.ones(n, 1, device="cuda")
    hist = {}
    for name, kw in (("fast", dict(fast_conv=True, use_graph=True)), ("plain", dict(fast_conv=False, use_graph=False))):
        torch.manual_seed(11)
        net = GridResNet(8, 8, 65, 1, 2, 64)
        cfg = T.TrainerConfig(batch_size=bs, max_groups=1, epochs_per_group=3, **kw)
        best, h = T.train_network(net, 0.01, (s, d, o, t), cfg, generator=torch.Generator(device="cuda").manual_seed(3))
        hist[name] = h
        assert "forward" not in net.residual_blocks[0].conv1.__dict__          # the patch is gone when train_network returns
    for a, b in zip(hist["fast"]["epochs"], hist["plain"]["epochs"]):
        for k in ("train_policy", "train_value", "val_policy", "val_value"):
            assert abs(a[k] - b[k]) < 2e-3 * max(1.0, abs(b[k])), (k, a, b)
    assert hist["fast"]["best_epoch"] == hist["plain"]["best_epoch"]
