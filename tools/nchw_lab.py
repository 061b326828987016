"""Where the any-board trunk convolution (cnn_wino.hip: wino_conv64_nchw_kernel) spends its time: the kernel is built with
-DSPRL_WINO_LAB into tools/libwino_lab.so and timed with stages switched off one at a time (results are wrong then; the
unmasked run is checked against conv2d).  Diagnostic, never shipped.
    python tools/nchw_lab.py [--game go9|go19] [--batch N]"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tools", "libwino_lab.so")


def build():
    src = os.path.join(ROOT, "sprl_amd", "csrc", "cnn_wino.hip")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize",
                               "-DSPRL_WINO_LAB", "-o", LIB, src])


def wino_f(w, tile, quad_order=False):
    """U = G g G^T in float64 -> the kernel's layout U4[p / 4][s][kb][lane][p % 4] (torch_eval.cpp: wino_transform)."""
    if tile == 4:
        G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]])
    else:
        G = np.array([[1 / 2, 0, 0], [1 / 2, 1 / 2, 1 / 2], [1 / 6, -1 / 6, 1 / 6], [1 / 6, 1 / 3, 2 / 3], [0, 0, 1]])
    n = G.shape[0]
    U = np.einsum("ai,kcij,bj->kcab", G, w.astype(np.float64), G).reshape(64, 64, n * n)      # [k][c][p]
    nq = (n * n + 3) // 4
    out = np.zeros((nq, 16, 4, 64, 4), np.float32)
    k, c, p = np.meshgrid(np.arange(64), np.arange(64), np.arange(n * n), indexing="ij")
    s = (c >> 2) if quad_order else 4 * (c >> 4) + (c & 3)              # layout T: a K step is one channel quad
    lane = ((c & 3) if quad_order else ((c >> 2) & 3)) * 16 + (k & 15)
    out[p >> 2, s, k >> 4, lane, p & 3] = U[k, c, p]
    return out.reshape(-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--game", default="go9")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--layout", default="t", choices=["t", "nchw"], help="t = layout T (the product path), nchw = the round-2 form")
    ap.add_argument("--lib", default=None, help="another lab build of cnn_wino.hip (e.g. -DSPRL_WINO_DEEP4=1)")
    a = ap.parse_args()
    if not a.lib:
        build()
    L = C.CDLL(a.lib or LIB)
    L.sprl_wino_conv64_nchw_tiled.argtypes = [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]
    L.sprl_wino_conv64_t.argtypes = [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]
    H = W = 9 if a.game == "go9" else 19
    B = a.batch or (8192 if a.game == "go9" else 2048)
    tile = a.tile or L.sprl_wino_nchw_tile(H, W)
    torch.manual_seed(1)
    n = B * 64 * H * W

    def act():
        flat = torch.zeros(n + 12, device="cuda")
        flat[4:n + 4] = torch.randn(n, device="cuda")
        return flat[4:n + 4].view(B, 64, H, W)

    x, res, y = act(), act(), act()
    w = torch.randn(64, 64, 3, 3) * 0.06
    lay_t = a.layout == "t"
    u = torch.from_numpy(wino_f(w.numpy(), tile, lay_t)).cuda()
    sc, sh = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.3
    if lay_t:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_gpu_cnn import from_layout_t, to_layout_t
        xt, rest = to_layout_t(x, tile), to_layout_t(res, tile)
        yt = torch.zeros_like(xt)

    def run(with_res=True):
        if lay_t:
            return L.sprl_wino_conv64_t(xt.data_ptr(), u.data_ptr(), sc.data_ptr(), sh.data_ptr(), rest.data_ptr() if with_res else None,
                                        yt.data_ptr(), B, H, W, 1, tile, None, None)
        return L.sprl_wino_conv64_nchw_tiled(x.data_ptr(), u.data_ptr(), sc.data_ptr(), sh.data_ptr(), res.data_ptr() if with_res else None,
                                             y.data_ptr(), B, H, W, 1, tile, None, None)

    assert L.sprl_wino_lab_set_dbg(0) == 0 and run() == 0
    torch.cuda.synchronize()
    nb = min(B, 64)
    if lay_t:
        y = from_layout_t(yt, tile, H, W)
    want = torch.relu(torch.nn.functional.conv2d(x[:nb].double(), w.cuda().double(), padding=1) * sc.double().view(1, -1, 1, 1)
                      + sh.double().view(1, -1, 1, 1) + res[:nb].double())
    print(f"{a.game}: {B} boards, F({tile}x{tile},3x3), {'layout T' if lay_t else 'NCHW'}; max |err| vs float64 conv2d on {nb} boards: {(y[:nb].double() - want).abs().max().item():.2e}")
    tiles = ((H + tile - 1) // tile) * ((W + tile - 1) // tile)
    flop = 2.0 * B * tiles * (tile + 2) ** 2 * 64 * 64

    def timeit(with_res=True, iters=20):
        for _ in range(3):
            run(with_res)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run(with_res)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    base = None
    for name, mask, with_res in (("everything on", 0, True), ("no residual (RES = 0 variant)", 0, False), ("no activation loads", 1, True),
                                 ("no loads, no LDS patch stores", 3, True), ("no input transform", 4, True),
                                 ("no MFMA loop (first K step only)", 8, True), ("no output stores", 16, True),
                                 ("only MFMA loop + transform (no loads/LDS stores/output stores)", 1 + 2 + 16, True),
                                 ("only loads + LDS stores + output (no transform, no MFMA)", 4 + 8, True)):
        L.sprl_wino_lab_set_dbg(mask)
        us = timeit(with_res)
        base = base or us
        print(f"  {name:62s} {us:8.1f} us  ({us - base:+7.1f})   {flop / us / 1e6 / 157.3:.3f} of the fp32 matrix peak if it were the full kernel")
    L.sprl_wino_lab_set_dbg(0)


if __name__ == "__main__":
    main()
