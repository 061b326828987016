"""A/B of the 8x8 trunk convolution's launch forms on the GPU box (diagnostic): one workgroup per board group (the round-3 form)
against the multi-group form (PERSIST: at most `persist_wgs` workgroups, the next group's prologue hidden in the current group's
last phases), each checked bit for bit against the one-workgroup-per-group output and against conv2d in float64.
    python tools/wino8_ab.py [--batch 13492 6746] [--grids 768 1024 1536]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def _w_index():
    """flat layout-W index of (channel k, row, col): x[n][g][i][cs][tile][j], k = 16 (g >> 2) + 4 cs + (g & 3), cell (4 ty + i, 4 tx + j)"""
    k = torch.arange(64).view(64, 1, 1)
    r = torch.arange(8).view(1, 8, 1)
    c = torch.arange(8).view(1, 1, 8)
    g, cs, tile = 4 * (k >> 4) + (k & 3), (k >> 2) & 3, 2 * (r >> 2) + (c >> 2)
    return (g * 256 + (r & 3) * 64 + cs * 16 + tile * 4 + (c & 3)).reshape(-1)


def to_layout_w(x):
    """[B][64][8][8] -> layout W [B][4096]"""
    out = torch.empty(x.shape[0], 4096, dtype=x.dtype, device=x.device)
    out[:, _w_index().to(x.device)] = x.reshape(x.shape[0], 4096)
    return out


def from_layout_w(y):
    return y[:, _w_index().to(y.device)].reshape(y.shape[0], 64, 8, 8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[13492, 6746])
    ap.add_argument("--grids", type=int, nargs="+", default=[-1, 1024, 1536, 2048, 2560])
    ap.add_argument("--iters", type=int, default=40)
    a = ap.parse_args()
    L = C.CDLL(os.path.join(ROOT, "sprl_amd", "libsprl_amd_torch.so"))
    L.sprl_wino_conv64_persist.argtypes = [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.sprl_wino_transform_weights.argtypes = [C.c_void_p, C.c_void_p]
    torch.manual_seed(2)
    w = (torch.randn(64, 64, 3, 3) * 0.06).contiguous()
    u = torch.zeros(36 * 64 * 64)
    L.sprl_wino_transform_weights(w.data_ptr(), u.data_ptr())
    u = u.cuda()
    sc, sh = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.3
    # warm the clocks before the first measurement
    wx = torch.randn(4096, 4096, device="cuda")
    for _ in range(50):
        wx = torch.tanh(wx @ wx * 1e-3)
    torch.cuda.synchronize()
    for B in a.batch:
        xs = torch.randn(B, 64, 8, 8, device="cuda")
        rs = torch.randn(B, 64, 8, 8, device="cuda")
        x, res = to_layout_w(xs), to_layout_w(rs)
        y = torch.empty_like(x)
        nchk = min(B, 64)
        want = torch.relu(torch.nn.functional.conv2d(xs[-nchk:].double(), w.double().cuda(), padding=1) * sc.double().view(1, -1, 1, 1)
                          + sh.double().view(1, -1, 1, 1) + rs[-nchk:].double())

        def run(grid, npre, with_res):
            return L.sprl_wino_conv64_persist(x.data_ptr(), u.data_ptr(), sc.data_ptr(), sh.data_ptr(), res.data_ptr() if with_res else None,
                                              y.data_ptr(), B, 8, 8, 1, None, grid, npre, None)

        def timeit(grid, npre, with_res):
            for _ in range(3):
                run(grid, npre, with_res)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run(grid, npre, with_res)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / a.iters * 1e3

        flop = 2.0 * B * 4 * 36 * 64 * 64
        print(f"8x8 trunk convolution, {B} boards = {(B + 3) // 4} groups")
        for with_res in (True, False):
            y.fill_(float("nan"))
            assert run(0, 2, with_res) == 0
            torch.cuda.synchronize()
            base = y.clone()
            if with_res:
                err = (from_layout_w(base[-nchk:]).double() - want).abs().max().item()
                print(f"  one workgroup per group vs conv2d float64 (last {nchk} boards): {err:.3e}")
                assert err < 1e-4
            t0 = timeit(0, 2, with_res)
            print(f"  res={int(with_res)} one workgroup per group                  {t0:8.1f} us   {flop / t0 / 1e6 / 157.3:.3f}")
            for npre in (2,):
                for grid in a.grids:
                    y.fill_(float("nan"))
                    assert run(grid, npre, with_res) == 0
                    torch.cuda.synchronize()
                    same = bool((y.view(torch.int32) == base.view(torch.int32)).all())
                    t = timeit(grid, npre, with_res)
                    print(f"  res={int(with_res)} multi-group, {grid:5d} workgroups, npre {npre}      {t:8.1f} us   {flop / t / 1e6 / 157.3:.3f}   "
                          f"({100 * (t0 / t - 1):+5.1f} %)   bit-identical: {same}")
                    assert same, "the multi-group form must not change a bit"
            t1 = timeit(0, 2, with_res)
            print(f"  res={int(with_res)} one workgroup per group (again)          {t1:8.1f} us")


if __name__ == "__main__":
    main()
