"""A/B of builds of the trunk-convolution kernels (cnn_wino.hip) that differ in compile-time switches: every build is compiled
into tools/libwino_lab_<tag>.so (here, on the CPU: hipcc cross-compiles), then on the GPU box each shape is run through all of
them on the same inputs - outputs compared bit for bit with the first build (and with conv2d in float64), launches timed
alternately.  Diagnostic, never shipped.
    python tools/conv_ab.py --build base= broll4=-DSPRL_WINO_BROLL=4            # in the container
    python tools/conv_ab.py --run base broll4 [--shapes 8x8 go9 go19] [--boards 13492 8192 2048]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nchw_lab  # noqa: E402


def lib_path(tag):
    return os.path.join(ROOT, "tools", f"libwino_lab_{tag}.so")


def build(specs):
    src = os.path.join(ROOT, "sprl_amd", "csrc", "cnn_wino.hip")
    for spec in specs:
        tag, _, flags = spec.partition("=")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize", "-o", lib_path(tag), src]
        cmd += [f for f in flags.split(",") if f]
        print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)


def shape_8x8(libs, B, iters, wgroup=()):
    """wgroup: tags of builds with -DSPRL_WINO_WGROUP=1 (layout W' = [n / 4][g][n % 4][256]; B a multiple of 4)"""
    from tools.experiments.wino8_ab import from_layout_w, to_layout_w
    torch.manual_seed(2)
    xs, rs = torch.randn(B, 64, 8, 8, device="cuda"), torch.randn(B, 64, 8, 8, device="cuda")
    x, res = to_layout_w(xs), to_layout_w(rs)
    w = torch.randn(64, 64, 3, 3) * 0.06
    u = torch.from_numpy(nchw_lab.wino_f(w.numpy(), 4)).cuda()
    sc, sh = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.3
    ys = {t: torch.empty_like(x) for t in libs}

    def regroup(t):                                       # [B][16 g][256] -> [B / 4][16 g][4][256]
        return t.view(B // 4, 4, 16, 256).permute(0, 2, 1, 3).contiguous().view(B, 4096)

    def ungroup(t):
        return t.view(B // 4, 16, 4, 256).permute(0, 2, 1, 3).contiguous().view(B, 4096)

    xg, resg = (regroup(x), regroup(res)) if wgroup else (None, None)

    def run(tag, with_res):
        L = libs[tag]
        xi, ri = (xg, resg) if tag in wgroup else (x, res)
        return L.sprl_wino_conv64(xi.data_ptr(), u.data_ptr(), sc.data_ptr(), sh.data_ptr(), ri.data_ptr() if with_res else None,
                                  ys[tag].data_ptr(), B, 8, 8, 1, None)

    run.post = {t: ungroup for t in wgroup}                # outputs of these builds come back in layout W'
    nb = min(B, 64)
    want = torch.relu(torch.nn.functional.conv2d(xs[-nb:].double(), w.double().cuda(), padding=1) * sc.double().view(1, -1, 1, 1)
                      + sh.double().view(1, -1, 1, 1) + rs[-nb:].double())
    flop = 2.0 * B * 4 * 36 * 64 * 64
    return run, ys, (lambda y: (from_layout_w(y[-nb:]).double() - want).abs().max().item()), flop, f"8x8, {B} boards (layout W)"


def shape_go(game, libs, B, iters):
    from test_gpu_cnn import from_layout_t, to_layout_t
    H = W = 9 if game == "go9" else 19
    tile = 3 if game == "go9" else 4
    torch.manual_seed(1)
    x, res = torch.randn(B, 64, H, W, device="cuda"), torch.randn(B, 64, H, W, device="cuda")
    w = torch.randn(64, 64, 3, 3) * 0.06
    u = torch.from_numpy(nchw_lab.wino_f(w.numpy(), tile, True)).cuda()
    sc, sh = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.3
    xt, rest = to_layout_t(x, tile), to_layout_t(res, tile)
    ys = {t: torch.zeros_like(xt) for t in libs}

    def run(tag, with_res):
        L = libs[tag]
        return L.sprl_wino_conv64_t(xt.data_ptr(), u.data_ptr(), sc.data_ptr(), sh.data_ptr(), rest.data_ptr() if with_res else None,
                                    ys[tag].data_ptr(), B, H, W, 1, tile, None, None)

    nb = min(B, 64)
    want = torch.relu(torch.nn.functional.conv2d(x[:nb].double(), w.cuda().double(), padding=1) * sc.double().view(1, -1, 1, 1)
                      + sh.double().view(1, -1, 1, 1) + res[:nb].double())
    tiles = ((H + tile - 1) // tile) ** 2
    flop = 2.0 * B * tiles * (tile + 2) ** 2 * 64 * 64
    return run, ys, (lambda y: (from_layout_t(y, tile, H, W)[:nb].double() - want).abs().max().item()), flop, \
        f"{game}, {B} boards, F({tile}x{tile},3x3), layout T"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", nargs="+", default=None, help="tag=flag,flag ... (flags comma-separated)")
    ap.add_argument("--run", nargs="+", default=None, help="tags; the first one is the reference for the bit comparison")
    ap.add_argument("--shapes", nargs="+", default=["8x8", "go9", "go19"])
    ap.add_argument("--boards", type=int, nargs="+", default=None)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--wgroup", nargs="*", default=[], help="tags built with -DSPRL_WINO_WGROUP=1 (8x8 only; fed the group-major layout)")
    ap.add_argument("--rounds", type=int, default=4)
    a = ap.parse_args()
    if a.build:
        build(a.build)
    if not a.run:
        return
    libs = {}
    for t in a.run:
        L = C.CDLL(lib_path(t))
        L.sprl_wino_conv64.argtypes = [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p]
        L.sprl_wino_conv64_t.argtypes = [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_void_p, C.c_void_p]
        libs[t] = L
    wx = torch.randn(4096, 4096, device="cuda")         # warm the clocks before the first measurement
    for _ in range(50):
        wx = torch.tanh(wx @ wx * 1e-3)
    torch.cuda.synchronize()
    default_boards = {"8x8": 13492, "go9": 8192, "go19": 2048}
    for i, shape in enumerate(a.shapes):
        B = a.boards[i] if a.boards and i < len(a.boards) else default_boards[shape]
        if shape == "8x8":
            B -= B % 4 if a.wgroup else 0
            run, ys, err_of, flop, title = shape_8x8(libs, B, a.iters, tuple(a.wgroup))
        else:
            run, ys, err_of, flop, title = shape_go(shape, libs, B, a.iters)
        print(title, flush=True)
        for with_res in (True, False):
            for t in libs:
                ys[t].fill_(float("nan")) if shape == "8x8" else ys[t].zero_()
                assert run(t, with_res) == 0
            torch.cuda.synchronize()
            ref = ys[a.run[0]]
            post = getattr(run, "post", {})
            same = {t: bool(((post[t](ys[t]) if t in post else ys[t]).view(torch.int32) == ref.view(torch.int32)).all()) for t in libs}
            err = err_of(ref) if with_res else float("nan")
            best = {t: 1e30 for t in libs}
            order = list(libs)
            for rnd in range(a.rounds):                  # alternate the builds, another one first in every round: the build that is
                for t in order[rnd % len(order):] + order[:rnd % len(order)]:      # timed first after a pause comes out 1-4 % slow
                    for _ in range(10):
                        run(t, with_res)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.iters):
                        run(t, with_res)
                    e1.record()
                    torch.cuda.synchronize()
                    best[t] = min(best[t], e0.elapsed_time(e1) / a.iters * 1e3)
            t0 = best[a.run[0]]
            print(f"  res={int(with_res)}" + (f"  ({a.run[0]} vs conv2d float64: {err:.2e})" if with_res else ""))
            for t in libs:
                print(f"    {t:14s} {best[t]:8.1f} us   {flop / best[t] / 1e6 / 157.3:.3f} of the fp32 matrix peak   ({100 * (t0 / best[t] - 1):+5.1f} %)"
                      f"   bit-identical to {a.run[0]}: {same[t]}", flush=True)


if __name__ == "__main__":
    main()
