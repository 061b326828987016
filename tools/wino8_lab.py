"""Stage costs of the 8x8 trunk convolution (cnn_wino.hip: wino_conv64_kernel): the kernel built with -DSPRL_WINO_LAB
(tools/libwino_lab.so) and timed with stages switched off / thinned one at a time (results are wrong then).  Diagnostic.
    python tools/wino8_lab.py [--batch 13492]"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import nchw_lab  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=13492)
    ap.add_argument("--lib", default=None, help="another lab build (e.g. -DSPRL_WINO_LAB_BREUSE: every B operand feeds two MFMAs)")
    a = ap.parse_args()
    if not a.lib:
        nchw_lab.build()
    L = C.CDLL(a.lib or nchw_lab.LIB)
    L.sprl_wino_conv64.argtypes = [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p]
    B = a.batch
    torch.manual_seed(2)
    x = torch.randn(B, 4096, device="cuda")
    res = torch.randn(B, 4096, device="cuda")
    y = torch.empty_like(x)
    w = torch.randn(64, 64, 3, 3) * 0.06
    u = torch.from_numpy(nchw_lab.wino_f(w.numpy(), 4)).cuda()
    sc, sh = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.3

    def run(with_res=True):
        return L.sprl_wino_conv64(x.data_ptr(), u.data_ptr(), sc.data_ptr(), sh.data_ptr(), res.data_ptr() if with_res else None,
                                  y.data_ptr(), B, 8, 8, 1, None)

    def timeit(with_res=True, iters=30):
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

    flop = 2.0 * B * 4 * 36 * 64 * 64
    base = None
    print(f"8x8 trunk convolution, {B} boards, {os.path.basename(a.lib or nchw_lab.LIB)}")
    for name, mask, with_res in (("everything on", 0, True), ("no residual (RES = 0 variant)", 0, False),
                                 ("no input transform", 4, True), ("filter quads loaded once", 64, True),
                                 ("no output stage", 128, True),
                                 ("no activation loads behind the prologue's", 1, True),
                                 ("activation loads answered by L2 (every workgroup reads boards 0-3)", 256, True),
                                 ("the same, without the residual", 256, False),
                                 ("activation loads from a 64 MB window (memory-side cache, not L2)", 512, True),
                                 ("the same, without the residual", 512, False),
                                 ("filters once + activation loads from L2", 64 + 256, True),
                                 ("no transform + filters once + no output stage (MFMA loop + activation loads)", 4 + 64 + 128, True)):
        L.sprl_wino_lab_set_dbg(mask)
        us = timeit(with_res)
        base = base or us
        print(f"  {name:92s} {us:8.1f} us  ({us - base:+7.1f})   {flop / us / 1e6 / 157.3:.3f}")
    L.sprl_wino_lab_set_dbg(0)


if __name__ == "__main__":
    main()
