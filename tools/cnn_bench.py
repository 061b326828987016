"""Micro-benchmark of the CNN evaluator on the GPU: the hand-written Winograd trunk convolution against
conv2d (MIOpen) + the epilogue kernel, and the whole network forward through the plugin with either path."""
import argparse
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sprl_amd import engine as E  # noqa: E402
from sprl_amd.network import make_network, trace_to_file  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--forward-only", action="store_true")
    a = ap.parse_args()
    E.load_library()
    plug = C.CDLL(os.path.join(os.path.dirname(E.DEFAULT_LIB), "libsprl_amd_torch.so"))
    plug.sprl_wino_conv64.argtypes = [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p]
    plug.sprl_wino_transform_weights.argtypes = [C.c_void_p, C.c_void_p]
    plug.sprl_torch_load.restype = C.c_void_p
    plug.sprl_torch_load.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    plug.sprl_torch_forward.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p, C.c_char_p, C.c_int]
    B = a.batch
    if not a.forward_only:
        x = torch.randn(B, 64, 8, 8, device="cuda")
        w = torch.randn(64, 64, 3, 3, device="cuda") * 0.06
        sc, sh = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda")
        u = np.zeros(36 * 64 * 64, np.float32)
        wc = np.ascontiguousarray(w.cpu().numpy())
        plug.sprl_wino_transform_weights(wc.ctypes.data, u.ctypes.data)
        ud = torch.from_numpy(u).cuda()
        y = torch.empty_like(x)
        t_w = timeit(lambda: plug.sprl_wino_conv64(x.data_ptr(), ud.data_ptr(), sc.data_ptr(), sh.data_ptr(), x.data_ptr(),
                                                   y.data_ptr(), B, 8, 8, 1, None))
        t_c = timeit(lambda: torch.relu_(torch.nn.functional.conv2d(x, w, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + x))
        t_c0 = timeit(lambda: torch.nn.functional.conv2d(x, w, padding=1))
        fl = 2.0 * B * 64 * 64 * 64 * 9
        print(f"batch {B}: winograd+epilogue kernel {t_w * 1e3:.1f} us ({fl / t_w / 1e9:.1f} direct-equivalent TFLOP/s, "
              f"{fl / 4 / t_w / 1e9:.1f} TFLOP/s of MFMA work); conv2d alone {t_c0 * 1e3:.1f} us; conv2d + torch epilogue {t_c * 1e3:.1f} us")
    with tempfile.TemporaryDirectory() as td:
        path = trace_to_file(make_network("othello", 2, 64, seed=0), os.path.join(td, "m.pt"), "othello")
        xin = (torch.rand(B, 3, 8, 8, device="cuda") > 0.6).float()
        lo, va = torch.zeros(B, 65, device="cuda"), torch.zeros(B, device="cuda")
        err = C.create_string_buffer(512)
        h = plug.sprl_torch_load(path.encode(), 0, err, 512)
        t = timeit(lambda: plug.sprl_torch_forward(h, xin.data_ptr(), B, 3, 8, 8, lo.data_ptr(), 65, va.data_ptr(), err, 512), 10, 3)
        print(f"whole network forward (2 blocks x 64), batch {B}: {t:.3f} ms  [SPRL_TORCH_NO_WINOGRAD={os.environ.get('SPRL_TORCH_NO_WINOGRAD', '')}]")


if __name__ == "__main__":
    main()
