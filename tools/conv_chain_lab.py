"""Would the 2-block forward run faster in SLICES of the batch?  (diagnostic)  tools/wino8_lab.py shows the 8x8 trunk convolution
17 % faster when its activation loads are answered by the memory-side cache instead of HBM.  In the product a convolution reads
what the previous one wrote ~110 MB of traffic earlier (two populations: twice that), too far back for the 256 MB cache; run in
slices of S boards - all four convolutions on one slice, then the next slice - the distance shrinks to ~3 x S x 16 KB.
This lab replays the forward's four trunk convolutions (plain -> residual -> plain -> residual over four buffers, as
torch_eval.cpp: forward_wino chains them) for TWO populations on two streams, captured into one HIP graph so that the host is
not in the way, once with whole-batch launches and once sliced.
    python tools/conv_chain_lab.py [--boards 6748] [--slices 0 1024 2048 3374] [--rounds 6] [--lib base]"""
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
    ap.add_argument("--boards", type=int, default=6748)
    ap.add_argument("--slices", type=int, nargs="+", default=[0, 1024, 2048, 3374])
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--lib", default="base", help="tools/libwino_lab_<lib>.so (python tools/conv_ab.py --build base=)")
    ap.add_argument("--populations", type=int, default=2)
    a = ap.parse_args()
    L = C.CDLL(os.path.join(ROOT, "tools", f"libwino_lab_{a.lib}.so"))
    L.sprl_wino_conv64.argtypes = [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p]
    B, P = a.boards, a.populations
    torch.manual_seed(3)
    us = [torch.from_numpy(nchw_lab.wino_f((torch.randn(64, 64, 3, 3) * 0.06).numpy(), 4)).cuda() for _ in range(4)]
    sc = [torch.rand(64, device="cuda") * 0.2 + 0.9 for _ in range(4)]
    sh = [torch.randn(64, device="cuda") * 0.05 for _ in range(4)]
    # per population: x0 (stem output), y1, y2, y3, y4 - every layer writes a buffer of its own, as the plugin does
    bufs = [[torch.randn(B, 4096, device="cuda") * (1.0 if i == 0 else 0.0) for i in range(5)] for _ in range(P)]
    wx = torch.randn(4096, 4096, device="cuda")
    for _ in range(50):
        wx = torch.tanh(wx @ wx * 1e-3)
    torch.cuda.synchronize()

    def forward(pop, stream, slice_boards):
        b = bufs[pop]
        S = slice_boards or B
        for s0 in range(0, B, S):
            n = min(S, B - s0)
            off = s0 * 4096 * 4
            ptr = [t.data_ptr() + off for t in b]
            for layer, (src, res, dst) in enumerate(((0, None, 1), (1, 0, 2), (2, None, 3), (3, 2, 4))):
                rc = L.sprl_wino_conv64(ptr[src], us[layer].data_ptr(), sc[layer].data_ptr(), sh[layer].data_ptr(),
                                        ptr[res] if res is not None else None, ptr[dst], n, 8, 8, 1, stream.cuda_stream)
                assert rc == 0

    ref = None
    flop = 2.0 * B * 4 * 36 * 64 * 64 * 4 * P * a.rounds
    print(f"four trunk convolutions per forward, {P} populations x {B} boards on {P} streams, {a.rounds} forwards each per graph, lib {a.lib}")
    for S in a.slices:
        for t in bufs:
            for y in t[1:]:
                y.zero_()
        main_s = torch.cuda.Stream()
        streams = [torch.cuda.Stream() for _ in range(P)]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main_s):
            for s in streams:
                s.wait_stream(main_s)
            for _ in range(a.rounds):
                for p in range(P):
                    forward(p, streams[p], S)
            for s in streams:
                main_s.wait_stream(s)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3)
        out = bufs[0][4].clone()
        same = True if ref is None else bool((out.view(torch.int32) == ref.view(torch.int32)).all())
        ref = out if ref is None else ref
        launches = 4 * ((B + (S or B) - 1) // (S or B))
        print(f"  slices of {S or B:5d} boards ({launches:3d} launches per forward): {best / a.rounds:9.1f} us per round of {P} forwards   "
              f"{flop / best / 1e6 / 157.3:.3f} of the fp32 matrix peak   outputs identical to the first form: {same}", flush=True)


if __name__ == "__main__":
    main()
