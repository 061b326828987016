"""One epoch of the trainer at the BASELINE shape (2 x 64 network, batch 1024, window resident in HBM) on synthetic samples:
wall time per optimiser step, and - run under `rocprofv3 --kernel-trace --stats` - the kernels a step is made of.
    python tools/trainer_profile.py [--samples 262144] [--graph]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=262144)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--graph", action="store_true", help="replay the optimiser step from a captured HIP graph (the TrainerConfig default)")
    ap.add_argument("--eager", action="store_true", help="eager steps (what a kernel trace should see)")
    ap.add_argument("--no-fast-conv", action="store_true", help="the trunk convolutions on the library's kernels (TrainerConfig.fast_conv off)")
    ap.add_argument("--no-gemm", action="store_true", help="MIOPEN_DEBUG_CONV_GEMM=0: the library's im2col + GEMM convolutions out of its choice")
    ap.add_argument("--benchmark", action="store_true", help="torch.backends.cudnn.benchmark = True: the library measures its solvers per shape")
    a = ap.parse_args()
    if a.no_gemm:
        os.environ["MIOPEN_DEBUG_CONV_GEMM"] = "0"
    torch.backends.cudnn.benchmark = bool(a.benchmark)
    from sprl_amd import trainer as T
    from sprl_amd.network import GridResNet
    torch.manual_seed(0)
    n, bs = a.samples, 1024
    s = (torch.rand(n, 3, 8, 8, device="cuda") > 0.5).float()
    d = torch.softmax(torch.randn(n, 65, device="cuda"), 1)
    o = torch.sign(torch.randn(n, 1, device="cuda"))
    t = torch.ones(n, 1, device="cuda")
    net = GridResNet(8, 8, 65, 1, 2, 64)
    cfg = T.TrainerConfig(batch_size=bs, max_groups=1, epochs_per_group=1, use_graph=a.graph and not a.eager, fast_conv=not a.no_fast_conv)
    T.train_network(net, 0.01, (s[:8 * bs], d[:8 * bs], o[:8 * bs], t[:8 * bs]), cfg)      # warm-up (kernel selection)
    torch.cuda.synchronize()
    cfg = T.TrainerConfig(batch_size=bs, max_groups=1, epochs_per_group=a.epochs, use_graph=a.graph and not a.eager, fast_conv=not a.no_fast_conv)
    t0 = time.time()
    best, hist = T.train_network(net, 0.01, (s, d, o, t), cfg)
    torch.cuda.synchronize()
    dt = time.time() - t0
    steps = a.epochs * ((int(0.9 * n) + bs - 1) // bs)
    print(f"trainer, BASELINE shape, graph={a.graph and not a.eager}, fast_conv={not a.no_fast_conv}, no_gemm={a.no_gemm}, benchmark={a.benchmark}: {steps} optimiser steps of batch {bs} + {a.epochs} validation passes in {dt:.2f} s = "
          f"{1e3 * dt / steps:.3f} ms/step, {steps * bs / dt:.0f} samples/s; losses {hist['epochs'][-1]['train_policy']:.4f} / "
          f"{hist['epochs'][-1]['train_value']:.4f}")


if __name__ == "__main__":
    main()
