"""Timeline of the 8x8 trunk convolution's waves (diagnostic): a build of cnn_wino.hip with -DSPRL_WINO_TRACE (the product kernel plus the stamps) writes the shader clock
at the marked points of the kernel (LAB_STAMP) for a window of workgroups; this prints where a wave's time goes - per segment of a
phase: waiting for the activation chunk, K step, input transform, barrier, K step - and how the two workgroups that share a CU
sit against each other.
    python tools/wino8_trace.py [--batch 13492] [--first 1024] [--count 1024] [--no-res]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import nchw_lab  # noqa: E402

SEG = ["wait+store chunk", "K step 2c+1", "transform V", "barrier", "K step 2c+2"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=13492)
    ap.add_argument("--first", type=int, default=1024)
    ap.add_argument("--count", type=int, default=512)
    ap.add_argument("--no-res", action="store_true")
    a = ap.parse_args()
    lib = os.path.join(ROOT, "tools", "libwino_lab_trace.so")      # built in the container: python tools/conv_ab.py --build trace=-DSPRL_WINO_TRACE
    L = C.CDLL(lib)
    L.sprl_wino_conv64.argtypes = [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p]
    L.sprl_wino_lab_set_trace.argtypes = [C.c_void_p, C.c_int, C.c_int]
    B = a.batch
    torch.manual_seed(2)
    x, res = torch.randn(B, 4096, device="cuda"), torch.randn(B, 4096, device="cuda")
    y = torch.empty_like(x)
    w = torch.randn(64, 64, 3, 3) * 0.06
    u = torch.from_numpy(nchw_lab.wino_f(w.numpy(), 4)).cuda()
    sc, sh = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.3

    def run():
        return L.sprl_wino_conv64(x.data_ptr(), u.data_ptr(), sc.data_ptr(), sh.data_ptr(), None if a.no_res else res.data_ptr(),
                                  y.data_ptr(), B, 8, 8, 1, None)

    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us_plain = e0.elapsed_time(e1) / 20 * 1e3
    groups = (B + 3) // 4
    first, count = min(a.first, max(groups - 1, 0)), min(a.count, groups - min(a.first, max(groups - 1, 0)))
    trace = torch.zeros(count * 4 * 128, dtype=torch.int64, device="cuda")
    assert L.sprl_wino_lab_set_trace(trace.data_ptr(), first, count) == 0
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    us_traced = e0.elapsed_time(e1) * 1e3
    L.sprl_wino_lab_set_trace(None, 0, 0)
    t = trace.cpu().numpy().reshape(count, 4, 128).astype(np.int64)
    t0 = t[:, :, 0].min()
    span = t[:, :, 52].max() - t0
    print(f"8x8 trunk convolution, {B} boards = {groups} workgroups, res={int(not a.no_res)}: {us_plain:.1f} us per launch "
          f"({us_traced:.1f} us with the trace on); workgroups {first}..{first + count - 1} traced, their stamps span {span} ticks")
    life = (t[:, :, 52] - t[:, :, 0])                      # [wg][wave]
    ticks_per_us = None
    # calibrate the clock: a launch of `groups` workgroups in 512 slots takes us_traced; lifetime in ticks x generations ~ launch time
    gens = groups / 512.0
    ticks_per_us = life.mean() * gens / us_traced
    print(f"  wave lifetime {life.mean():.0f} ticks (min {life.min()}, max {life.max()}); {gens:.2f} generations of 512 workgroups -> "
          f"about {ticks_per_us:.0f} ticks per us if the slots never idle")
    pro = np.stack([t[:, :, 1] - t[:, :, 0], t[:, :, 2] - t[:, :, 1], t[:, :, 3] - t[:, :, 2]], -1).reshape(-1, 3).mean(0)
    print(f"  prologue: start -> first chunks stored {pro[0]:.0f}, -> first V built {pro[1]:.0f}, K step 0 {pro[2]:.0f} ticks")
    seg = np.zeros((count, 4, 8, 5), np.int64)
    for c in range(8):
        for k in range(5):
            seg[:, :, c, k] = t[:, :, 5 + 6 * c + k] - t[:, :, 4 + 6 * c + k]
    print("  phase segments, mean ticks per wave (phases 0..7):")
    for k in range(5):
        print(f"    {SEG[k]:18s} " + " ".join(f"{seg[:, :, c, k].mean():7.0f}" for c in range(8)) + f"   | mean of phases 1-6: {seg[:, :, 1:7, k].mean():7.0f}")
    inner = np.zeros((count, 4, 8, 3), np.int64)          # K step 2c+1: request issue, first five quads, last four
    for c in range(8):
        inner[:, :, c, 0] = t[:, :, 64 + 2 * c] - t[:, :, 5 + 6 * c]
        inner[:, :, c, 1] = t[:, :, 65 + 2 * c] - t[:, :, 64 + 2 * c]
        inner[:, :, c, 2] = t[:, :, 6 + 6 * c] - t[:, :, 65 + 2 * c]
    for k, name in enumerate(("  of it: request", "  quads 0-4", "  quads 5-8")):
        print(f"    {name:18s} " + " ".join(f"{inner[:, :, c, k].mean():7.0f}" for c in range(8)))
    tot = seg[:, :, 1:7, :].sum(-1).mean()
    print(f"    one phase (1-6)    {tot:7.0f} ticks; MFMA issue needs 2 x 36 x 32 = 2304 cycles of the SIMD's matrix pipe per wave and phase")
    out = (t[:, :, 52] - t[:, :, 51]).mean()
    print(f"  output stage {out:.0f} ticks")
    # the two workgroups of a CU: same XCC, SE, CU; SIMD by wave
    hw, xcc = t[:, :, 63], t[:, :, 62] & 0xf
    cu = ((xcc << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 0xf))[:, 0]
    slot = (hw & 0xf)[:, 0]
    print(f"  wave slots used on the SIMDs: {np.bincount(slot.astype(int))}")
    shown = 0
    for cid in np.unique(cu):
        idx = np.nonzero(cu == cid)[0]
        if len(idx) < 2:
            continue
        # pairs that overlap in time
        for i in range(len(idx)):
            for j in range(i + 1, len(idx)):
                A, Bg = idx[i], idx[j]
                lo, hi = max(t[A, 0, 0], t[Bg, 0, 0]), min(t[A, 0, 52], t[Bg, 0, 52])
                if hi - lo < 0.5 * life.mean() or shown >= 3:
                    continue
                shown += 1
                print(f"  CU {cid:#x}: workgroups {first + A} (slot {slot[A]}) and {first + Bg} (slot {slot[Bg]}) overlap for {hi - lo} ticks; wave 0, "
                      f"ticks from the older one's start:")
                base = min(t[A, 0, 0], t[Bg, 0, 0])
                for g in (A, Bg):
                    ks = []
                    for c in range(8):
                        ks.append(f"[{t[g, 0, 5 + 6 * c] - base}-{t[g, 0, 6 + 6 * c] - base}] T [{t[g, 0, 8 + 6 * c] - base}-{t[g, 0, 9 + 6 * c] - base}]")
                    print(f"    wg {first + g}: start {t[g, 0, 0] - base}, K steps " + " ".join(ks) + f" end {t[g, 0, 52] - base}")
    # share of the pipe: fraction of the traced span in which a wave of a SIMD is inside a K step, per SIMD, for fully covered CUs
    both = 0
    none = 0
    total = 0
    for cid in np.unique(cu):
        idx = np.nonzero(cu == cid)[0]
        if len(idx) < 4:
            continue
        lo, hi = t[idx, 0, 0].min(), t[idx, 0, 52].max()
        n = int(hi - lo)
        if n <= 0 or n > 50_000_000:
            continue
        busy = np.zeros(n, np.int8)
        for g in idx:
            iv = [(t[g, 0, 2], t[g, 0, 3])]
            for c in range(8):
                iv.append((t[g, 0, 5 + 6 * c], t[g, 0, 6 + 6 * c]))
                iv.append((t[g, 0, 8 + 6 * c], t[g, 0, 9 + 6 * c]))
            for s0, s1 in iv:
                busy[int(s0 - lo):int(s1 - lo)] += 1
        # only the stretch in which two workgroups are resident
        res_cnt = np.zeros(n, np.int8)
        for g in idx:
            res_cnt[int(t[g, 0, 0] - lo):int(t[g, 0, 52] - lo)] += 1
        m = res_cnt >= 2
        total += int(m.sum())
        both += int(((busy >= 2) & m).sum())
        none += int(((busy == 0) & m).sum())
    if total:
        print(f"  SIMD 0 of the CUs with >= 4 traced workgroups, while two workgroups are resident: both waves inside a K step "
              f"{100.0 * both / total:.1f} % of the time, exactly one {100.0 * (total - both - none) / total:.1f} %, none {100.0 * none / total:.1f} %")


if __name__ == "__main__":
    main()
