#!/usr/bin/env python3
"""Timeline statistics of a rocprofv3 --kernel-trace CSV (kernel_trace.csv): how the launches of the two populations' streams
share the GPU.  Prints, for the window [--skip, 1 - --skip] of the trace:
  * wall time, time with >= 1 kernel executing, time with >= 2 executing, idle time;
  * per kernel class: launches, mean duration, share of the wall, and the mean duration of the launches that ran with no other
    queue's kernel beside them against those that overlapped one;
  * per queue: the gaps between one launch's end and the next launch's start.
usage: tools/timeline.py <kernel_trace.csv> [--skip 0.1] [--json out.json]"""
import argparse
import csv
import json
import sys
from collections import defaultdict

CLASSES = [("wino_conv64_nchw", "conv_nchw"), ("wino_conv64_kernel", "conv"), ("step_kernel_wide", "tree_wide"), ("step_kernel", "tree"),
           ("step_match", "tree"), ("stem_", "stem"), ("tail_", "tail"), ("policy_fc", "tail"), ("leaf_gather", "scan_gather"),
           ("leaf_scan", "scan_gather"), ("records_", "records")]


def classify(name):
    for key, cls in CLASSES:
        if key in name:
            return cls
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--skip", type=float, default=0.1, help="fraction of the trace dropped at each end")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    rows = []
    with open(args.csv, newline="") as f:
        rd = csv.DictReader(f)
        for r in rd:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            q = r.get("Queue_Id") or r.get("Stream_Id") or "0"
            rows.append((s, e, q, classify(r["Kernel_Name"])))
    if not rows:
        sys.exit("empty trace")
    rows.sort()
    t_first, t_last = rows[0][0], max(r[1] for r in rows)
    lo = t_first + args.skip * (t_last - t_first)
    hi = t_last - args.skip * (t_last - t_first)
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
    wall = hi - lo
    # sweep: number of kernels executing over time
    ev = []
    for s, e, q, c in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth_time = defaultdict(int)
    depth, last = 0, lo
    for t, dlt in ev:
        depth_time[depth] += t - last
        last = t
        depth += dlt
    depth_time[depth] += hi - last
    busy1 = sum(v for k, v in depth_time.items() if k >= 1)
    busy2 = sum(v for k, v in depth_time.items() if k >= 2)
    # per class; overlap with a kernel of ANOTHER queue: fraction of the launch's own duration covered
    by_q = defaultdict(list)
    for r in rows:
        by_q[r[2]].append(r)
    queues = sorted(by_q)
    other = {q: sorted((s, e) for qq in queues if qq != q for s, e, _, _ in by_q[qq]) for q in queues}
    import bisect
    stats = defaultdict(lambda: dict(n=0, sum=0, alone_n=0, alone_sum=0, ov_n=0, ov_sum=0, ov_cover=0.0))
    for q in queues:
        oth = other[q]
        starts = [s for s, _ in oth]
        for s, e, _, c in by_q[q]:
            st = stats[c]
            st["n"] += 1
            st["sum"] += e - s
            i = bisect.bisect_left(starts, s)
            cover = 0
            j = max(0, i - 4)
            while j < len(oth) and oth[j][0] < e:
                a, b = max(s, oth[j][0]), min(e, oth[j][1])
                if b > a:
                    cover += b - a
                j += 1
            frac = cover / max(1, e - s)
            if frac < 0.05:
                st["alone_n"] += 1
                st["alone_sum"] += e - s
            else:
                st["ov_n"] += 1
                st["ov_sum"] += e - s
                st["ov_cover"] += frac
    out = {"wall_ms": wall / 1e6, "busy_any_share": busy1 / wall, "busy_two_or_more_share": busy2 / wall, "idle_share": 1 - busy1 / wall,
           "queues": len(queues), "classes": {}, "gaps": {}}
    print(f"window {wall / 1e6:.1f} ms, {len(rows)} launches on {len(queues)} queues")
    print(f"  >= 1 kernel executing {busy1 / wall:6.1%}   >= 2 executing {busy2 / wall:6.1%}   idle {1 - busy1 / wall:6.1%}")
    print(f"  {'class':12s} {'launches':>9s} {'mean us':>9s} {'sum/wall':>9s} {'alone n':>8s} {'alone us':>9s} {'overl n':>8s} {'overl us':>9s} {'cover':>6s}")
    for c, st in sorted(stats.items(), key=lambda kv: -kv[1]["sum"]):
        an, on = max(1, st["alone_n"]), max(1, st["ov_n"])
        print(f"  {c:12s} {st['n']:9d} {st['sum'] / st['n'] / 1e3:9.1f} {st['sum'] / wall:9.1%} {st['alone_n']:8d} {st['alone_sum'] / an / 1e3:9.1f} "
              f"{st['ov_n']:8d} {st['ov_sum'] / on / 1e3:9.1f} {st['ov_cover'] / on:6.2f}")
        out["classes"][c] = {"launches": st["n"], "mean_us": st["sum"] / st["n"] / 1e3, "sum_over_wall": st["sum"] / wall,
                             "alone_launches": st["alone_n"], "alone_mean_us": st["alone_sum"] / an / 1e3,
                             "overlapped_launches": st["ov_n"], "overlapped_mean_us": st["ov_sum"] / on / 1e3,
                             "overlapped_mean_cover": st["ov_cover"] / on}
    for q in queues:
        seq = sorted(by_q[q])
        gaps = [seq[i + 1][0] - seq[i][1] for i in range(len(seq) - 1)]
        gaps = [g for g in gaps if g < 5e6]           # (step boundaries: host work between iterations)
        if not gaps:
            continue
        gaps.sort()
        tot = sum(gaps)
        print(f"  queue {q}: {len(seq)} launches, gaps mean {tot / len(gaps) / 1e3:.2f} us, median {gaps[len(gaps) // 2] / 1e3:.2f}, "
              f"p90 {gaps[int(0.9 * len(gaps))] / 1e3:.2f}, sum/wall {tot / wall:.1%}")
        out["gaps"][q] = {"launches": len(seq), "mean_us": tot / len(gaps) / 1e3, "median_us": gaps[len(gaps) // 2] / 1e3,
                          "p90_us": gaps[int(0.9 * len(gaps))] / 1e3, "sum_over_wall": tot / wall}
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
