#!/usr/bin/env python3
"""Builds profiles/conv_kernel_traffic.json and profiles/tree_kernel_traffic.json from the PMC summaries that
tools/profile_round.sh <tag> wino_conv64_kernel and tools/profile_pmc.sh <tag>_tree leave in gpurun_out/ (per-kernel means of
FETCH_SIZE / WRITE_SIZE / SQ_* collected in separate rocprofv3 passes) and from the bench line of the kernel-trace pass.
    python tools/traffic_json.py r03p
Corrections as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE x 2 for wide (16 B/lane) streaming reads, WRITE_SIZE
exact; the tree kernel's scattered row reads use the factor calibrated with tools/calib_fetch.hip (profiles/r01c_calib_*)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out") if os.path.exists(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_fetch_summary.csv")) else os.path.join(ROOT, "profiles")


def summary(name):
    out = {}
    with open(os.path.join(src, name), newline="") as f:
        for row in csv.DictReader(f):
            out.setdefault(row["kernel"], {})[row["counter"]] = (int(row["dispatches"]), float(row["mean_per_dispatch"]))
    return out


def variant(kname):
    inside = kname.split("wino_conv64_kernel<", 1)[1].split(">", 1)[0].replace(" ", "").split(",")
    heads, res = int(inside[2]), int(inside[3]) if len(inside) > 3 else 1
    stem = int(inside[4]) if len(inside) > 4 else 0
    return f"HEADS={heads},RES={res},STEM={stem}", heads, res, stem


bench = json.load(open(os.path.join(src, f"{tag}_bench.json")))
rl = bench["roofline"]
boards = rl["boards_per_launch"]
fetch, write, sq = summary(f"{tag}_pmc_fetch_summary.csv"), summary(f"{tag}_pmc_write_summary.csv"), summary(f"{tag}_pmc_sq_summary.csv")
per, tot_b, tot_a, tot_n = {}, 0.0, 0.0, 0
sqv = {}
for k in fetch:
    if "wino_conv64_kernel<8, 8" not in k:
        continue
    name, heads, res, stem = variant(k)
    n, f_kb = fetch[k]["FETCH_SIZE"]
    _, w_kb = write[k]["WRITE_SIZE"]
    hbm = (2.0 * f_kb + w_kb) * 1024.0
    # per board and launch: 16 KB of activations in, + 16 KB residual (RES), + 16 KB out (or 768 B of head maps with HEADS)
    # (HEADS = 1: 768 B of head maps; HEADS = 2, round 4: the forward ends in the launch - 65 logits + 1 value = 264 B - and the
    # 50 KB of FC weights come from L2)
    # STEM = 1 (round 4): the launch reads the 768 B of input planes, writes x0 (16 KB) and reads it back through L2 (no HBM bytes)
    algo = boards * ((768 + 16384 if stem else 16384) + (16384 if res else 0) + (264 if heads == 2 else 768 if heads else 16384))
    per[name] = {"FETCH_SIZE_KB_per_launch": f_kb, "WRITE_SIZE_KB_per_launch": w_kb, "launches": n, "hbm_bytes_per_launch": hbm, "algorithmic": algo}
    tot_b += hbm * n
    tot_a += algo * n
    tot_n += n
    s = sq[k]
    xcd = s["GRBM_GUI_ACTIVE"][1] / 8.0
    sqv[name] = {"mfma_busy_share_of_simd_cycles": s["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (xcd * 1024.0), "xcd_cycles_per_launch": xcd,
                 "valu_per_mfma": s["SQ_INSTS_VALU"][1] / s["SQ_INSTS_MFMA"][1],
                 "wave_cycles_waiting_share": s["SQ_WAIT_INST_ANY"][1] / s["SQ_WAVE_CYCLES"][1],
                 "wave_cycles_issuing_share": s["SQ_ACTIVE_INST_ANY"][1] / s["SQ_WAVE_CYCLES"][1]}
kt = {}
with open(os.path.join(src, f"{tag}_bench_kernel_stats.csv"), newline="") as f:
    for row in csv.DictReader(f):
        kt[row["Name"]] = (int(row["Calls"]), float(row["AverageNs"]))
conv_calls = sum(c for k, (c, a) in kt.items() if "wino_conv64_kernel<8, 8" in k)
conv_avg = sum(c * a for k, (c, a) in kt.items() if "wino_conv64_kernel<8, 8" in k) / max(1, conv_calls) / 1e6
conv = {"kernel": "wino_conv64_kernel<8,8,HEADS,RES,STEM> (cnn_wino.hip; per forward of the 2-block net: STEM = 1 (the stem in its prologue), RES = 1, "
                  "RES = 0, and the last one with the head convolutions and the FC layers fused, HEADS = 2)",
        "command": f"python bench.py --populations 1 (4096 games, 800 it/move, CNN), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes "
                   f"(tools/profile_round.sh {tag} wino_conv64_kernel; tools/traffic_json.py {tag})",
        "per_variant": per, "fetch_correction": 2.0, "write_correction": 1.0,
        "correction_source": "MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read "
                             "(16 B/lane); WRITE_SIZE is exact for 16-B-per-lane streaming stores",
        "hbm_bytes_per_launch": tot_b / tot_n, "boards_per_launch": boards, "algorithmic_hbm_bytes_per_launch": tot_a / tot_n,
        "traffic_over_algorithmic": tot_b / tot_a, "sq": sqv, "rocprof_kernel_trace_avg_launch_ms": conv_avg,
        "bench_event_avg_launch_ms": rl["avg_launch_ms"],
        "sources": [f"profiles/{tag}_pmc_fetch_summary.csv", f"profiles/{tag}_pmc_write_summary.csv", f"profiles/{tag}_pmc_sq_summary.csv",
                    f"profiles/{tag}_bench_kernel_stats.csv", f"profiles/{tag}_bench.json"],
        "note": "launches include the rounds in which no leaf was queued (the grid is sized for the capacity and every workgroup leaves at once); "
                "GRBM_GUI_ACTIVE is summed over the 8 XCDs"}
json.dump(conv, open(os.path.join(ROOT, "profiles", "conv_kernel_traffic.json"), "w"), indent=1)
print("conv: traffic / algorithmic", round(conv["traffic_over_algorithmic"], 3), "MFMA busy", {k: round(v["mfma_busy_share_of_simd_cycles"], 3) for k, v in sqv.items()},
      "VALU/MFMA", {k: round(v["valu_per_mfma"], 2) for k, v in sqv.items()}, "avg launch ms", round(conv_avg, 4))

ttag = tag + "_tree"
if os.path.exists(os.path.join(src, f"{ttag}_pmc_fetch_summary.csv")):
    old = json.load(open(os.path.join(ROOT, "profiles", "tree_kernel_traffic.json")))
    f, w, s1, s2 = (summary(f"{ttag}_pmc_{x}_summary.csv") for x in ("fetch", "write", "sq", "sq2"))
    k = next(k for k in f if "step_kernel<Othello>" in k)
    n, f_kb = f[k]["FETCH_SIZE"]
    _, w_kb = w[k]["WRITE_SIZE"]
    fc, wc = old["fetch_correction"], old["write_correction"]
    hbm = (fc * f_kb + wc * w_kb) * 1024.0
    tr = bench["roofline_tree"]
    algo = tr["traversals_per_launch"] * tr["bytes_per_traversal"]
    tree = {"kernel": "step_kernel<Othello>", "command": f"python bench.py --populations 1 (4096 games, 800 it/move, CNN), rocprofv3 --pmc passes in separate runs (tools/profile_pmc.sh {ttag})",
            "FETCH_SIZE_KB_per_launch": f_kb, "WRITE_SIZE_KB_per_launch": w_kb, "fetch_correction": fc, "write_correction": wc, "calibration": old["calibration"],
            "hbm_bytes_per_launch": hbm, "launches": n, "algorithmic_bytes_per_launch": algo, "traffic_over_algorithmic": hbm / algo,
            "sq": {"waves_per_launch": s2[k]["SQ_WAVES"][1], "wave_cycles_waiting_share": s1[k]["SQ_WAIT_ANY"][1] / s1[k]["SQ_WAVE_CYCLES"][1],
                   "wave_cycles_issuing_share": s1[k]["SQ_ACTIVE_INST_ANY"][1] / s1[k]["SQ_WAVE_CYCLES"][1], "xcd_cycles_per_launch": s2[k]["GRBM_GUI_ACTIVE"][1] / 8.0,
                   "valu_insts": s1[k]["SQ_INSTS_VALU"][1], "salu_insts": s1[k]["SQ_INSTS_SALU"][1], "vmem_insts": s1[k]["SQ_INSTS_VMEM"][1]},
            "sources": [f"profiles/{ttag}_pmc_{x}_summary.csv" for x in ("fetch", "write", "sq", "sq2")] + ["profiles/r01c_calib_FETCH_SIZE_summary.csv"],
            "round2": {"hbm_bytes_per_launch": old["hbm_bytes_per_launch"], "launches": old["launches"]}}
    json.dump(tree, open(os.path.join(ROOT, "profiles", "tree_kernel_traffic.json"), "w"), indent=1)
    print("tree: traffic / algorithmic", round(tree["traffic_over_algorithmic"], 3), "waiting", round(tree["sq"]["wave_cycles_waiting_share"], 3))
