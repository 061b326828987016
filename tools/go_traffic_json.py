#!/usr/bin/env python3
"""profiles/conv_kernel_traffic_<game>.json from the PMC summaries of tools/go_conv_pmc.sh (gpurun_out/<tag>_<game>_conv_pmc_*):
HBM bytes per launch of the any-board trunk convolution = 2 x FETCH_SIZE (gfx950 reports half of a wide streaming read,
MI355X_MICROARCH.md, HBM) + WRITE_SIZE, the boards per launch of the profiled rounds, and the algorithmic bytes (layout T: every
launch reads 64 channels x padded cells per board, writes as much, every second launch reads a residual of the same size).
    python tools/go_traffic_json.py <tag> <go9|go19>"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, game = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out")
if not os.path.exists(os.path.join(src, f"{tag}_{game}_conv_pmc_fetch_summary.csv")):
    src = os.path.join(ROOT, "profiles")


def summary(name):
    out = {}
    with open(os.path.join(src, name), newline="") as f:
        for row in csv.DictReader(f):
            out.setdefault(row["kernel"], {})[row["counter"]] = (int(row["dispatches"]), float(row["mean_per_dispatch"]))
    return out


run = None
for line in open(os.path.join(src, f"{tag}_{game}_conv_pmc_fetch.log")):
    if line.startswith("JSON "):
        run = json.loads(line[5:])
fetch, write = summary(f"{tag}_{game}_conv_pmc_fetch_summary.csv"), summary(f"{tag}_{game}_conv_pmc_write_summary.csv")
width = 9 if game == "go9" else 19
tile = 3 if game == "go9" else 4
tiles = ((width + tile - 1) // tile) ** 2
board_bytes = 64 * tile * tile * tiles * 4                      # layout T: sprl_wino_t_board_floats
tot_n, tot_b, per = 0, 0.0, {}
for k in fetch:
    n, f_kb = fetch[k]["FETCH_SIZE"]
    _, w_kb = write[k]["WRITE_SIZE"]
    hbm = (2.0 * f_kb + w_kb) * 1024.0
    res = ", 1, 1>" in k.replace(" ", "").replace(",", ", ") or "1,1>" in k.replace(" ", "")
    per[k] = {"launches": n, "FETCH_SIZE_KB_per_launch": f_kb, "WRITE_SIZE_KB_per_launch": w_kb, "hbm_bytes_per_launch": hbm}
    tot_n += n
    tot_b += hbm * n
# launches of the profiled run: warm + timed rounds, 12 per forward; boards per launch from the timed rounds' evaluations
boards = run["nn_evals"] / max(1, run["rounds"])
algo = boards * board_bytes * 2.5                                # in + out, + a residual on every second launch
out = {"kernel": "wino_conv64_nchw_kernel (layout T)", "game": game,
       "command": f"tools/go_conv_pmc.sh {tag} {game}: python tools/go_bench.py --only {game} --cnn-only (one population, {run['games']} resident games), "
                  "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes",
       "per_variant": per, "fetch_correction": 2.0, "write_correction": 1.0,
       "hbm_bytes_per_launch": tot_b / max(1, tot_n), "boards_per_launch": boards, "algorithmic_hbm_bytes_per_launch": algo,
       "traffic_over_algorithmic": tot_b / max(1, tot_n) / algo,
       "note": "the launch count includes the warm-up rounds (smaller batches at the start of the games), so hbm_bytes_per_launch is a "
               "mean over slightly smaller launches than boards_per_launch; bench.py scales it by the boards of its own launches",
       "sources": [f"profiles/{tag}_{game}_conv_pmc_fetch_summary.csv", f"profiles/{tag}_{game}_conv_pmc_write_summary.csv"]}
json.dump(out, open(os.path.join(ROOT, "profiles", f"conv_kernel_traffic_{game}.json"), "w"), indent=1)
print(game, "conv traffic / algorithmic", round(out["traffic_over_algorithmic"], 3), "boards per launch", round(boards, 1))
