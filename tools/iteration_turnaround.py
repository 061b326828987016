"""Whole-iteration turnaround on one MI355X: self-play -> ingest -> train -> hot swap, per iteration, at the
Othello BASELINE configuration (800 traversals/move, batch 8 / queue 4, the 2 x 64 network, the reference controller's training
constants: batch 1024, up to 10 x 10 epochs with the "best epoch is recent" stopping rule, window of the last 10 iterations,
scripts/othello_controller.py:31-55,128-241,292-343).  The reference plays 384 workers x 5 games = 1920 games per steady-state
iteration (OTHWorker.cpp:13-21); --games sets that number here.
    python tools/iteration_turnaround.py [--iters 3] [--games 1920] [--out gpurun_out/iteration_turnaround.txt]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--games", type=int, default=1920)
    ap.add_argument("--traversals", type=int, default=800)
    ap.add_argument("--init-traversals", type=int, default=800, help="the reference's iteration 0 searches 131072 traversals with the "
                    "uniform evaluator (OTHWorker.cpp:21); the steady-state budget here keeps the run short")
    ap.add_argument("--max-groups", type=int, default=10)
    ap.add_argument("--graph", action="store_true", help="TrainerConfig.use_graph: the optimiser step replayed from a captured HIP graph")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "iteration_turnaround.txt"))
    a = ap.parse_args()
    from sprl_amd import trainer as T
    from sprl_amd.pipeline import LoopConfig, SelfPlayTrainLoop
    cfg = LoopConfig(game="othello", num_iters=a.iters, init_games=a.games, init_traversals=a.init_traversals, init_max_batch=8,
                     init_max_queue=4, games=a.games, traversals=a.traversals, concurrent_games=a.games, num_blocks=2, num_channels=64)
    tcfg = T.TrainerConfig(max_groups=a.max_groups, use_graph=a.graph)
    lines = []
    loop = SelfPlayTrainLoop(cfg, tcfg, log=lambda s: (print(s, flush=True), lines.append(s)))
    hist = loop.run()
    keys = ("t_swap", "t_selfplay", "t_ingest", "t_window", "t_train", "t_export", "t_total")
    steady = hist[1:] or hist
    mean = {k: sum(h[k] for h in steady) / len(steady) for k in keys}
    summary = ("steady-state iteration (mean of iterations 1..): " + ", ".join(f"{k[2:]} {mean[k]:.2f} s" for k in keys) +
               f"; training = {100 * mean['t_train'] / mean['t_total']:.0f} % of the turnaround, self-play {100 * mean['t_selfplay'] / mean['t_total']:.0f} %")
    print(summary)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        f.write("\n".join(lines) + "\n" + summary + "\n" + json.dumps(hist) + "\n")


if __name__ == "__main__":
    main()
