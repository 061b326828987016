"""Agent-vs-agent evaluation — the command line of the reference's Evaluate.exe (cpp/src/Evaluate.cpp:37-170) on the
device match engine (`sprl_match_play`): all games of the match run concurrently, two trees per game.

    python -m sprl_amd.evaluate <modelPath0> <modelPath1> <numGames> <numTraversals> <maxBatchSize> <maxQueueSize> \\
        <model0UseSymmetrize> <model0UseParentQ> <model1UseSymmetrize> <model1UseParentQ> [--game othello|connect_four|go7|go9|go19]

"random" as a model path selects the uniform evaluator (Evaluate.cpp:72-74); "heuristic" the Othello heuristic.
The reference binary is compiled for one game at a time (Evaluate.cpp:55-67); here it is the --game option.
"""
import argparse
import sys

import numpy as np

from . import engine as E


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("model0")
    ap.add_argument("model1")
    ap.add_argument("num_games", type=int)
    ap.add_argument("num_traversals", type=int)
    ap.add_argument("max_batch", type=int)
    ap.add_argument("max_queue", type=int)
    ap.add_argument("sym0", type=int)
    ap.add_argument("parent_q0", type=int)
    ap.add_argument("sym1", type=int)
    ap.add_argument("parent_q1", type=int)
    ap.add_argument("--game", default="connect_four", choices=["othello", "connect_four", "go7", "go9", "go19"])
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--seed", type=int, default=None, help="default: from the clock, like the reference")
    ap.add_argument("--concurrent", type=int, default=4096)
    a = ap.parse_args(argv)
    seed = a.seed if a.seed is not None else int(np.random.SeedSequence().entropy & 0x7fffffffffffffff)
    cfg = E.default_config(a.game, device=a.device, concurrent_games=min(a.concurrent, a.num_games),
                           num_traversals=a.num_traversals, max_batch=a.max_batch, max_queue=a.max_queue,
                           dir_eps=0.25, dir_alpha=0.1, u_weight=1.0, add_noise=1, seed=seed)   # Evaluate.cpp:94-112
    w, _, n = E.play_match(cfg, dict(model=a.model0, use_symmetry=a.sym0 > 0, parent_q=a.parent_q0 > 0),
                           dict(model=a.model1, use_symmetry=a.sym1 > 0, parent_q=a.parent_q1 > 0), a.num_games)
    w0, w1, d = E.match_score(w)
    print(f"Player 0 wins: {w0}")             # Evaluate.cpp:163-165
    print(f"Player 1 wins: {w1}")
    print(f"Draws: {d}")
    print(f"Average game length: {float(np.mean(n)):.1f} plies")
    return 0


if __name__ == "__main__":
    sys.exit(main())
