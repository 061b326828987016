"""Throughput of device match play (heuristic vs random, Othello) with the reference harness timed beside it when
oracle/_ref is present.  Diagnostic, not the headline bench."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from sprl_amd import engine as E  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=2048)
    ap.add_argument("--traversals", type=int, default=200)
    ap.add_argument("--ref-games", type=int, default=4)
    a = ap.parse_args()
    cfg = E.default_config("othello", concurrent_games=a.games, num_traversals=a.traversals, dir_eps=0.25, dir_alpha=0.1,
                           u_weight=1.0, seed=5)
    t = time.time()
    w, _, n = E.play_match(cfg, dict(model="heuristic"), dict(model="random"), a.games)
    dt = time.time() - t
    print(f"device: {a.games} games, {a.traversals} traversals/move: {dt:.2f} s = {a.games / dt:.1f} games/s; "
          f"heuristic/random/draw = {E.match_score(w)}, mean plies {n.mean():.1f}")
    from oracle import pyref
    if a.ref_games and pyref.available():
        t = time.time()
        pyref.match("othello", 1, 0, a.ref_games, a.traversals, 8, 4, 1, 1, 1, 1, 5, 1, 160)
        dt = time.time() - t
        print(f"reference (1 core): {a.ref_games} games in {dt:.2f} s = {a.ref_games / dt:.2f} games/s")


if __name__ == "__main__":
    main()
