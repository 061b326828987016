"""Throughput of the Go configurations (BASELINE configs 4-5: 9x9 and 19x19, 1600 iterations/move, batch 16 / queue 8) on
one GPU, over a fixed number of search rounds in the middle of the games (a 19x19 game lasts up to 722 plies):
traversals/s, expansions/s, moves/s, for the in-kernel uniform evaluator and for a traced CNN through the LibTorch path.
Secondary measurement; the headline bench is bench.py."""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprl_amd import engine as E  # noqa: E402
from sprl_amd.network import make_network, trace_to_file  # noqa: E402


LIB = [None]


def measure(game, games, model, rounds, warm):
    cfg = E.default_config(game, LIB[0], concurrent_games=games, num_traversals=1600, seed=3)
    eng = E.Engine(cfg, LIB[0])
    eng.set_model(model)
    eng.begin(games)
    eng.step(warm)
    s0 = eng.stats()
    t = time.time()
    eng.step(rounds)
    s1 = eng.stats()                           # (reads the device counters: waits for the rounds just queued)
    dt = time.time() - t
    d = {k: s1[k] - s0[k] for k in ("traversals", "expansions", "plies", "nn_evals", "kernel_launches")}
    info = eng.evaluator_info()
    eng.close()
    return dt, d, info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games9", type=int, default=1024)
    ap.add_argument("--games19", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=400)
    ap.add_argument("--warm", type=int, default=20, help="rounds played before the timed ones (100 rounds = one move)")
    ap.add_argument("--no-cnn", action="store_true")
    ap.add_argument("--cnn-only", action="store_true")
    ap.add_argument("--only", default=None, choices=["go9", "go19"])
    ap.add_argument("--lib", default=None, help="alternative libsprl_amd.so (A/B measurements)")
    ap.add_argument("--blocks", type=int, default=6)          # go_controller.py: MODEL_NUM_BLOCKS = 6
    a = ap.parse_args()
    if a.lib:
        LIB[0] = E.load_library(a.lib)
    with tempfile.TemporaryDirectory() as td:
        for game, games in (("go9", a.games9), ("go19", a.games19)):
            if a.only and game != a.only:
                continue
            cnn = trace_to_file(make_network(game, a.blocks, 64, seed=0), os.path.join(td, f"{game}.pt"), game)
            for name, model, rounds in (("uniform evaluator (in kernel)", "random", a.rounds), ("traced CNN", cnn, a.rounds // 4)):
                if (a.no_cnn and model != "random") or (a.cnn_only and model == "random"):
                    continue
                dt, d, info = measure(game, games, model, rounds, a.warm)
                print(f"{game}: {games} games, {name} [{info}]: {rounds} rounds in {dt:.2f} s -> "
                      f"{d['traversals'] / dt / 1e6:.2f} M traversals/s, {d['expansions'] / dt / 1e6:.2f} M expansions/s, "
                      f"{d['traversals'] / dt / 1600:.0f} moves/s, {d['nn_evals'] / dt / 1e6:.2f} M evals/s", flush=True)
                print("JSON " + json.dumps(dict(game=game, games=games, evaluator=name, rounds=rounds, seconds=dt, **d)), flush=True)


if __name__ == "__main__":
    main()
