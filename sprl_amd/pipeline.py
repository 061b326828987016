"""Self-play <-> training in ONE process on the GPU — SURVEY §8(f) rank 1.

The reference couples its two halves through a shared directory: workers poll for `traced_<run>_iteration_<i>.pt`
every 30 s (cpp/src/selfplay/GridWorker.hpp:35-55), the controller polls for the three `.npy` files every 10 s
(scripts/othello_controller.py:66-125) and re-reads fp32-expanded samples from disk.  Here one loop owns both: the
engine's compact records are expanded once, go to HBM as tensors (the trainer's replay window), and the newly
trained model is handed back to the engine with `set_model` — no polling, no `.npy` round trip.  The reference file
layout can still be written alongside (`write_files=True`) so the reference tooling keeps working.
"""
import os
import tempfile
from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np
import torch

from . import engine as E
from . import trainer as T
from .network import GAME_SHAPES, GridResNet, trace_to_file

ENGINE_GAME = {"othello": "othello", "connect_four": "connect_four", "go7": "go7", "go9": "go9", "go19": "go19"}


@dataclass
class LoopConfig:
    game: str = "othello"
    num_iters: int = 25
    init_games: int = 10                   # iteration 0 budgets (othello_controller.py:31-34, OTHWorker.cpp:17-20)
    init_traversals: int = 2048
    init_max_batch: int = 1
    init_max_queue: int = 1
    games: int = 5                         # steady-state budgets (:36-39)
    traversals: int = 512
    max_batch: int = 8
    max_queue: int = 4
    concurrent_games: Optional[int] = None
    num_blocks: int = 2                    # :44-45
    num_channels: int = 64
    seed: int = 1
    device: int = 0
    run_name: str = "run"
    root: Optional[str] = None             # where models (and optional record files) go; temp dir when None
    write_files: bool = False


class SelfPlayTrainLoop:
    """engine.run -> records -> HBM tensors -> train_network -> traced model -> engine.set_model, per iteration."""

    def __init__(self, cfg: LoopConfig, trainer_cfg: Optional[T.TrainerConfig] = None, lib=None,
                 forward_factory: Optional[Callable] = None, train_device: Optional[str] = None, log=print):
        self.cfg, self.tcfg = cfg, trainer_cfg or T.TrainerConfig()
        self.lib = lib or E.load_library()
        self.log = log
        self.forward_factory = forward_factory      # tests on the CPU emulator: callable(net) -> engine forward callback
        rows, cols, actions, hist = GAME_SHAPES[cfg.game]
        self.net = GridResNet(rows, cols, actions, hist, cfg.num_blocks, cfg.num_channels)
        self.train_device = train_device or (f"cuda:{cfg.device}" if torch.cuda.is_available() else "cpu")
        self.window = T.ReplayWindow(self.tcfg, self.train_device)
        self.root = cfg.root or tempfile.mkdtemp(prefix="sprl_loop_")
        os.makedirs(os.path.join(self.root, "data", "models", cfg.run_name), exist_ok=True)
        self.model_path = None                      # None = iteration 0: the built-in initial evaluator ("random")
        self.next_stream = 1
        self.history = []

    def _engine(self, iteration):
        c = self.cfg
        first = iteration == 0
        games = c.init_games if first else c.games
        kw = dict(device=c.device, concurrent_games=min(c.concurrent_games or games, games),
                  num_traversals=c.init_traversals if first else c.traversals,
                  max_batch=c.init_max_batch if first else c.max_batch,
                  max_queue=c.init_max_queue if first else c.max_queue, seed=c.seed, stream_base=self.next_stream)
        self.next_stream += games
        return E.Engine(E.default_config(ENGINE_GAME[c.game], self.lib, **kw), self.lib), games

    def self_play(self, iteration):
        eng, games = self._engine(iteration)
        if self.model_path is None:
            eng.set_model("random")                 # GridWorker.hpp:125-127
        elif self.forward_factory is not None:
            eng.set_forward(self.forward_factory(self.net))
        else:
            eng.set_model(self.model_path)          # hot swap: no polling for the file
        rec = eng.run(games)
        states, dists, outcomes = rec.expand()
        if self.cfg.write_files:
            d = os.path.join(self.root, "data", "games", self.cfg.run_name, "0", "0")
            os.makedirs(d, exist_ok=True)
            rec.write_npy(os.path.join(d, f"{self.cfg.run_name}_iteration_{iteration}"))
        stats = eng.stats()
        rec.close()
        eng.close()
        return states, dists, outcomes, stats

    def step(self, iteration):
        states, dists, outcomes, stats = self.self_play(iteration)
        self.window.add(iteration, states, dists, outcomes)
        lr = T.learning_rate_for(self.tcfg, iteration)
        best, hist = T.train_network(self.net, lr, self.window.training_tensors(iteration), self.tcfg)
        self.model_path = os.path.join(self.root, "data", "models", self.cfg.run_name,
                                       f"traced_{self.cfg.run_name}_iteration_{iteration}.pt")
        T.export_best(self.net, best, self.cfg.game, self.model_path)
        rec = dict(iteration=iteration, samples=int(states.shape[0]), games=stats["games"], lr=lr,
                   best_epoch=hist["best_epoch"], best_val=hist["best_val"], model=self.model_path)
        self.history.append(rec)
        self.log(f"iteration {iteration}: {rec['games']} games, {rec['samples']} samples, best val {rec['best_val']:.4f} "
                 f"@ epoch {rec['best_epoch']}")
        return rec

    def run(self):
        for it in range(self.cfg.num_iters):
            self.step(it)
        return self.history
