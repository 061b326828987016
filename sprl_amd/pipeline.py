"""Self-play <-> training in ONE process on the GPU — SURVEY §8(f) rank 1.

The reference couples its two halves through a shared directory: workers poll for `traced_<run>_iteration_<i>.pt`
every 30 s (cpp/src/selfplay/GridWorker.hpp:35-55), the controller polls for the three `.npy` files every 10 s
(scripts/othello_controller.py:66-125) and re-reads fp32-expanded samples from disk.  Here one loop owns both: the
engine's compact records are expanded ON THE DEVICE straight into the tensors of the trainer's replay window
(sprl_engine_expand_records: no host copy of the records, no fp32 x8 expansion on the CPU), and the newly trained
network goes back to the engine through memory (sprl_engine_set_model_buffer: a TorchScript archive in a buffer, no
`.pt` file, no polling).  One engine serves all steady-state iterations.  The reference file layout can still be
written alongside (`write_files=True`) so the reference tooling keeps working.
"""
import os
import tempfile
import time
from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np
import torch

from . import engine as E
from . import trainer as T
from .network import GAME_SHAPES, GridResNet

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
        self.root = cfg.root or (tempfile.mkdtemp(prefix="sprl_loop_") if cfg.write_files else None)
        if cfg.write_files:
            os.makedirs(os.path.join(self.root, "data", "models", cfg.run_name), exist_ok=True)
        self.model_path = None                      # write_files only: the traced file of the last iteration (reference tooling)
        self.traced = None                          # None = iteration 0: the built-in initial evaluator ("random"); afterwards
                                                    # the TorchScript archive of the last trained network, in memory (bytes)
        self._eng, self._eng_sig = None, None
        self.next_stream = 1
        self.history = []

    def _engine(self, iteration):
        c = self.cfg
        first = iteration == 0
        games = c.init_games if first else c.games
        sig = (min(c.concurrent_games or games, games), c.init_traversals if first else c.traversals,
               c.init_max_batch if first else c.max_batch, c.init_max_queue if first else c.max_queue)
        if self._eng is None or sig != self._eng_sig:       # iteration 0 has its own budgets; afterwards the engine is kept
            if self._eng is not None:
                self._eng.close()
            kw = dict(device=c.device, concurrent_games=sig[0], num_traversals=sig[1], max_batch=sig[2], max_queue=sig[3],
                      seed=c.seed, stream_base=self.next_stream)
            self._eng = E.Engine(E.default_config(ENGINE_GAME[c.game], self.lib, **kw), self.lib)
            self._eng_sig = sig
        self.next_stream += games
        return self._eng, games

    def _sync(self):
        if torch.cuda.is_available() and str(self.train_device).startswith("cuda"):
            torch.cuda.synchronize()

    def self_play(self, iteration):
        """Returns (states, dists, outcomes, stats); stats carries the stage times of this call: `t_swap` (the new model handed
        to the engine: sprl_engine_set_model_buffer, i.e. TorchScript load + weight transforms), `t_selfplay` (the games) and
        `t_ingest` (records expanded on the device into the sample tensors)."""
        eng, games = self._engine(iteration)
        t0 = time.perf_counter()
        if self.traced is None:
            eng.set_model("random")                 # GridWorker.hpp:125-127
        elif self.forward_factory is not None:
            eng.set_forward(self.forward_factory(self.net))
        else:
            eng.set_model_bytes(self.traced)        # hot swap through memory: no file, no polling
        t1 = time.perf_counter()
        eng.begin(games)
        done = 0
        while done < games:
            done, _ = eng.step(64)
        t2 = time.perf_counter()
        _, samples, _ = eng.records_info()
        rows, cols, actions, hist = GAME_SHAPES[self.cfg.game]
        # the engine's "device" is the GPU for the product library, host memory for the CPU emulator build used in tests
        dev = torch.device(self.train_device) if self.lib.sprl_device_available() and torch.cuda.is_available() and \
            getattr(self.lib, "_name", "").endswith("libsprl_amd.so") else torch.device("cpu")
        states = torch.empty((samples, 2 * hist + 1, rows, cols), dtype=torch.float32, device=dev)
        dists = torch.empty((samples, actions), dtype=torch.float32, device=dev)
        outcomes = torch.empty((samples,), dtype=torch.float32, device=dev)
        eng.expand_records_into(states.data_ptr(), dists.data_ptr(), outcomes.data_ptr(), samples)
        if self.cfg.write_files:
            rec = eng.collect()
            d = os.path.join(self.root, "data", "games", self.cfg.run_name, "0", "0")
            os.makedirs(d, exist_ok=True)
            rec.write_npy(os.path.join(d, f"{self.cfg.run_name}_iteration_{iteration}"))
            rec.close()
        else:
            eng.finish()
        self._sync()
        t3 = time.perf_counter()
        stats = dict(eng.stats(), games=games, t_swap=t1 - t0, t_selfplay=t2 - t1, t_ingest=t3 - t2)     # (a kept engine's counters run on across iterations)
        return states, dists, outcomes, stats

    def step(self, iteration):
        """One iteration of the reference controller + worker fleet (scripts/othello_controller.py:243-343): self-play with the
        current network, ingest, train, export.  The record carries the wall time of every stage:
        t_swap / t_selfplay / t_ingest (see self_play), t_window (replay window concatenated), t_train (train_network: epochs x
        optimiser steps + validation), t_export (best weights traced to an in-memory TorchScript archive), t_total."""
        t_begin = time.perf_counter()
        states, dists, outcomes, stats = self.self_play(iteration)
        t0 = time.perf_counter()
        self.window.add(iteration, states, dists, outcomes)
        lr = T.learning_rate_for(self.tcfg, iteration)
        tensors = self.window.training_tensors(iteration)
        self._sync()
        t1 = time.perf_counter()
        best, hist = T.train_network(self.net, lr, tensors, self.tcfg)
        self._sync()
        t2 = time.perf_counter()
        # the best-validation weights traced into a TorchScript archive IN MEMORY (what othello_controller.py:237-239 writes to
        # traced_<run>_iteration_<i>.pt); the file itself only exists with write_files=True, for the reference tooling
        self.traced = T.export_best_bytes(self.net, best, self.cfg.game) if self.forward_factory is None else True
        if self.cfg.write_files:
            self.model_path = os.path.join(self.root, "data", "models", self.cfg.run_name,
                                           f"traced_{self.cfg.run_name}_iteration_{iteration}.pt")
            if self.traced is True:
                T.export_best(self.net, best, self.cfg.game, self.model_path)
            else:
                with open(self.model_path + ".tmp", "wb") as f:
                    f.write(self.traced)
                os.replace(self.model_path + ".tmp", self.model_path)
        t3 = time.perf_counter()
        n_window = int(tensors[0].shape[0])
        epochs = len(hist["epochs"])
        steps_per_epoch = (int((1.0 - self.tcfg.val_fraction) * n_window) + self.tcfg.batch_size - 1) // self.tcfg.batch_size
        rec = dict(iteration=iteration, samples=int(states.shape[0]), games=stats["games"], lr=lr,
                   best_epoch=hist["best_epoch"], best_val=hist["best_val"], model=self.model_path,
                   window_samples=n_window, epochs=epochs, optimiser_steps=epochs * steps_per_epoch,
                   t_swap=stats["t_swap"], t_selfplay=stats["t_selfplay"], t_ingest=stats["t_ingest"], t_window=t1 - t0,
                   t_train=t2 - t1, t_export=t3 - t2, t_total=t3 - t_begin)
        self.history.append(rec)
        self.log(f"iteration {iteration}: {rec['games']} games, {rec['samples']} samples, best val {rec['best_val']:.4f} "
                 f"@ epoch {rec['best_epoch']}; swap {rec['t_swap']:.2f} s, self-play {rec['t_selfplay']:.2f} s, ingest "
                 f"{rec['t_ingest']:.3f} s, window {rec['t_window']:.3f} s ({n_window} samples), train {rec['t_train']:.2f} s "
                 f"({epochs} epochs x {steps_per_epoch} steps), export {rec['t_export']:.2f} s, total {rec['t_total']:.2f} s")
        return rec

    def run(self):
        for it in range(self.cfg.num_iters):
            self.step(it)
        self.close()
        return self.history

    def close(self):
        if self._eng is not None:
            self._eng.close()
            self._eng = None
