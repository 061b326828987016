"""Trainer step on the GPU — SURVEY §8(f) rank 2, the consumer side of the self-play records.

Mirrors `train_network` and the replay window of the reference controller
(scripts/othello_controller.py:128-241 and :292-343): recency-weighted cross-entropy on the tempered visit
distributions + recency-weighted MSE on the outcomes, AdamW, 90/10 split, best-validation snapshot, the
"stop unless the best epoch is recent" rule, LR decay at milestone iterations, last-N-iterations window with linear
weights.  MI355X-first differences: the window lives in HBM as tensors (no DataLoader, batches are index gathers
on the device), the best model is kept as an in-memory state_dict and exported once, and with
torch.distributed initialised (backend "nccl" = RCCL) every rank trains on the samples of its own game shard with
gradient all-reduce (DistributedDataParallel) — the reference's `prototype/ddp.py` ambition.
"""
import copy
import os
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from .network import trace_to_bytes, trace_to_file

EPS = 1e-8  # othello_controller.py:150

# Library tuning for the training step on ROCm (round 4, profiles/r04z_trainer_variants.txt: 2.60 -> 2.17 ms per step of batch 1024):
# MIOpen's im2col + GEMM convolution family works one image at a time (hundreds of tiny launches per step when its heuristic
# picks it) - left out of its choice, as the evaluator plugin does (torch_eval.cpp); and train_network lets the library MEASURE its
# solvers per shape (cudnn.benchmark) instead of guessing.  Both only select among the library's fp32 kernels.
os.environ.setdefault("MIOPEN_DEBUG_CONV_GEMM", "0")


@dataclass
class TrainerConfig:
    batch_size: int = 1024                 # othello_controller.py:52-55
    lr_init: float = 0.01
    lr_decay_factor: float = 0.1
    lr_milestone_iters: List[int] = field(default_factory=lambda: [5, 10, 20])
    max_groups: int = 10                   # :49-50
    epochs_per_group: int = 10
    num_past_iters_to_train: int = 10      # :47
    linear_weighting: bool = True          # :45
    val_fraction: float = 0.1              # :135-136
    fast_conv: bool = False                # the residual trunk's 3x3 convolutions (forward and backward-data) on the hand-written
                                           # Winograd / fp32-MFMA kernel instead of the library's (trainer_ops.WinoConv3x3).  Correct
                                           # (tests/test_gpu_cnn.py) and measured (profiles/r04zh_*, r04zj_*): the convolution itself
                                           # takes 21 us where the library's takes 90-120, but the two layout conversions and the filter
                                           # transform around every call bring the op to ~50 us and five launches - +5 % with the graph
                                           # replay, SLOWER in the eager loop.  Off until BatchNorm / ReLU / the residual add live in
                                           # layout W too
    use_graph: bool = True                 # MI355X: replay the optimiser step (gather, forward, losses, backward, AdamW) from ONE
                                           # captured HIP graph per full-size batch instead of ~100 kernel launches: the eager step
                                           # is bound by its launches (round 4: 6.5 -> 2.4 ms per step incl. validation on a GPU box
                                           # with slow host cores, profiles/r04h_trainer_*.txt).  Single GPU, CUDA device and the
                                           # random split only; otherwise, or when the capture fails, the eager loop runs.


def weighted_losses(logits, value, target_pdf, target_value, weight):
    """Policy: sum_i w_i * CE(pdf_i, softmax(logits_i)) / sum w; value: sum_i w_i (z_i - v_i)^2 / sum w
    (othello_controller.py:160-170), weights = sample timestamps."""
    p = torch.softmax(logits, dim=1)
    wsum = torch.sum(weight)
    policy = torch.sum(-torch.sum(target_pdf * torch.log(p + EPS), dim=1, keepdim=True) * weight) / wsum
    val = torch.sum((target_value - value) ** 2 * weight) / wsum
    return policy, val


class ReplayWindow:
    """The last `num_past_iters_to_train` iterations of samples, resident on `device`
    (othello_controller.py:292-338)."""

    def __init__(self, cfg: TrainerConfig, device):
        self.cfg, self.device = cfg, torch.device(device)
        self.items = []          # (states, dists, outcomes[N,1], timestamps[N,1]) per iteration

    def add(self, iteration, states, dists, outcomes):
        as_t = lambda a: (a if torch.is_tensor(a) else torch.from_numpy(a)).to(self.device, torch.float32)
        s, d, o = as_t(states), as_t(dists), as_t(outcomes).reshape(-1, 1)
        stamp = float(iteration + 1) if self.cfg.linear_weighting else 1.0      # :94
        t = torch.full((s.shape[0], 1), stamp, device=self.device)
        assert s.shape[0] == d.shape[0] == o.shape[0]
        self.items.append((s, d, o, t))
        while len(self.items) > self.cfg.num_past_iters_to_train:
            self.items.pop(0)

    def training_tensors(self, iteration):
        s = torch.cat([i[0] for i in self.items])
        d = torch.cat([i[1] for i in self.items])
        o = torch.cat([i[2] for i in self.items])
        t = torch.cat([i[3] for i in self.items])
        if self.cfg.linear_weighting:
            t = t - max(0, iteration + 1 - self.cfg.num_past_iters_to_train)    # :338
        assert torch.min(t) > 0
        return s, d, o, t


def learning_rate_for(cfg: TrainerConfig, iteration: int) -> float:
    """LR after the decays applied at the start of every milestone iteration up to `iteration` (:305-307)."""
    lr = cfg.lr_init
    for m in cfg.lr_milestone_iters:
        if iteration >= m:
            lr *= cfg.lr_decay_factor
    return lr


def _evaluate(net, tensors, batches):
    """Sums over validation batches of the two weighted losses as one float64 DEVICE tensor [policy sum, value sum, batch
    count]: nothing is read back per batch, the caller synchronises once per epoch."""
    s, d, o, t = tensors
    acc = torch.zeros(3, device=s.device, dtype=torch.float64)
    measure = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = False               # (forward-only shapes, the last one of a size of its own: no solver search)
    try:
        return _evaluate_batches(net, tensors, batches, acc)
    finally:
        torch.backends.cudnn.benchmark = measure


def _evaluate_batches(net, tensors, batches, acc):
    s, d, o, t = tensors
    with torch.no_grad():
        for b in batches:
            lo, va = net(s[b])
            p, v = weighted_losses(lo, va, d[b], o[b], t[b])
            acc[0] += p.double()
            acc[1] += v.double()
            acc[2] += 1.0
    return acc


class _GraphStep:
    """One optimiser step - gather the batch from the HBM-resident window, forward, the two weighted losses, backward, AdamW, loss
    sums - captured once as a HIP graph (torch.cuda.CUDAGraph) and replayed per batch: the step of the 2 x 64 network is a chain
    of ~100 short kernels and the stock eager loop is bound by their launches, not by their execution (profiles/r04*_trainer_*).
    The only per-batch input is the index vector, copied into a static buffer.  Capture needs a few warm-up steps (library
    kernel selection must not happen inside a capture); they are real optimiser steps, so parameters, BatchNorm statistics and the
    AdamW state are put back IN PLACE afterwards - the captured graph holds their addresses."""

    def __init__(self, net, opt, tensors, batch_size, warm_idx):
        s, d, o, t = tensors
        dev = s.device
        self.idx = warm_idx.clone()
        self.acc = torch.zeros(2, device=dev, dtype=torch.float64)
        saved = {k: v.clone() for k, v in net.state_dict().items()}

        def body():
            i = self.idx
            lo, va = net(s[i])
            pl, vl = weighted_losses(lo, va, d[i], o[i], t[i])
            opt.zero_grad(set_to_none=True)
            (pl + vl).backward()
            opt.step()
            self.acc[0] += pl.detach().double()
            self.acc[1] += vl.detach().double()

        net.train()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                body()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            body()
        with torch.no_grad():                          # undo the warm-up steps, in place
            for k, v in net.state_dict().items():
                v.copy_(saved[k])
            for st in opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
        self.acc.zero_()

    def __call__(self, batch_idx):
        self.idx.copy_(batch_idx)
        self.graph.replay()

    def drain(self, tacc):
        """once per epoch: the loss sums the replays have accumulated in the graph's static buffer"""
        tacc += self.acc
        self.acc.zero_()


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else None


def train_network(net, learning_rate, tensors, cfg: TrainerConfig, generator: Optional[torch.Generator] = None,
                  ddp: bool = False, log=None, index_plan=None):
    """One controller iteration of training (scripts/othello_controller.py:128-241).  Returns (best_state_dict, history);
    `net` itself keeps training weights (the reference traces the best-validation snapshot but carries the live network into
    the next iteration).

    index_plan: optional callable epoch -> (train_batches, val_batches), lists of index tensors; replaces the random 90/10
    split and the per-epoch shuffles (tests replay the batches the reference controller drew, tests/golden/g_trainer.npz).

    ddp=True with torch.distributed initialised (backend "nccl" = RCCL): every rank trains on the samples of its own game
    shard with gradient all-reduce.  Shards differ in size (game lengths vary), so the number of optimiser steps per epoch is
    agreed first (MAX over ranks; shorter shards wrap around) and the validation sums are all-reduced before the best-epoch
    and early-stop decisions: every rank takes the same decisions and exports the same weights."""
    s, d, o, t = tensors
    device = s.device
    if device.type == "cuda" and not torch.backends.cudnn.benchmark:
        torch.backends.cudnn.benchmark = True                                   # (see the note at the top of the file)
        try:
            return train_network(net, learning_rate, tensors, cfg, generator, ddp, log, index_plan)
        finally:
            torch.backends.cudnn.benchmark = False
    if cfg.fast_conv and device.type == "cuda" and not ddp and not getattr(net, "_sprl_fast_trunk", False):
        from . import trainer_ops
        if trainer_ops.available():
            net._sprl_fast_trunk = True
            try:
                with trainer_ops.fast_trunk(net):
                    return train_network(net, learning_rate, tensors, cfg, generator, ddp, log, index_plan)
            finally:
                del net._sprl_fast_trunk
    net.to(device)
    n = s.shape[0]
    dist = _dist() if ddp else None
    if index_plan is None:
        perm = torch.randperm(n, device=device, generator=generator)
        n_train = int((1.0 - cfg.val_fraction) * n)                              # :135-138
        train_idx, val_idx = perm[:n_train], perm[n_train:]
        steps = (train_idx.numel() + cfg.batch_size - 1) // cfg.batch_size
        smallest = train_idx.numel()
        if dist is not None:                                                     # same number of all-reduces on every rank;
            # the empty-shard error is agreed COLLECTIVELY in the same all-reduce: a rank that raised alone would
            # leave the others blocked in this collective until the RCCL watchdog fires.  [max steps, -min training samples]
            st = torch.tensor([steps, -train_idx.numel()], device=device)
            dist.all_reduce(st, op=dist.ReduceOp.MAX)
            steps, smallest = int(st[0].item()), -int(st[1].item())
        if smallest == 0:
            raise ValueError(f"train_network: a rank's shard (this rank's: {n} samples) leaves nothing for training after the "
                             f"{cfg.val_fraction:.0%} validation split")
    model = net
    if dist is not None:
        from torch.nn.parallel import DistributedDataParallel
        model = DistributedDataParallel(net, device_ids=[device.index] if device.type == "cuda" else None)
    graph_step = None
    opt = None
    if cfg.use_graph and device.type == "cuda" and dist is None and index_plan is None and train_idx.numel() >= cfg.batch_size:
        before = copy.deepcopy(net.state_dict())
        try:
            opt = torch.optim.AdamW(model.parameters(), lr=learning_rate, capturable=True)
            graph_step = _GraphStep(net, opt, tensors, cfg.batch_size, train_idx[:cfg.batch_size])
        except Exception as exc:                                                 # (an optimisation only: same arithmetic either way)
            if log:
                log(f"HIP graph capture of the optimiser step failed ({exc}); eager steps")
            net.load_state_dict(before)
            graph_step, opt = None, None
    if opt is None:
        opt = torch.optim.AdamW(model.parameters(), lr=learning_rate)           # :144
    best_val, best_epoch, best_state = float("inf"), 0, copy.deepcopy(net.state_dict())
    history = []
    for group in range(cfg.max_groups):
        for epoch in range(cfg.epochs_per_group):
            e = epoch + group * cfg.epochs_per_group
            model.train()
            if index_plan is not None:
                train_batches, val_batches = index_plan(e)
            else:
                order = train_idx[torch.randperm(train_idx.numel(), device=device, generator=generator)]
                if order.numel() < steps * cfg.batch_size and dist is not None:     # wrap a short shard around
                    order = order.repeat((steps * cfg.batch_size + order.numel() - 1) // order.numel())[:steps * cfg.batch_size]
                train_batches = [order[k:k + cfg.batch_size] for k in range(0, order.numel(), cfg.batch_size)]
                val_batches = [val_idx[k:k + cfg.batch_size] for k in range(0, val_idx.numel(), cfg.batch_size)]
            # losses are accumulated on the device in float64 (the same sums as the reference's Python floats, :181-182) and read
            # back ONCE per epoch: the reference's `.item()` per batch would drain the GPU queue after every optimiser step
            tacc = torch.zeros(2, device=device, dtype=torch.float64)
            nb = 0
            for b in train_batches:
                if graph_step is not None and b.numel() == cfg.batch_size:
                    graph_step(b)
                    nb += 1
                    continue
                # the library measures its solvers per SHAPE: worth it for the full-size batch every step uses, not for the one
                # short batch at the end of an epoch whose size changes with the window (a search costs about a second per shape)
                measure = torch.backends.cudnn.benchmark
                torch.backends.cudnn.benchmark = measure and b.numel() == cfg.batch_size
                lo, va = model(s[b])
                pl, vl = weighted_losses(lo, va, d[b], o[b], t[b])
                opt.zero_grad(set_to_none=True)
                (pl + vl).backward()
                opt.step()
                tacc[0] += pl.detach().double()
                tacc[1] += vl.detach().double()
                nb += 1
                torch.backends.cudnn.benchmark = measure
            if graph_step is not None:
                graph_step.drain(tacc)
            model.eval()
            if dist is not None:                                                 # BatchNorm running statistics: average the ranks'
                for buf in net.buffers():                                        # (the parameters are already identical)
                    if buf.is_floating_point():
                        dist.all_reduce(buf)
                        buf /= dist.get_world_size()
            # [val policy, val value, val batches, train policy, train value, train batches]; a rank without validation batches
            # contributes zeros, and the training sums stand in only when NO rank has any (a tiny window)
            sums = torch.cat([_evaluate(net, tensors, val_batches), tacc, torch.tensor([float(nb)], device=device, dtype=torch.float64)])
            tp, tv = sums[3:5].tolist()                                          # this rank's own training sums (the epoch's
            if dist is not None:                                                 # host synchronisation); one decision for all ranks
                dist.all_reduce(sums)
            vp, vv, vn, gtp, gtv, gnb = sums.tolist()
            if vn < 0.5:
                vp, vv, vn = gtp, gtv, gnb
            vp, vv = vp / max(1.0, vn), vv / max(1.0, vn)
            val_loss = vp + vv
            nb = nb or 1                                                         # (an index_plan without training batches)
            history.append(dict(epoch=e, train_policy=tp / nb, train_value=tv / nb, val_policy=vp, val_value=vv))
            if val_loss < best_val:                                              # :210-219
                best_val, best_epoch = val_loss, e
                best_state = copy.deepcopy(net.state_dict())
            if log:
                log(f"epoch {e}: train {tp / nb:.4f}/{tv / nb:.4f} val {vp:.4f}/{vv:.4f}")
        # keep going only while the best epoch is among the last half group (:231-233)
        if best_epoch < (group + 1) * cfg.epochs_per_group - cfg.epochs_per_group // 2:
            break
    net.eval()
    return best_state, dict(best_epoch=best_epoch, best_val=best_val, epochs=history)


def export_best(net, best_state, game, path):
    """Trace the best-validation weights to `path` (CPU weights, like othello_controller.py:237-239)."""
    snap = copy.deepcopy(net).cpu()
    snap.load_state_dict(best_state)
    return trace_to_file(snap.eval(), path, game)


def export_best_bytes(net, best_state, game) -> bytes:
    """The same traced archive in memory, for the in-process hot swap (no file)."""
    snap = copy.deepcopy(net).cpu()
    snap.load_state_dict(best_state)
    return trace_to_bytes(snap.eval(), game)
