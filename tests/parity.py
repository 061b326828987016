"""Shared parity checks: an engine library (gfx950 build on the GPU box, SIMT-emulator build on CPU)
against the CPU oracle in portable-math mode.  Bit-exact: boards, movers, pdf bits, outcomes, counters."""
import numpy as np

from oracle import pyoracle as po
from sprl_amd import engine as E

OGAME = {"othello": po.GAME_OTHELLO, "c4": po.GAME_C4, "go": po.GAME_GO7, "go7_wide": po.GAME_GO7, "go9": po.GAME_GO9, "go19": po.GAME_GO19}


def oracle_config(game, cfg, kind, forward=None):
    return po.make_config(OGAME[game], cfg.num_traversals, max_batch=cfg.max_batch, max_queue=cfg.max_queue,
                          dir_eps=cfg.dir_eps, dir_alpha=cfg.dir_alpha, u_weight=cfg.u_weight,
                          early_cutoff=cfg.early_cutoff, early_exp=cfg.early_exp, rest_exp=cfg.rest_exp,
                          use_sym=cfg.use_symmetry, add_noise=cfg.add_noise, eval_kind=kind,
                          math_mode=po.MATH_PORTABLE, mask_frame=cfg.mask_frame, forward=forward,
                          resign_threshold=cfg.resign_threshold, resign_min_ply=cfg.resign_min_ply)


def run_engine(lib, game, num_games, model="random", forward=None, **cfg_kw):
    cfg = E.default_config(game, lib, **cfg_kw)
    eng = E.Engine(cfg, lib)
    if forward is not None:
        eng.set_forward(forward)
    else:
        eng.set_model(model)
    rec = eng.run(num_games)
    st = eng.stats()
    eng.close()
    return cfg, rec, st


def assert_same_games(rec, ora):
    boards, players = rec.expand_boards()
    _, dists, outcomes = rec.expand()
    assert boards.shape == ora["boards"].shape, (boards.shape, ora["boards"].shape)
    assert (boards == ora["boards"]).all()
    assert (players == ora["players"]).all()
    assert (dists.view(np.uint32) == ora["dists"].view(np.uint32)).all()
    assert (outcomes == ora["outcomes"]).all()
    nsym = rec.nsym if rec.use_symmetry else 1
    assert (rec.ply_offset * nsym == ora["offsets"]).all()


def assert_same_counters(st, ost):
    for k in ("games", "plies", "traversals", "levels", "expansions", "nn_evals", "terminal_hits", "gray_hits",
              "dup_hits", "nodes_created"):
        assert st[k] == ost[k], (k, st[k], ost[k])


def check_case(lib, game, num_games, model="random", seed=7, **cfg_kw):
    kind = {"random": po.EVAL_RANDOM, "heuristic": po.EVAL_HEURISTIC}[model]
    cfg, rec, st = run_engine(lib, game, num_games, model, seed=seed, **cfg_kw)
    ora = po.selfplay(oracle_config(game, cfg, kind), num_games, seed, cfg.stream_base, True)
    assert_same_games(rec, ora)
    assert_same_counters(st, ora["stats"])
    return rec, st


def toy_forward_numpy(planes, A):
    """A deterministic stand-in network evaluated sample by sample in float64 (so that batch composition
    cannot change a bit): logits from fixed pseudo-random projections of the planes, value = tanh."""
    n = planes.shape[0]
    x = planes.reshape(n, -1).astype(np.float64)
    rng = np.random.default_rng(1234)
    w = rng.standard_normal((x.shape[1], A)) * 0.35
    v = rng.standard_normal(x.shape[1]) * 0.2
    logits = np.stack([x[i] @ w for i in range(n)]).astype(np.float32)
    value = np.tanh(np.array([x[i] @ v for i in range(n)])).astype(np.float32)
    return logits, value


MATCH_EVAL = {"random": po.EVAL_RANDOM, "heuristic": po.EVAL_HEURISTIC}


def match_config(lib, game, **kw):
    """Tree options of the reference's Evaluate.cpp:94-112 (eps 0.25, alpha 0.1, noise on; u-weight 1.0 is the
    UCTTree default there)."""
    base = dict(dir_eps=0.25, dir_alpha=0.1, u_weight=1.0, add_noise=1)
    base.update(kw)
    return E.default_config(game, lib, **base)


def check_match(lib, game, agents, num_games, seed=11, forwards=(None, None), oracle_forwards=(None, None), max_plies=160, **cfg_kw):
    """agents: two dicts(model=..., use_symmetry=..., parent_q=...).  Device match == oracle match, move for move."""
    cfg = match_config(lib, game, seed=seed, **cfg_kw)
    specs = []
    for a, f in zip(agents, forwards):
        s = dict(a)
        if f is not None:
            s["forward"] = f
        specs.append(s)
    winners, actions, nplies = E.play_match(cfg, specs[0], specs[1], num_games, max_plies=max_plies, lib=lib)
    ocfg = []
    for a, of in zip(agents, oracle_forwards):
        kind = po.EVAL_CALLBACK if of is not None else MATCH_EVAL[a["model"]]
        ocfg.append(po.make_config(OGAME[game], cfg.num_traversals, max_batch=cfg.max_batch, max_queue=cfg.max_queue,
                                   dir_eps=cfg.dir_eps, dir_alpha=cfg.dir_alpha, u_weight=cfg.u_weight,
                                   use_sym=1 if a.get("use_symmetry", True) else 0, add_noise=cfg.add_noise,
                                   eval_kind=kind, math_mode=po.MATH_PORTABLE, mask_frame=cfg.mask_frame, forward=of,
                                   init_q=0 if a.get("parent_q", True) else 1))
    ow, oa, on = po.match(ocfg[0], ocfg[1], num_games, seed, cfg.stream_base, max_plies=max_plies)
    assert (nplies == on).all(), (nplies, on)
    assert (actions == oa).all()
    assert (winners == ow).all()
    return winners, actions, nplies


# ---- distributional parity of the CNN path (SURVEY section 7 hard-part iii) -----------------------------------------------
def game_statistics(ply_offset, pdfs, outcome_player0, evals=None):
    """Per-game (plies, outcome for Player ZERO, mean entropy of the recorded root-visit pdfs)."""
    lengths = np.diff(ply_offset)
    p = np.asarray(pdfs, np.float64)
    h = -(p * np.log(np.where(p > 0, p, 1.0))).sum(1)
    ent = np.array([h[ply_offset[g]:ply_offset[g + 1]].mean() for g in range(len(lengths))])
    return dict(plies=lengths.astype(np.float64), outcome=np.asarray(outcome_player0, np.float64), entropy=ent,
                evals=None if evals is None else np.asarray(evals, np.float64))


def reference_cnn_games(args):
    """Pool worker (spawn context, never touches the GPU): `games` Othello games of the REFERENCE's selfPlay + GridNetwork on
    LibTorch-CPU (oracle/_ref/libsprl_ref_torch.so, networks/GridNetwork.hpp:62-145, selfplay/SelfPlay.hpp:51-192) with the traced
    model; returns per game (plies, outcome of Player ZERO, network evaluations, mean pdf entropy)."""
    import os
    model, games, trav, seed, stream = args
    os.environ["OMP_NUM_THREADS"] = "1"
    os.environ["MKL_NUM_THREADS"] = "1"
    import torch
    torch.set_num_threads(1)
    from oracle import pyref
    out = []
    for g in range(games):
        r = pyref.selfplay("othello", 0, 1, trav, 8, 4, 0.25, 0.3, seed, stream + g, True, model_path=model)
        n = len(r["players"]) // 8
        off = np.array([0, n])
        first_mover = int(r["players"][0])                   # outcomes are rewards[mover] (SelfPlay.hpp:154-163)
        z0 = float(r["outcomes"][0]) * (1.0 if first_mover == 0 else -1.0)
        st = game_statistics(off, r["dists"][::8], [z0])
        out.append((float(n), z0, float(r["evals"]), float(st["entropy"][0])))
    return out


# ---- f-2: the reference controller's training call, replayed (tests/golden/gen_golden.py: g_trainer) ----------------------------
def replay_trainer_fixture(g, device, atol):
    """Feeds the batches the REFERENCE's train_network drew (scripts/othello_controller.py:128-241) through sprl_amd.trainer on
    `device`; checks best epoch, number of epochs, and the outputs of the best / live network on the probe batch.
    Returns the largest absolute deviation from the reference's outputs."""
    import os
    import sys
    import torch
    from sprl_amd import trainer as T
    from sprl_amd.network import GridResNet
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import netfill
    seed, n = int(g["seed"][0]), int(g["n"][0])
    states = torch.from_numpy(netfill.othello_like_inputs(n, seed)).to(device)
    tensors = (states, torch.from_numpy(g["dists"]).to(device), torch.from_numpy(g["outcomes"]).to(device),
               torch.from_numpy(g["stamps"]).to(device))
    ntr, nva = (int(v) for v in g["batches_per_epoch"])
    batches = [torch.from_numpy(g[f"batch{i}"].astype(np.int64)).to(device) for i in range(int(g["epochs"][0]) * (ntr + nva))]

    def plan(epoch):
        k = epoch * (ntr + nva)
        return batches[k:k + ntr], batches[k + ntr:k + ntr + nva]

    net = netfill.fill_state_dict(GridResNet(8, 8, 65, 1, 1, 8), seed).to(device)
    cfg = T.TrainerConfig(batch_size=int(g["batch_size"][0]), max_groups=int(g["max_groups"][0]),
                          epochs_per_group=int(g["epochs_per_group"][0]))
    best, hist = T.train_network(net, float(g["lr"][0]), tensors, cfg, index_plan=plan)
    assert hist["best_epoch"] == int(g["best_epoch"][0])
    assert len(hist["epochs"]) == int(g["epochs"][0])             # the "best epoch is recent" continuation rule (:231-233)
    with torch.no_grad():
        flo, fva = net.eval()(states[:16])
        snap = GridResNet(8, 8, 65, 1, 1, 8).to(device)
        snap.load_state_dict(best)
        blo, bva = snap.eval()(states[:16])
    dev = 0.0
    for ours, key in ((flo, "final_logits"), (fva, "final_value"), (blo, "best_logits"), (bva, "best_value")):
        dev = max(dev, float(np.abs(ours.cpu().numpy() - g[key]).max()))
        np.testing.assert_allclose(ours.cpu().numpy(), g[key], atol=atol, rtol=0)
    np.testing.assert_allclose(g["traced_logits"], g["best_logits"], atol=1e-6, rtol=0)
    return dev
