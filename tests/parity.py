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


def check_match(lib, game, agents, num_games, seed=11, forwards=(None, None), oracle_forwards=(None, None), **cfg_kw):
    """agents: two dicts(model=..., use_symmetry=..., parent_q=...).  Device match == oracle match, move for move."""
    cfg = match_config(lib, game, seed=seed, **cfg_kw)
    specs = []
    for a, f in zip(agents, forwards):
        s = dict(a)
        if f is not None:
            s["forward"] = f
        specs.append(s)
    winners, actions, nplies = E.play_match(cfg, specs[0], specs[1], num_games, max_plies=160, lib=lib)
    ocfg = []
    for a, of in zip(agents, oracle_forwards):
        kind = po.EVAL_CALLBACK if of is not None else MATCH_EVAL[a["model"]]
        ocfg.append(po.make_config(OGAME[game], cfg.num_traversals, max_batch=cfg.max_batch, max_queue=cfg.max_queue,
                                   dir_eps=cfg.dir_eps, dir_alpha=cfg.dir_alpha, u_weight=cfg.u_weight,
                                   use_sym=1 if a.get("use_symmetry", True) else 0, add_noise=cfg.add_noise,
                                   eval_kind=kind, math_mode=po.MATH_PORTABLE, mask_frame=cfg.mask_frame, forward=of,
                                   init_q=0 if a.get("parent_q", True) else 1))
    ow, oa, on = po.match(ocfg[0], ocfg[1], num_games, seed, cfg.stream_base, max_plies=160)
    assert (nplies == on).all(), (nplies, on)
    assert (actions == oa).all()
    assert (winners == ow).all()
    return winners, actions, nplies
