"""The CPU oracle (oracle/sprl_oracle.c) against the golden vectors produced from the reference itself
(tests/golden/gen_golden.py) and, where the prebuilt reference library is present, against live runs.
Integer/byte/float results are compared bit-for-bit (libm math mode = the reference's own libm calls)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from oracle import pyref

SEED = 12345
GAMES = {"othello": po.GAME_OTHELLO, "c4": po.GAME_C4, "go": po.GAME_GO7}


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_g6_rng_streams(golden):
    g = golden("g6_rng.npz")
    assert (po.rng_stream(SEED, 1, 1000) == g["raw_seed12345_stream1"]).all()
    assert (po.rng_stream(987654321987, 77, 64) == g["raw_seed987654321987_stream77"]).all()


def test_g6_distributions(golden):
    g = golden("g6_rng.npz")
    L = po.lib()
    r = po.RNG()
    L.orc_rng_seed(C.byref(r), SEED, 2)
    got = np.array([L.orc_uniform_int(C.byref(r), 0, int(k) - 1) for k in g["uniform_int_k"]], np.int32)
    assert (got == g["uniform_int"]).all()
    fl = np.array([L.orc_uniform_float(C.byref(r)) for _ in range(256)], np.float32)
    assert (bits(fl) == bits(g["uniform_float"])).all()
    for alpha, k in ((0.3, 10), (0.5, 7), (0.2, 50), (1.0, 5), (2.5, 6), (0.3, 1)):
        want = g[f"dirichlet_a{alpha}_k{k}"]
        for row in want:
            v = np.zeros(k, np.float32)
            L.orc_dirichlet(C.byref(r), alpha, k, po.vp(v), po.MATH_LIBM)
            assert (bits(v) == bits(row)).all(), (alpha, k)
    cdf = g["cdf"]
    sc = np.array([L.orc_sample_cdf(C.byref(r), po.vp(cdf), len(cdf)) for _ in range(256)], np.int32)
    assert (sc == g["sample_cdf"]).all()
    assert set(sc.tolist()) <= {2, 3, 5, 6}          # zero-probability entries are never sampled
    assert r.state == int(g["final_state"][0])       # same number of engine draws consumed


@pytest.mark.parametrize("game", ["othello", "c4", "go"])
def test_g1_playouts(golden, game):
    g = golden("g1_playouts.npz")
    for i in range(4):
        seed = int(g[f"{game}_{i}_seed"][0])
        r = po.playout(GAMES[game], seed, 1)
        for k in ("boards", "players", "actions", "terminal"):
            assert (r[k] == g[f"{game}_{i}_{k}"]).all(), (game, i, k)
        assert (bits(r["masks"]) == bits(g[f"{game}_{i}_masks"])).all()
        assert (r["rewards"] == g[f"{game}_{i}_rewards"]).all()
        assert r["terminal"][-1] == 1


def test_g10_c4_known_answer(golden):
    """cpp/tests/test_c4.cpp:5-26 restated: actions {3,3,4,4,2,3,1} end in a win for player ZERO."""
    g = golden("g1_playouts.npz")
    assert int(g["c4_known_answer_ok"][0]) == 1 and g["c4_known_answer_rewards"].tolist() == [1.0, -1.0]
    board = -np.ones(42, np.int8)
    mask = np.ones(7, np.float32)
    player, term, winner = 0, 0, -1
    for a in (3, 3, 4, 4, 2, 3, 1):
        assert term == 0 and winner == -1
        board, mask, term, winner = po.step(po.GAME_C4, board, player, mask, a)
        player = 1 - player
    assert term == 1 and winner == 0
    assert (mask == 0).all()                          # ConnectFourNode.cpp:69-71


@pytest.mark.parametrize("game", ["othello", "c4"])
def test_g2_symmetries(golden, game):
    g = golden("g2_symmetries.npz")
    bo, do, inv = po.symmetrize(GAMES[game], g[f"{game}_board"], g[f"{game}_dist"])
    assert (bo == g[f"{game}_boards_out"]).all()
    assert (bits(do) == bits(g[f"{game}_dists_out"])).all()
    assert (inv == g[f"{game}_inverse"]).all()
    # inverse really inverts
    for s in range(len(inv)):
        b2, _, _ = po.symmetrize(GAMES[game], bo[s], do[s])
        assert (b2[inv[s]] == g[f"{game}_board"]).all()


@pytest.mark.parametrize("key,game,kind,mb,mq,alpha", [
    ("othello_k0_b8q4", "othello", 0, 8, 4, 0.3), ("othello_k0_b1q1", "othello", 0, 1, 1, 0.3),
    ("othello_k1_b8q4", "othello", 1, 8, 4, 0.3), ("othello_k1_b1q1", "othello", 1, 1, 1, 0.3),
    ("c4_k0_b8q4", "c4", 0, 8, 4, 0.5), ("c4_k0_b1q1", "c4", 0, 1, 1, 0.5), ("go_k0_b16q8", "go", 0, 16, 8, 0.2)])
def test_g4_search_trace(golden, key, game, kind, mb, mq, alpha):
    g = golden("g4_search.npz")
    cfg = po.make_config(GAMES[game], 200, max_batch=mb, max_queue=mq, dir_alpha=alpha, eval_kind=kind)
    st, tr, ch = po.search_trace(cfg, 3, SEED, 1)
    assert (tr == g[key + "_trav"]).all() and (ch == g[key + "_chosen"]).all()
    assert (bits(st) == bits(g[key + "_stats"])).all()


def test_g8_q1_wrong_frame_mask(golden):
    """without noise the start-position priors are 0.25 x4 or all zero depending on the symmetry
    drawn for the root evaluation; the oracle reproduces whichever the reference drew."""
    g = golden("g4_search.npz")
    cfg = po.make_config(po.GAME_OTHELLO, 16, add_noise=0)
    st, _, _ = po.search_trace(cfg, 1, SEED, 1)
    assert (bits(st) == bits(g["othello_nonoise_stats"])).all()
    pri = st[0, 2]
    assert set(np.unique(pri).tolist()) <= {0.0, 0.25}


CASES = [
    ("oth_random", po.GAME_OTHELLO, dict(num_traversals=60), 2, 1, True),
    ("oth_heur", po.GAME_OTHELLO, dict(num_traversals=40, eval_kind=po.EVAL_HEURISTIC), 1, 5, True),
    ("c4_random", po.GAME_C4, dict(num_traversals=100), 4, 1, True),
    ("c4_single_stream", po.GAME_C4, dict(num_traversals=100), 3, 9, False),
    ("oth_nosym_b1q1", po.GAME_OTHELLO, dict(num_traversals=30, max_batch=1, max_queue=1, use_sym=0, add_noise=0),
     1, 3, True),
    ("go_random", po.GAME_GO7, dict(num_traversals=120), 3, 1, True),
]


@pytest.mark.parametrize("name,game,kw,ngames,stream,per_game", CASES)
def test_g5_whole_games(golden, name, game, kw, ngames, stream, per_game):
    g = golden("g5_games.npz")
    cfg = po.make_config(game, **kw)
    r = po.selfplay(cfg, ngames, SEED, stream, per_game)
    assert (r["offsets"] == g[name + "_offsets"]).all()
    assert (r["boards"] == g[name + "_boards"]).all()
    assert (r["players"] == g[name + "_players"]).all()
    if name + "_sizes" in g.files:
        assert (r["sizes"] == g[name + "_sizes"]).all()          # 8-ply history: number of valid plies per sample
    assert (bits(r["dists"]) == bits(g[name + "_dists"])).all()
    assert (r["outcomes"] == g[name + "_outcomes"]).all()
    s = r["stats"]
    assert s["games"] == ngames and s["traversals"] >= s["plies"] * cfg.num_traversals


@pytest.mark.parametrize("run,game,ngames,trav,alpha", [("gold", po.GAME_C4, 3, 100, 0.5),
                                                        ("goldoth", po.GAME_OTHELLO, 1, 30, 0.3)])
def test_g5_worker_npy_bytes(golden, tmp_path, run, game, ngames, trav, alpha):
    """Byte-identical .npy streams vs the reference's runWorker + vendored npy writer."""
    g = golden("g5_worker_npy.npz")
    cfg = po.make_config(game, trav, dir_alpha=alpha)
    r = po.selfplay(cfg, ngames, SEED, 1, per_game_stream=False)
    prefix = str(tmp_path / f"{run}_iteration_0")
    assert po.write_records(cfg, prefix, r) == 0
    for part in ("states", "distributions", "outcomes"):
        got = np.frombuffer(open(f"{prefix}_{part}.npy", "rb").read(), np.uint8)
        assert got.shape == g[f"{run}_{part}"].shape and (got == g[f"{run}_{part}"]).all(), part
        arr = np.load(f"{prefix}_{part}.npy")
        assert arr.dtype == np.float32 and arr.shape[0] == len(r["players"])


def test_g7_decode(golden):
    """GridNetwork::evaluate post-processing (exp, wrong-frame mask, sum==0 -> uniform, *1/sum)."""
    import torch
    from sprl_amd.network import GridResNet
    g7, g9 = golden("g7_decode.npz"), golden("g9_network.npz")
    net = GridResNet(8, 8, 65, 1, 1, 8)
    net.load_state_dict({k[4:]: torch.from_numpy(g9[k]) for k in g9.files if k.startswith("sd::")})
    net.eval()

    def fwd(x):
        with torch.no_grad():
            lo, va = net(torch.from_numpy(x))
        return lo.numpy(), va.numpy()

    cb = po.make_forward(fwd, po.GAME_OTHELLO)
    cfg = po.make_config(po.GAME_OTHELLO, 1, eval_kind=po.EVAL_CALLBACK, forward=cb)
    pol, val = po.evaluate(cfg, g7["boards"], g7["players"], g7["masks"])
    np.testing.assert_allclose(pol, g7["policy"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(val, g7["value"], rtol=0, atol=1e-6)
    assert pol[3, 64] == 1.0 and pol[3, :64].sum() == 0.0
    # all-masked-out logits underflow -> uniform fallback
    A = 65
    logits = np.full(A, -200.0, np.float32)
    mask = np.zeros(A, np.float32)
    mask[[3, 9, 40]] = 1.0
    out = np.zeros(A, np.float32)
    po.lib().orc_decode_policy(A, po.vp(logits), po.vp(mask), po.vp(out), po.MATH_LIBM)
    assert out[[3, 9, 40]].tolist() == [np.float32(1.0) / 3] * 3 and out.sum() == pytest.approx(1.0)
    # overflow: exp(+inf-ish) -> inf/inf = nan is what the reference computes; we just must not crash
    logits[:] = 100.0
    po.lib().orc_decode_policy(A, po.vp(logits), po.vp(mask), po.vp(out), po.MATH_LIBM)


def test_g9_network_contract(golden):
    """Our GridResNet == the reference BasicGridNetwork on the captured state_dict (1e-5 abs)."""
    import torch
    from sprl_amd.network import GridResNet
    g9 = golden("g9_network.npz")
    net = GridResNet(8, 8, 65, 1, 1, 8)
    missing = net.load_state_dict({k[4:]: torch.from_numpy(g9[k]) for k in g9.files if k.startswith("sd::")})
    assert not missing.missing_keys and not missing.unexpected_keys
    net.eval()
    with torch.no_grad():
        lo, va = net(torch.from_numpy(g9["input"]))
    np.testing.assert_allclose(lo.numpy(), g9["logits"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(va.numpy(), g9["value"], atol=1e-5, rtol=0)
    assert va.shape == (5, 1)


def test_g9b_baseline_shape_network_contract(golden):
    """BASELINE shape (2 blocks x 64 channels): our GridResNet with the seed-reproducible weights of
    tests/golden/netfill.py == the reference's BasicGridNetwork outputs captured in g9b (CPU fp32, same PyTorch kernels)."""
    import sys
    import torch
    from sprl_amd.network import GridResNet
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import netfill
    g = golden("g9b_baseline_network.npz")
    for gi, gain in enumerate(g["gains"]):
        net = netfill.fill_state_dict(GridResNet(8, 8, 65, 1, 2, 64), int(g["seed"][0]) + gi, float(gain)).eval()
        assert list(net.state_dict().keys()) == [str(k) for k in g["keys"]]
        with torch.no_grad():
            lo, va = net(torch.from_numpy(g["input"]))
        scale = float(np.abs(g[f"logits{gi}"]).max())               # 0.1, 3.3, 16
        np.testing.assert_allclose(lo.numpy(), g[f"logits{gi}"], atol=max(1e-5, 1e-6 * scale), rtol=0)
        np.testing.assert_allclose(va.numpy(), g[f"value{gi}"], atol=1e-5, rtol=0)
        # how far fp32 itself is from the exact (float64) forward: the yardstick for the GPU tolerance - the reference's own
        # CPU fp32 forward is 1.1e-5 off at |logits| = 16, i.e. an ABSOLUTE 1e-5 cannot hold at trained-network scale
        assert np.abs(g[f"logits{gi}"] - g[f"logits_f64_{gi}"]).max() < max(5e-6, 1e-6 * scale)


def test_g9c_go_shape_network_contract(golden):
    """The Go-shape networks of BASELINE configs 4 / 5 (6 blocks x 64 channels, 17 planes; scripts/go_controller.py:44-45): our
    GridResNet with netfill weights and netfill's Go-like inputs == the reference's BasicGridNetwork outputs captured in g9c
    (CPU fp32, same PyTorch kernels) - what the -m gpu test then holds the hand-written path to."""
    import sys
    import torch
    from sprl_amd.network import GridResNet
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import netfill
    g = golden("g9c_go_networks.npz")
    seed = int(g["seed"][0])
    assert int(g["blocks"][0]) == 6 and int(g["channels"][0]) == 64 and int(g["history"][0]) == 8
    for width, actions in ((9, 82), (19, 362)):
        x = torch.from_numpy(netfill.go_like_inputs(int(g[f"n{width}"][0]), width, 8, seed + width))
        assert x.shape[1] == 17 and float(x.max()) == 1.0
        for gi, gain in enumerate(g["gains"]):
            net = netfill.fill_state_dict(GridResNet(width, width, actions, 8, 6, 64), seed + 100 * width + gi, float(gain)).eval()
            with torch.no_grad():
                lo, va = net(x)
            scale = float(np.abs(g[f"logits{width}_{gi}"]).max())
            np.testing.assert_allclose(lo.numpy(), g[f"logits{width}_{gi}"], atol=max(1e-5, 1e-6 * scale), rtol=0)
            np.testing.assert_allclose(va.numpy(), g[f"value{width}_{gi}"], atol=1e-5, rtol=0)
            assert np.abs(g[f"logits{width}_{gi}"] - g[f"logits{width}_f64_{gi}"]).max() < max(1e-5, 1e-6 * scale)


@pytest.mark.skipif(not pyref.available(), reason="prebuilt reference library not present")
@pytest.mark.parametrize("game,kind,trav,mb,mq,alpha,ngames", [
    ("othello", 0, 200, 8, 4, 0.3, 2), ("othello", 1, 100, 8, 4, 0.3, 2), ("othello", 0, 64, 1, 1, 0.3, 1),
    ("c4", 0, 100, 8, 4, 0.5, 6), ("c4", 0, 512, 8, 4, 0.5, 2), ("go", 0, 400, 16, 8, 0.2, 3)])
def test_live_reference_whole_games(game, kind, trav, mb, mq, alpha, ngames):
    """Fresh seeds (not in the fixtures): oracle == reference build, bit for bit."""
    seed = 424242 + trav
    r = pyref.selfplay(game, kind, ngames, trav, mb, mq, 0.25, alpha, seed, 3, True)
    cfg = po.make_config(GAMES[game], trav, max_batch=mb, max_queue=mq, dir_alpha=alpha, eval_kind=kind)
    o = po.selfplay(cfg, ngames, seed, 3, True)
    assert (o["offsets"] == r["offsets"]).all()
    for k in ("boards", "players", "sizes", "outcomes"):
        assert (o[k] == r[k]).all()
    assert (bits(o["dists"]) == bits(r["dists"])).all()


# ---- Go at 9x9 (BASELINE config 4): fixtures from the reference compiled with ONLY GO_BOARD_WIDTH = 9 / GO_KOMI = 7.5 changed
# (oracle/Makefile: ref_go9; tests/golden/gen_golden.py: g_go9) -------------------------------------------------------
def test_go9_playouts_pinned(golden):
    g = golden("g_go9.npz")
    for i in range(4):
        r = po.playout(po.GAME_GO9, int(g[f"playout_{i}_seed"][0]), 1)
        for k in ("boards", "players", "actions", "terminal"):
            assert r[k].shape == g[f"playout_{i}_{k}"].shape and (r[k] == g[f"playout_{i}_{k}"]).all(), (i, k)
        assert (bits(r["masks"]) == bits(g[f"playout_{i}_masks"])).all()
        assert (r["rewards"] == g[f"playout_{i}_rewards"]).all()          # Tromp-Taylor area + komi 7.5
        assert r["terminal"][-1] == 1 and r["boards"].shape[1] == 8 * 81 and r["masks"].shape[1] == 82


def test_go9_search_trace_pinned(golden):
    g = golden("g_go9.npz")
    cfg = po.make_config(po.GAME_GO9, 200, max_batch=16, max_queue=8, dir_alpha=0.2)
    st, tr, ch = po.search_trace(cfg, 3, SEED, 1)
    assert (tr == g["trace_trav"]).all() and (ch == g["trace_chosen"]).all()
    assert (bits(st) == bits(g["trace_stats"])).all()


@pytest.mark.parametrize("name,kw,ngames,stream", [
    ("games", dict(num_traversals=64, max_batch=16, max_queue=8, dir_alpha=0.2), 2, 1),
    ("games_nosym", dict(num_traversals=40, max_batch=4, max_queue=2, dir_alpha=0.2, use_sym=0, add_noise=0), 1, 7)])
def test_go9_whole_games_pinned(golden, name, kw, ngames, stream):
    g = golden("g_go9.npz")
    r = po.selfplay(po.make_config(po.GAME_GO9, **kw), ngames, SEED, stream, True)
    assert (r["offsets"] == g[name + "_offsets"]).all()
    for k in ("boards", "players", "sizes", "outcomes"):
        assert (r[k] == g[f"{name}_{k}"]).all(), k
    assert (bits(r["dists"]) == bits(g[name + "_dists"])).all()


@pytest.mark.skipif(not pyref.available(variant="go9"), reason="prebuilt 9x9 reference library not present")
def test_go9_live_reference():
    """Fresh seeds: oracle at width 9 == the reference compiled at width 9, whole games at the worker's 16/8 batching."""
    assert pyref.lib(variant="go9").ref_go_board_width() == 9 and pyref.lib(variant="go9").ref_go_komi() == 7.5
    for trav, ngames, seed in ((160, 2, 99001), (48, 3, 99002)):
        r = pyref.selfplay("go9", 0, ngames, trav, 16, 8, 0.25, 0.2, seed, 2, True)
        o = po.selfplay(po.make_config(po.GAME_GO9, trav, max_batch=16, max_queue=8, dir_alpha=0.2), ngames, seed, 2, True)
        assert (o["offsets"] == r["offsets"]).all()
        for k in ("boards", "players", "sizes", "outcomes"):
            assert (o[k] == r[k]).all(), k
        assert (bits(o["dists"]) == bits(r["dists"])).all()
    for seed in (5, 6, 7):
        r, o = pyref.playout("go9", seed, 3), po.playout(po.GAME_GO9, seed, 3)
        assert all((o[k] == r[k]).all() for k in ("boards", "players", "actions", "terminal", "rewards"))
        assert (bits(o["masks"]) == bits(r["masks"])).all()


# ---- Go at 19x19 (BASELINE config 5): fixtures from the reference compiled with FOUR lines changed - GO_BOARD_WIDTH = 19,
# GO_KOMI = 7.5 and Coord / LibertyCount int8_t -> int16_t (games/GoNode.hpp:16,20,36-37; oracle/Makefile: ref_go19;
# tests/golden/gen_golden.py: g_go19) -----------------------------------------------------------------------------------
def test_go19_playouts_pinned(golden):
    g = golden("g_go19.npz")
    for i in range(4):
        r = po.playout(po.GAME_GO19, int(g[f"playout_{i}_seed"][0]), 1, 800)
        for k in ("boards", "players", "actions", "terminal"):
            assert r[k].shape == g[f"playout_{i}_{k}"].shape and (r[k] == g[f"playout_{i}_{k}"]).all(), (i, k)
        assert (bits(r["masks"]) == bits(g[f"playout_{i}_masks"])).all()
        assert (r["rewards"] == g[f"playout_{i}_rewards"]).all()          # Tromp-Taylor area + komi 7.5
        assert r["terminal"][-1] == 1 and r["boards"].shape[1] == 8 * 361 and r["masks"].shape[1] == 362
        assert len(r["actions"]) > 300                                    # long enough for captures, ko and superko to occur


def test_go19_search_trace_pinned(golden):
    g = golden("g_go19.npz")
    cfg = po.make_config(po.GAME_GO19, 200, max_batch=16, max_queue=8, dir_alpha=0.2)
    st, tr, ch = po.search_trace(cfg, 3, SEED, 1)
    assert (tr == g["trace_trav"]).all() and (ch == g["trace_chosen"]).all()
    assert (bits(st) == bits(g["trace_stats"])).all()


@pytest.mark.parametrize("name,kw,ngames,stream", [
    ("games", dict(num_traversals=32, max_batch=16, max_queue=8, dir_alpha=0.2), 1, 1),
    ("games_nosym", dict(num_traversals=40, max_batch=4, max_queue=2, dir_alpha=0.2, use_sym=0, add_noise=0), 1, 7)])
def test_go19_whole_games_pinned(golden, name, kw, ngames, stream):
    g = golden("g_go19.npz")
    r = po.selfplay(po.make_config(po.GAME_GO19, **kw), ngames, SEED, stream, True)
    assert (r["offsets"] == g[name + "_offsets"]).all()
    for k in ("boards", "players", "sizes", "outcomes"):
        assert (r[k] == g[f"{name}_{k}"]).all(), k
    assert (bits(r["dists"]) == bits(g[name + "_dists"])).all()


@pytest.mark.skipif(not pyref.available(variant="go19"), reason="prebuilt 19x19 reference library not present")
def test_go19_live_reference():
    """Fresh seeds: oracle at width 19 == the reference compiled at width 19 - whole games at the worker's 16/8 batching and a
    search at the worker's 1600-traversal budget (GoWorker.cpp:23)."""
    L = pyref.lib(variant="go19")
    assert L.ref_go_board_width() == 19 and L.ref_go_komi() == 7.5
    for trav, ngames, seed in ((64, 1, 99101), (24, 2, 99102)):
        r = pyref.selfplay("go19", 0, ngames, trav, 16, 8, 0.25, 0.2, seed, 2, True)
        o = po.selfplay(po.make_config(po.GAME_GO19, trav, max_batch=16, max_queue=8, dir_alpha=0.2), ngames, seed, 2, True)
        assert (o["offsets"] == r["offsets"]).all()
        for k in ("boards", "players", "sizes", "outcomes"):
            assert (o[k] == r[k]).all(), k
        assert (bits(o["dists"]) == bits(r["dists"])).all()
    st, tr, ch = pyref.search_trace("go19", 0, 2, 1600, 16, 8, 0.25, 0.2, 99103, 4)
    st2, tr2, ch2 = po.search_trace(po.make_config(po.GAME_GO19, 1600, max_batch=16, max_queue=8, dir_alpha=0.2), 2, 99103, 4)
    assert (tr == tr2).all() and (ch == ch2).all() and (bits(st) == bits(st2)).all()
    for seed in (5, 6, 7):
        r, o = pyref.playout("go19", seed, 3, 800), po.playout(po.GAME_GO19, seed, 3, 800)
        assert all((o[k] == r[k]).all() for k in ("boards", "players", "actions", "terminal", "rewards"))
        assert (bits(o["masks"]) == bits(r["masks"])).all()


def _match_cfg(game, kind, trav, mb, mq, sym, parent_q):
    # tree options of Evaluate.cpp:94-112 (eps 0.25, alpha 0.1, noise on, default u-weight 1.0)
    return po.make_config(GAMES[game], trav, max_batch=mb, max_queue=mq, dir_eps=0.25, dir_alpha=0.1, u_weight=1.0,
                          use_sym=sym, add_noise=1, eval_kind=kind, init_q=0 if parent_q else 1)


def test_g10_matches(golden):
    """Agent-vs-agent games (Evaluate.cpp through UCTNetworkAgent + playGame): move lists, lengths and winners."""
    g = golden("g10_matches.npz")
    for i, game in enumerate(g["games"]):
        k0, k1, n, trav, mb, mq, s0, p0, s1, p1, seed = (int(v) for v in g["cases"][i])
        w, a, npl = po.match(_match_cfg(str(game), k0, trav, mb, mq, s0, p0), _match_cfg(str(game), k1, trav, mb, mq, s1, p1),
                             n, seed, 1, 160)
        assert (npl == g[f"nplies{i}"]).all()
        assert (a == g[f"actions{i}"]).all()
        assert (w == g[f"winners{i}"]).all()
        for k in range(n):
            assert po.replay_winner(GAMES[str(game)], a[k, :npl[k]]) == w[k]


@pytest.mark.skipif(not pyref.available(), reason="prebuilt reference library not present")
def test_live_reference_matches():
    for game, k0, k1 in (("othello", 1, 0), ("c4", 0, 0)):
        r = pyref.match(game, k0, k1, 3, 72, 8, 4, 1, 0, 1, 1, 31337, 5, 160)
        o = po.match(_match_cfg(game, k0, 72, 8, 4, 1, 0), _match_cfg(game, k1, 72, 8, 4, 1, 1), 3, 31337, 5, 160)
        for x, y in zip(r, o):
            assert (x == y).all()
