"""TEST INFRASTRUCTURE: ctypes binding of oracle/liboracle.so (the plain-C restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")

GAME_OTHELLO, GAME_C4, GAME_GO7, GAME_GO9, GAME_GO19 = 0, 1, 2, 3, 4
EVAL_RANDOM, EVAL_HEURISTIC, EVAL_CALLBACK = 0, 1, 2
MATH_LIBM, MATH_PORTABLE = 0, 1
MASK_REFERENCE, MASK_SYMMETRISED = 0, 1

FORWARD_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float))


class Config(C.Structure):
    _fields_ = [
        ("game", C.c_int32), ("num_traversals", C.c_int32), ("max_batch", C.c_int32), ("max_queue", C.c_int32),
        ("dir_eps", C.c_float), ("dir_alpha", C.c_float), ("u_weight", C.c_float),
        ("early_cutoff", C.c_int32), ("early_exp", C.c_float), ("rest_exp", C.c_float),
        ("use_sym", C.c_int32), ("add_noise", C.c_int32), ("eval_kind", C.c_int32), ("math_mode", C.c_int32),
        ("mask_frame", C.c_int32), ("init_q", C.c_int32), ("resign_threshold", C.c_float), ("resign_min_ply", C.c_int32),
        ("forward", FORWARD_FN), ("forward_user", C.c_void_p),
    ]


class Stats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "games", "plies", "traversals", "expansions", "nn_evals", "terminal_hits", "gray_hits", "dup_hits",
        "levels", "nodes_created", "max_live_nodes")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class RNG(C.Structure):
    _fields_ = [("state", C.c_uint64), ("inc", C.c_uint64)]


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        L.orc_rng_seed.argtypes = [C.POINTER(RNG), C.c_uint64, C.c_int]
        L.orc_rng_next.argtypes = [C.POINTER(RNG)]
        L.orc_rng_next.restype = C.c_uint32
        L.orc_uniform_int.argtypes = [C.POINTER(RNG), C.c_int, C.c_int]
        L.orc_uniform_float.argtypes = [C.POINTER(RNG)]
        L.orc_uniform_float.restype = C.c_float
        L.orc_dirichlet.argtypes = [C.POINTER(RNG), C.c_float, C.c_int, C.c_void_p, C.c_int]
        L.orc_sample_cdf.argtypes = [C.POINTER(RNG), C.c_void_p, C.c_int]
        L.orc_playout.argtypes = [C.c_int, C.c_uint64, C.c_int, C.c_int] + [C.c_void_p] * 6
        L.orc_step.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                               C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_symmetrize_board.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_symmetrize_dist.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_evaluate.argtypes = [C.POINTER(Config), C.c_int] + [C.c_void_p] * 6
        L.orc_encode_planes.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_decode_policy.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_search_trace.argtypes = [C.POINTER(Config), C.c_int, C.c_uint64, C.c_int] + [C.c_void_p] * 3
        L.orc_selfplay.argtypes = [C.POINTER(Config), C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int] + \
            [C.c_void_p] * 6 + [C.POINTER(Stats)]
        L.orc_match.argtypes = [C.POINTER(Config), C.POINTER(Config), C.c_int, C.c_uint64, C.c_int] + [C.c_void_p] * 3 + [C.c_int]
        L.orc_write_npy_f32.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_write_records.argtypes = [C.POINTER(Config), C.c_char_p, C.c_int] + [C.c_void_p] * 5
        _lib = L
    return _lib


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


GEOM = {GAME_OTHELLO: dict(rows=8, cols=8, cells=64, A=65, nsym=8, hist=1),
        GAME_C4: dict(rows=6, cols=7, cells=42, A=7, nsym=2, hist=1),
        GAME_GO7: dict(rows=7, cols=7, cells=49, A=50, nsym=8, hist=8),
        GAME_GO9: dict(rows=9, cols=9, cells=81, A=82, nsym=8, hist=8),
        GAME_GO19: dict(rows=19, cols=19, cells=361, A=362, nsym=8, hist=8)}

# the reference workers' constants: OTHWorker.cpp:24-28, C4Worker.cpp:23-27, constants.hpp:6-10
DEFAULTS = {
    GAME_OTHELLO: dict(max_batch=8, max_queue=4, dir_eps=0.25, dir_alpha=0.3),
    GAME_C4: dict(max_batch=8, max_queue=4, dir_eps=0.25, dir_alpha=0.5),
    GAME_GO7: dict(max_batch=16, max_queue=8, dir_eps=0.25, dir_alpha=0.2),     # GoWorker.cpp:23-27
    GAME_GO9: dict(max_batch=16, max_queue=8, dir_eps=0.25, dir_alpha=0.2),
    GAME_GO19: dict(max_batch=16, max_queue=8, dir_eps=0.25, dir_alpha=0.2),
}


def make_config(game, num_traversals, *, max_batch=None, max_queue=None, dir_eps=None, dir_alpha=None,
                u_weight=1.1, early_cutoff=15, early_exp=0.98, rest_exp=10.0, use_sym=1, add_noise=1,
                eval_kind=EVAL_RANDOM, math_mode=MATH_LIBM, mask_frame=MASK_REFERENCE, forward=None, init_q=0,
                resign_threshold=0.0, resign_min_ply=0):
    d = DEFAULTS[game]
    cfg = Config()
    cfg.game = game
    cfg.num_traversals = num_traversals
    cfg.max_batch = d["max_batch"] if max_batch is None else max_batch
    cfg.max_queue = d["max_queue"] if max_queue is None else max_queue
    cfg.dir_eps = d["dir_eps"] if dir_eps is None else dir_eps
    cfg.dir_alpha = d["dir_alpha"] if dir_alpha is None else dir_alpha
    cfg.u_weight = u_weight
    cfg.early_cutoff = early_cutoff
    cfg.early_exp = early_exp
    cfg.rest_exp = rest_exp
    cfg.use_sym = use_sym
    cfg.add_noise = add_noise
    cfg.eval_kind = eval_kind
    cfg.math_mode = math_mode
    cfg.mask_frame = mask_frame
    cfg.init_q = init_q
    cfg.resign_threshold = resign_threshold
    cfg.resign_min_ply = resign_min_ply
    if forward is not None:
        cfg.forward = forward
    return cfg


def make_forward(fn, game):
    """Wrap a python callable planes[n,P,R,C] -> (logits[n,A], values[n]) as an orc_forward_fn."""
    g = GEOM[game]

    def _cb(user, n, planes, logits, values):
        x = np.ctypeslib.as_array(planes, shape=(n, 2 * g["hist"] + 1, g["rows"], g["cols"]))
        lo, va = fn(x.copy())
        np.ctypeslib.as_array(logits, shape=(n, g["A"]))[:] = np.asarray(lo, np.float32).reshape(n, g["A"])
        np.ctypeslib.as_array(values, shape=(n,))[:] = np.asarray(va, np.float32).reshape(n)

    return FORWARD_FN(_cb)


def selfplay(cfg, num_games, seed, stream_base=1, per_game_stream=True, cap=None):
    g = GEOM[cfg.game]
    if cap is None:
        cap = num_games * (2 * g["cells"] + 4) * (g["nsym"] if cfg.use_sym else 1)
    boards = np.zeros((cap, g["hist"] * g["cells"]), np.int8)
    players = np.zeros(cap, np.int8)
    sizes = np.zeros(cap, np.int8)
    dists = np.zeros((cap, g["A"]), np.float32)
    outcomes = np.zeros(cap, np.float32)
    offs = np.zeros(num_games + 1, np.int32)
    st = Stats()
    n = lib().orc_selfplay(C.byref(cfg), num_games, seed, stream_base, int(per_game_stream), cap,
                           vp(boards), vp(players), vp(sizes), vp(dists), vp(outcomes), vp(offs), C.byref(st))
    if n < 0:
        raise RuntimeError("oracle selfplay: capacity exceeded")
    return dict(boards=boards[:n], players=players[:n], sizes=sizes[:n], dists=dists[:n], outcomes=outcomes[:n],
                offsets=offs, stats=st.as_dict())


def search_trace(cfg, moves, seed, stream=1):
    g = GEOM[cfg.game]
    stats = np.zeros((moves, 3, g["A"]), np.float32)
    trav = np.zeros(moves, np.int32)
    chosen = np.zeros(moves, np.int16)
    m = lib().orc_search_trace(C.byref(cfg), moves, seed, stream, vp(stats), vp(trav), vp(chosen))
    return stats[:m], trav[:m], chosen[:m]


def playout(game, seed, stream=1, max_plies=200):
    g = GEOM[game]
    boards = np.zeros((max_plies, g["hist"] * g["cells"]), np.int8)
    players = np.zeros(max_plies, np.int8)
    actions = np.zeros(max_plies, np.int16)
    masks = np.zeros((max_plies, g["A"]), np.float32)
    terminal = np.zeros(max_plies, np.int8)
    rewards = np.zeros((max_plies, 2), np.float32)
    n = lib().orc_playout(game, seed, stream, max_plies, vp(boards), vp(players), vp(actions), vp(masks),
                          vp(terminal), vp(rewards))
    return dict(boards=boards[:n], players=players[:n], actions=actions[:n], masks=masks[:n],
                terminal=terminal[:n], rewards=rewards[:n])


def step(game, board, player, mask, action):
    g = GEOM[game]
    board = np.ascontiguousarray(board, np.int8)
    mask = np.ascontiguousarray(mask, np.float32)
    nb = np.zeros(g["cells"], np.int8)
    nm = np.zeros(g["A"], np.float32)
    t, w = C.c_int(), C.c_int()
    lib().orc_step(game, vp(board), int(player), vp(mask), int(action), vp(nb), vp(nm), C.byref(t), C.byref(w))
    return nb, nm, t.value, w.value


def replay_winner(game, actions):
    """Replay a move list from the start position with the oracle's rules (Othello / Connect Four: no history needed).
    Every move must be legal and the list must end exactly at a terminal position; returns the winner colour (-1 = draw)."""
    g = GEOM[game]
    board = np.zeros(g["cells"], np.int8)
    mask = np.zeros(g["A"], np.float32)
    pl = C.c_int()
    lib().orc_start(game, vp(board), C.byref(pl), vp(mask))
    player, term, win = pl.value, 0, -1
    for a in actions:
        assert not term, "moves after the end of the game"
        assert mask[int(a)] > 0, f"illegal move {int(a)}"
        board, mask, term, win = step(game, board, player, mask, int(a))
        player = 1 - player
    assert term, "the move list stops before the game ends"
    return win


def symmetrize(game, board, dist):
    g = GEOM[game]
    board = np.ascontiguousarray(board, np.int8)
    dist = np.ascontiguousarray(dist, np.float32)
    bo = np.zeros((g["nsym"], g["cells"]), np.int8)
    do = np.zeros((g["nsym"], g["A"]), np.float32)
    inv = np.zeros(g["nsym"], np.int8)
    for s in range(g["nsym"]):
        lib().orc_symmetrize_board(game, s, vp(board), vp(bo[s]))
        lib().orc_symmetrize_dist(game, s, vp(dist), vp(do[s]))
        inv[s] = lib().orc_inverse_symmetry(game, s)
    return bo, do, inv


def evaluate(cfg, boards, players, masks):
    g = GEOM[cfg.game]
    boards = np.ascontiguousarray(boards, np.int8)
    players = np.ascontiguousarray(players, np.int8)
    masks = np.ascontiguousarray(masks, np.float32)
    n = len(players)
    pol = np.zeros((n, g["A"]), np.float32)
    val = np.zeros(n, np.float32)
    lib().orc_evaluate(C.byref(cfg), n, vp(boards), vp(players), None, vp(masks), vp(pol), vp(val))
    return pol, val


def rng_stream(seed, stream, n):
    r = RNG()
    lib().orc_rng_seed(C.byref(r), seed, stream)
    return np.array([lib().orc_rng_next(C.byref(r)) for _ in range(n)], np.uint32)


def write_records(cfg, prefix, res):
    return lib().orc_write_records(C.byref(cfg), prefix.encode(), len(res["players"]), vp(res["boards"]),
                                   vp(res["players"]), vp(res["sizes"]), vp(res["dists"]), vp(res["outcomes"]))


def match(cfg0, cfg1, num_games, seed, stream_base=1, max_plies=256):
    """Net-vs-net games (Evaluate.cpp): returns winners[int8], actions[num_games, max_plies], nplies."""
    winners = np.zeros(num_games, np.int8)
    actions = np.full((num_games, max_plies), -1, np.int16)
    nplies = np.zeros(num_games, np.int32)
    lib().orc_match(C.byref(cfg0), C.byref(cfg1), num_games, seed, stream_base, vp(winners), vp(actions), vp(nplies),
                    max_plies)
    return winners, actions, nplies
