"""TEST INFRASTRUCTURE: ctypes binding of oracle/_ref/libsprl_ref*.so — the reference's own sources
compiled in place by oracle/Makefile (see ref_harness.cpp).  Available only where the prebuilt
.so exists (built in the container that has /root/reference; it travels to the GPU box).

Only tests/, tests/golden/gen_golden.py and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_ref", "libsprl_ref.so")
LIB_TORCH = os.path.join(HERE, "_ref", "libsprl_ref_torch.so")
# the same torch-free slice with games/GoNode.hpp's GO_BOARD_WIDTH / GO_KOMI set to 9 / 7.5 (oracle/Makefile: ref_go9);
# its ref_go_* entry points play 9x9 Go, everything else is identical to LIB
LIB_GO9 = os.path.join(HERE, "_ref", "libsprl_ref_go9.so")
# ... and at 19 / 7.5 with Coord / LibertyCount widened to int16_t (games/GoNode.hpp:36-37; oracle/Makefile: ref_go19)
LIB_GO19 = os.path.join(HERE, "_ref", "libsprl_ref_go19.so")
GO_VARIANTS = {"go9": (LIB_GO9, 9), "go19": (LIB_GO19, 19)}

GEOM = {"othello": dict(cells=64, A=65, nsym=8, hist=1), "c4": dict(cells=42, A=7, nsym=2, hist=1),
        "go": dict(cells=49, A=50, nsym=8, hist=8), "go9": dict(cells=81, A=82, nsym=8, hist=8),
        "go19": dict(cells=361, A=362, nsym=8, hist=8)}


def _variant(game):
    return game if game in GO_VARIANTS else None


def _fn(game):
    return "go" if game in GO_VARIANTS else game


def available(torch=False, variant=None):
    if variant in GO_VARIANTS:
        return os.path.exists(GO_VARIANTS[variant][0])
    return os.path.exists(LIB_TORCH if torch else LIB)


_libs = {}


def lib(torch=False, variant=None):
    key = variant or bool(torch)
    if key not in _libs:
        if torch:
            import torch as _t  # noqa: F401  (loads libtorch so the harness' DT_NEEDED entries resolve)
        L = C.CDLL(GO_VARIANTS[variant][0] if variant in GO_VARIANTS else (LIB_TORCH if torch else LIB))
        if variant in GO_VARIANTS:
            assert L.ref_go_board_width() == GO_VARIANTS[variant][1]
        L.ref_seed.argtypes = [C.c_uint64, C.c_int]
        L.ref_uniform_int.argtypes = [C.c_int, C.c_int]
        L.ref_uniform_float.restype = C.c_float
        L.ref_dirichlet.argtypes = [C.c_float, C.c_int, C.c_void_p]
        L.ref_sample_cdf.argtypes = [C.c_void_p, C.c_int]
        L.ref_rng_raw.argtypes = [C.c_int, C.c_void_p]
        L.ref_rng_state.restype = C.c_uint64
        for g in ("othello", "c4"):
            getattr(L, f"ref_{g}_playout").argtypes = [C.c_uint64, C.c_int, C.c_int] + [C.c_void_p] * 6
            getattr(L, f"ref_{g}_symmetrize").argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4
            getattr(L, f"ref_{g}_selfplay").argtypes = [C.c_int] * 5 + [C.c_float] * 2 + [C.c_int] * 2 + \
                [C.c_uint64] + [C.c_int] * 3 + [C.c_void_p] * 5
            getattr(L, f"ref_{g}_search_trace").argtypes = [C.c_int] * 5 + [C.c_float] * 2 + [C.c_int] * 2 + \
                [C.c_uint64, C.c_int] + [C.c_void_p] * 3
        L.ref_go_playout.argtypes = [C.c_uint64, C.c_int, C.c_int] + [C.c_void_p] * 6
        L.ref_go_selfplay.argtypes = [C.c_int] * 5 + [C.c_float] * 2 + [C.c_int] * 2 + [C.c_uint64] + [C.c_int] * 3 + \
            [C.c_void_p] * 6
        L.ref_go_search_trace.argtypes = [C.c_int] * 5 + [C.c_float] * 2 + [C.c_int] * 2 + [C.c_uint64, C.c_int] + \
            [C.c_void_p] * 3
        L.ref_go_komi.restype = C.c_float
        for g in ("othello", "c4"):
            getattr(L, f"ref_{g}_match").argtypes = [C.c_int] * 10 + [C.c_uint64, C.c_int] + [C.c_void_p] * 3 + [C.c_int]
        L.ref_othello_step.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4
        L.ref_othello_evaluate.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5
        L.ref_c4_known_answer.argtypes = [C.c_void_p]
        L.ref_write_npy_f32.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_void_p]
        if torch:
            L.ref_torch_othello_evaluate.argtypes = [C.c_char_p, C.c_int] + [C.c_void_p] * 5
            for g in ("othello", "c4"):
                getattr(L, f"ref_torch_{g}_selfplay").argtypes = [C.c_char_p] + [C.c_int] * 4 + [C.c_float] * 2 + \
                    [C.c_uint64] + [C.c_int] * 3 + [C.c_void_p] * 6
            L.ref_othello_run_worker.argtypes = [C.c_char_p, C.c_char_p] + [C.c_int] * 5 + [C.c_float] * 2 + \
                [C.c_uint64, C.c_int]
            L.ref_c4_run_worker.argtypes = [C.c_char_p, C.c_char_p] + [C.c_int] * 4 + [C.c_float] * 2 + \
                [C.c_uint64, C.c_int]
        _libs[key] = L
    return _libs[key]


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


def selfplay(game, eval_kind, num_games, traversals, max_batch, max_queue, eps, alpha, seed, stream_base=1,
             per_game_stream=True, use_sym=1, add_noise=1, model_path=None):
    g = GEOM[game]
    variant = _variant(game)
    fn_game = _fn(game)
    cap = num_games * (2 * g["cells"] + 8 if fn_game == "go" else 130) * (g["nsym"] if use_sym else 1)
    boards = np.zeros((cap, g["hist"] * g["cells"]), np.int8)
    players = np.zeros(cap, np.int8)
    sizes = np.ones(cap, np.int8)
    dists = np.zeros((cap, g["A"]), np.float32)
    outcomes = np.zeros(cap, np.float32)
    offs = np.zeros(num_games + 1, np.int32)
    evals = None
    if model_path is None:
        extra = (vp(sizes),) if fn_game == "go" else ()
        n = getattr(lib(variant=variant), f"ref_{fn_game}_selfplay")(eval_kind, num_games, traversals, max_batch, max_queue, eps, alpha,
                                                   use_sym, add_noise, seed, stream_base, int(per_game_stream), cap,
                                                   vp(boards), vp(players), vp(dists), vp(outcomes), vp(offs), *extra)
    else:
        ne = np.zeros(1, np.int64)
        n = getattr(lib(True), f"ref_torch_{game}_selfplay")(model_path.encode(), num_games, traversals, max_batch,
                                                             max_queue, eps, alpha, seed, stream_base,
                                                             int(per_game_stream), cap, vp(boards), vp(players),
                                                             vp(dists), vp(outcomes), vp(offs), vp(ne))
        evals = int(ne[0])
    if n < 0:
        raise RuntimeError("reference selfplay: capacity exceeded")
    return dict(boards=boards[:n], players=players[:n], sizes=sizes[:n], dists=dists[:n], outcomes=outcomes[:n],
                offsets=offs, evals=evals)


def search_trace(game, eval_kind, moves, traversals, max_batch, max_queue, eps, alpha, seed, stream=1,
                 use_sym=1, add_noise=1):
    g = GEOM[game]
    stats = np.zeros((moves, 3, g["A"]), np.float32)
    trav = np.zeros(moves, np.int32)
    chosen = np.zeros(moves, np.int16)
    m = getattr(lib(variant=_variant(game)), f"ref_{_fn(game)}_search_trace")(eval_kind, moves, traversals, max_batch, max_queue, eps, alpha,
                                                   use_sym, add_noise, seed, stream, vp(stats), vp(trav), vp(chosen))
    return stats[:m], trav[:m], chosen[:m]


def playout(game, seed, stream=1, max_plies=200):
    g = GEOM[game]
    boards = np.zeros((max_plies, g["hist"] * g["cells"]), np.int8)
    players = np.zeros(max_plies, np.int8)
    actions = np.zeros(max_plies, np.int16)
    masks = np.zeros((max_plies, g["A"]), np.float32)
    terminal = np.zeros(max_plies, np.int8)
    rewards = np.zeros((max_plies, 2), np.float32)
    n = getattr(lib(variant=_variant(game)), f"ref_{_fn(game)}_playout")(seed, stream, max_plies, vp(boards), vp(players), vp(actions),
                                              vp(masks), vp(terminal), vp(rewards))
    return dict(boards=boards[:n], players=players[:n], actions=actions[:n], masks=masks[:n],
                terminal=terminal[:n], rewards=rewards[:n])


def symmetrize(game, board, player, dist):
    g = GEOM[game]
    board = np.ascontiguousarray(board, np.int8)
    dist = np.ascontiguousarray(dist, np.float32)
    bo = np.zeros((g["nsym"], g["cells"]), np.int8)
    do = np.zeros((g["nsym"], g["A"]), np.float32)
    inv = np.zeros(g["nsym"], np.int8)
    getattr(lib(), f"ref_{game}_symmetrize")(vp(board), int(player), vp(dist), vp(bo), vp(do), vp(inv))
    return bo, do, inv


def othello_evaluate(kind, boards, players, masks, model_path=None):
    boards = np.ascontiguousarray(boards, np.int8)
    players = np.ascontiguousarray(players, np.int8)
    masks = np.ascontiguousarray(masks, np.float32)
    n = len(players)
    pol = np.zeros((n, 65), np.float32)
    val = np.zeros(n, np.float32)
    if model_path is None:
        lib().ref_othello_evaluate(kind, n, vp(boards), vp(players), vp(masks), vp(pol), vp(val))
    else:
        lib(True).ref_torch_othello_evaluate(model_path.encode(), n, vp(boards), vp(players), vp(masks), vp(pol),
                                             vp(val))
    return pol, val


def match(game, kind0, kind1, num_games, traversals, max_batch, max_queue, sym0, pq0, sym1, pq1, seed, stream_base=1,
          max_plies=256):
    """Evaluate.cpp-style matches through the reference's UCTNetworkAgent + playGame."""
    winners = np.zeros(num_games, np.int8)
    actions = np.full((num_games, max_plies), -1, np.int16)
    nplies = np.zeros(num_games, np.int32)
    rc = getattr(lib(), f"ref_{game}_match")(kind0, kind1, num_games, traversals, max_batch, max_queue, sym0, pq0, sym1,
                                            pq1, seed, stream_base, vp(winners), vp(actions), vp(nplies), max_plies)
    if rc != 0:
        raise RuntimeError("reference match: replay disagreed with playGame")
    return winners, actions, nplies
