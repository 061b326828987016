"""Kernel-logic parity WITHOUT a GPU: the product kernel source (sprl_amd/csrc/step_kernel.h) compiled for
the CPU SIMT emulator (tests/emu) must reproduce the oracle's self-play games bit for bit.  The emulator is
test infrastructure only — the GPU parity tests proper are in test_gpu_parity.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from sprl_amd import engine as E
import parity

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    return E.load_library(os.path.join(EMU_DIR, "libsprl_emu.so"))


def test_c4_random(emu):
    parity.check_case(emu, "c4", 3, concurrent_games=2, num_traversals=40)


def test_othello_random(emu):
    parity.check_case(emu, "othello", 2, concurrent_games=2, num_traversals=30)


def test_othello_heuristic(emu):
    parity.check_case(emu, "othello", 2, model="heuristic", concurrent_games=2, num_traversals=30)


def test_othello_compaction_with_tiny_arena(emu):
    """Bump allocation + Cheney compaction alone (node recycling switched off): the fallback path of every game and the
    allocator of the multi-strip kernel."""
    rec, st = parity.check_case(emu, "othello", 2, concurrent_games=2, num_traversals=40, node_cap=120,
                                spare_arenas=2, no_recycle=1)
    assert st["compactions"] > 0 and st["max_nodes_in_arena"] <= 120 and st["nodes_recycled"] == 0


def test_node_recycling_keeps_arena_small(emu):
    """The nodes of pruned siblings are reused (UCTNode::pruneChildrenExcept frees them in the reference): whole games stay
    bit-identical to the oracle while the arena's high-water mark follows the per-move budget, not the game length - and the
    tiny arena that forced compactions above now needs none."""
    rec, st = parity.check_case(emu, "othello", 2, concurrent_games=2, num_traversals=40, node_cap=120, spare_arenas=2)
    assert st["compactions"] == 0 and st["nodes_recycled"] > 0.5 * st["nodes_created"] and st["max_nodes_in_arena"] <= 120
    # same games without recycling create the same number of nodes but need several times the space
    _, _, st0 = parity.run_engine(emu, "othello", 2, concurrent_games=2, num_traversals=40, no_recycle=1, seed=7)
    assert st0["nodes_created"] == st["nodes_created"] and st0["max_nodes_in_arena"] > 4 * st["max_nodes_in_arena"]
    for game, kw in (("c4", dict(num_traversals=60, node_cap=200)), ("go", dict(num_traversals=48, node_cap=260))):
        rec, st = parity.check_case(emu, game, 2, concurrent_games=2, seed=17, **kw)
        assert st["nodes_recycled"] > 0 and st["compactions"] == 0


def test_child_indices_beyond_16_bits(emu):
    """a move's tree may exceed 65535 nodes (the reference worker's iteration-0 budget is 131072
    traversals/move at batch 1, OTHWorker.cpp:17-20).  Arenas above 65535 nodes switch to 24-bit child indices (u16 row +
    u8 row).  The allocator is started at node 65400 (test hook), so a short game's ids run across the 16-bit boundary -
    with bump allocation, with recycling and through a compaction."""
    rec, st = parity.check_case(emu, "othello", 1, concurrent_games=1, num_traversals=40, node_cap=70000, no_recycle=1, seed=3,
                                alloc_base=65400)
    assert st["max_nodes_in_arena"] > 65535 + 500 and st["compactions"] == 0
    rec, st = parity.check_case(emu, "othello", 1, concurrent_games=1, num_traversals=40, node_cap=70000, seed=3, alloc_base=65520)
    assert st["max_nodes_in_arena"] > 65535 and st["nodes_recycled"] > 0
    rec, st = parity.check_case(emu, "go", 1, concurrent_games=1, num_traversals=60, node_cap=65400 + 700, spare_arenas=2,
                                no_recycle=1, seed=5, alloc_base=65400)
    assert st["compactions"] > 0


def test_othello_no_symmetry_no_noise_batch1(emu):
    parity.check_case(emu, "othello", 1, concurrent_games=1, num_traversals=25, use_symmetry=0, add_noise=0,
                      max_batch=1, max_queue=1)


def test_othello_symmetrised_mask_option(emu):
    parity.check_case(emu, "othello", 1, concurrent_games=1, num_traversals=30, mask_frame=E.MASK_SYMMETRISED)


def test_go7_random(emu):
    """Go 7x7 as the reference compiles it: pass competes in the arg-max, captures, positional superko against the
    ancestor positions, double-pass / depth-cap endings, Tromp-Taylor + komi, 8-ply history states."""
    rec, st = parity.check_case(emu, "go", 3, concurrent_games=2, num_traversals=48)
    assert rec.history == 8 and rec.planes == 17 and rec.actions == 50


def test_go7_compaction_and_batch1(emu):
    parity.check_case(emu, "go", 2, concurrent_games=2, num_traversals=40, node_cap=150, spare_arenas=2, seed=31)
    parity.check_case(emu, "go", 1, concurrent_games=1, num_traversals=20, max_batch=1, max_queue=1, use_symmetry=0,
                      add_noise=0, seed=5)


def test_go7_network_path_toy_forward(emu):
    """17-plane history encoding (symmetrised per ply) -> forward -> decode, against the oracle's GridNetwork restatement."""
    A = 50

    def engine_forward(planes_ptr, batch, logits_ptr, value_ptr):
        planes = np.ctypeslib.as_array(C.cast(planes_ptr, C.POINTER(C.c_float)), shape=(batch, 17, 7, 7))
        lo, va = parity.toy_forward_numpy(planes, A)
        np.ctypeslib.as_array(C.cast(logits_ptr, C.POINTER(C.c_float)), shape=(batch, A))[:] = lo
        np.ctypeslib.as_array(C.cast(value_ptr, C.POINTER(C.c_float)), shape=(batch,))[:] = va
        return 0

    cfg, rec, st = parity.run_engine(emu, "go", 2, forward=engine_forward, concurrent_games=2, num_traversals=24, seed=5)
    cb = po.make_forward(lambda x: parity.toy_forward_numpy(x, A), po.GAME_GO7)
    ora = po.selfplay(parity.oracle_config("go", cfg, po.EVAL_CALLBACK, forward=cb), 2, 5, 1, True)
    parity.assert_same_games(rec, ora)
    parity.assert_same_counters(st, ora["stats"])
    states, _, _ = rec.expand()
    assert states.shape[1:] == (17, 7, 7)


def test_wide_kernel_at_reference_size(emu):
    """The multi-strip kernel (step_kernel_wide.h) on Go 7x7, where the oracle is pinned to the reference build."""
    parity.check_case(emu, "go7_wide", 2, concurrent_games=2, num_traversals=40)


def test_go9_two_strip_rows(emu):
    """Go 9x9 (BASELINE config 4 geometry): 81 points + pass = rows of two wavefront strips, 2-word bit boards."""
    rec, st = parity.check_case(emu, "go9", 2, concurrent_games=2, num_traversals=40)
    assert rec.cells == 81 and rec.actions == 82 and rec.planes == 17
    rec, st = parity.check_case(emu, "go9", 2, concurrent_games=2, num_traversals=60, node_cap=200, spare_arenas=2, seed=9,
                                no_recycle=1)
    assert st["compactions"] > 0 and st["nodes_recycled"] == 0
    # the same games with node recycling (the default): the 200-node arenas are never compacted
    rec, st = parity.check_case(emu, "go9", 2, concurrent_games=2, num_traversals=60, node_cap=200, spare_arenas=0, seed=9)
    assert st["compactions"] == 0 and st["nodes_recycled"] > 0.5 * st["nodes_created"] and st["max_nodes_in_arena"] <= 200


def test_go19_six_strip_rows(emu):
    """Go 19x19 (BASELINE config 5 geometry): 361 points + pass, rows of six strips, 6-word bit boards."""
    rec, st = parity.check_case(emu, "go19", 1, concurrent_games=1, num_traversals=12, seed=4)
    assert rec.cells == 361 and rec.actions == 362 and st["nodes_recycled"] > 0


def test_go9_network_path_toy_forward(emu):
    A = 82

    def engine_forward(planes_ptr, batch, logits_ptr, value_ptr):
        planes = np.ctypeslib.as_array(C.cast(planes_ptr, C.POINTER(C.c_float)), shape=(batch, 17, 9, 9))
        lo, va = parity.toy_forward_numpy(planes, A)
        np.ctypeslib.as_array(C.cast(logits_ptr, C.POINTER(C.c_float)), shape=(batch, A))[:] = lo
        np.ctypeslib.as_array(C.cast(value_ptr, C.POINTER(C.c_float)), shape=(batch,))[:] = va
        return 0

    cfg, rec, st = parity.run_engine(emu, "go9", 2, forward=engine_forward, concurrent_games=2, num_traversals=24, seed=5)
    cb = po.make_forward(lambda x: parity.toy_forward_numpy(x, A), po.GAME_GO9)
    ora = po.selfplay(parity.oracle_config("go9", cfg, po.EVAL_CALLBACK, forward=cb), 2, 5, 1, True)
    parity.assert_same_games(rec, ora)
    parity.assert_same_counters(st, ora["stats"])


def test_more_games_than_slots_and_stream_base(emu):
    rec, st = parity.check_case(emu, "c4", 5, concurrent_games=2, num_traversals=30, seed=99, stream_base=17)
    assert rec.num_games == 5 and st["games"] == 5


def test_network_path_toy_forward(emu):
    """Encode (symmetrised planes) -> forward -> decode (exp / wrong-frame mask / normalise / inverse symmetry)
    through the callback evaluator, against the oracle's restatement of GridNetwork::evaluate."""
    A, cells = 65, 64

    def engine_forward(planes_ptr, batch, logits_ptr, value_ptr):
        planes = np.ctypeslib.as_array(C.cast(planes_ptr, C.POINTER(C.c_float)), shape=(batch, 3, 8, 8))
        lo, va = parity.toy_forward_numpy(planes, A)
        np.ctypeslib.as_array(C.cast(logits_ptr, C.POINTER(C.c_float)), shape=(batch, A))[:] = lo
        np.ctypeslib.as_array(C.cast(value_ptr, C.POINTER(C.c_float)), shape=(batch,))[:] = va
        return 0

    cfg, rec, st = parity.run_engine(emu, "othello", 2, forward=engine_forward, concurrent_games=2,
                                     num_traversals=24, seed=5)
    cb = po.make_forward(lambda x: parity.toy_forward_numpy(x, A), po.GAME_OTHELLO)
    ora = po.selfplay(parity.oracle_config("othello", cfg, po.EVAL_CALLBACK, forward=cb), 2, 5, 1, True)
    parity.assert_same_games(rec, ora)
    parity.assert_same_counters(st, ora["stats"])


def test_npy_files_match_oracle_writer(emu, tmp_path):
    cfg, rec, _ = parity.run_engine(emu, "c4", 2, concurrent_games=2, num_traversals=30, seed=3)
    rec.write_npy(str(tmp_path / "run_iteration_0"))
    ocfg = parity.oracle_config("c4", cfg, po.EVAL_RANDOM)
    ora = po.selfplay(ocfg, 2, 3, 1, True)
    po.write_records(ocfg, str(tmp_path / "ora_iteration_0"), ora)
    for part in ("states", "distributions", "outcomes"):
        a = open(tmp_path / f"run_iteration_0_{part}.npy", "rb").read()
        b = open(tmp_path / f"ora_iteration_0_{part}.npy", "rb").read()
        assert a == b, part
    assert not list(tmp_path.glob("*.tmp"))
    s = np.load(tmp_path / "run_iteration_0_states.npy")
    assert s.dtype == np.float32 and s.shape[1:] == (3, 6, 7)


def test_error_codes(emu):
    cfg = E.default_config("othello", emu, concurrent_games=1, num_traversals=40, node_cap=40, spare_arenas=1)
    eng = E.Engine(cfg, emu)
    with pytest.raises(E.SprlError) as ei:
        eng.run(1)
    assert ei.value.code == -3          # SPRL_E_NODEPOOL: live subtree cannot fit 40 nodes
    eng.close()
    for bad in (dict(max_queue=9), dict(num_traversals=0), dict(stream_base=0), dict(concurrent_games=0),
                dict(num_traversals=4, max_queue=4)):
        with pytest.raises(E.SprlError) as ei:
            E.Engine(E.default_config("othello", emu, **bad), emu)
        assert ei.value.code == -1
    eng = E.Engine(E.default_config("c4", emu, concurrent_games=1), emu)
    with pytest.raises(E.SprlError):
        eng.set_model("heuristic")      # Othello only
    with pytest.raises(E.SprlError) as ei:
        eng.step(1)                     # begin() not called
    assert ei.value.code == -5
    eng.close()


# ---- match play (Evaluate.cpp): two trees per game, move + RNG hand-over between launches ----

def test_match_c4_random_vs_random(emu):
    agents = [dict(model="random", use_symmetry=True, parent_q=True), dict(model="random", use_symmetry=False, parent_q=False)]
    w, _, n = parity.check_match(emu, "c4", agents, 5, concurrent_games=2, num_traversals=40)
    assert (n > 6).all()


def test_match_othello_random_vs_heuristic(emu):
    agents = [dict(model="random", use_symmetry=True, parent_q=False), dict(model="heuristic", use_symmetry=True, parent_q=True)]
    w, _, n = parity.check_match(emu, "othello", agents, 4, concurrent_games=4, num_traversals=32, seed=777)
    assert sum(E.match_score(w)) == 4 and (n >= 9).all()


def test_match_go7_and_compaction(emu):
    agents = [dict(model="random", use_symmetry=True, parent_q=True), dict(model="random", use_symmetry=True, parent_q=True)]
    parity.check_match(emu, "go", agents, 2, concurrent_games=2, num_traversals=24, max_batch=4, max_queue=2, node_cap=200)
    # more games than pairs and a search that fits into ONE launch: the side that ends a game starts the pair's next game and may
    # post its first move there before the partner has picked up the last move of the old game (found in round 3: the one-entry
    # mailbox lost that move and the match stalled; the mailbox has two entries per slot now)
    parity.check_match(emu, "go", agents, 3, concurrent_games=2, num_traversals=8, max_batch=4, max_queue=2)
    parity.check_match(emu, "c4", agents, 7, concurrent_games=2, num_traversals=8, max_batch=4, max_queue=2)


def test_match_wide_boards(emu):
    """Evaluate.cpp-style matches on the multi-strip kernel (round 3: play.hpp:24-70 for boards wider than 8): Go 7x7 through the
    wide kernel and Go 9x9, agents with different symmetrisation / InitQ, a forced compaction - move lists, lengths and winners
    equal the oracle's playGame restatement."""
    agents = [dict(model="random", use_symmetry=True, parent_q=True), dict(model="random", use_symmetry=False, parent_q=False)]
    parity.check_match(emu, "go7_wide", agents, 3, concurrent_games=2, num_traversals=24, max_batch=4, max_queue=2, max_plies=120)
    parity.check_match(emu, "go9", agents, 2, concurrent_games=2, num_traversals=20, max_batch=4, max_queue=2, max_plies=200, seed=5)
    parity.check_match(emu, "go9", agents[::-1], 2, concurrent_games=1, num_traversals=16, max_batch=4, max_queue=2, max_plies=200, seed=6,
                       node_cap=160, spare_arenas=2)


def test_match_network_agents_toy_forward(emu):
    """Two network agents with DIFFERENT forwards: rows of the dense batch are split per agent."""
    A = 65

    def make(scale):
        def engine_forward(planes_ptr, batch, logits_ptr, value_ptr):
            planes = np.ctypeslib.as_array(C.cast(planes_ptr, C.POINTER(C.c_float)), shape=(batch, 3, 8, 8))
            lo, va = parity.toy_forward_numpy(planes, A)
            np.ctypeslib.as_array(C.cast(logits_ptr, C.POINTER(C.c_float)), shape=(batch, A))[:] = lo * np.float32(scale)
            np.ctypeslib.as_array(C.cast(value_ptr, C.POINTER(C.c_float)), shape=(batch,))[:] = va
            return 0

        def oracle_forward(x):
            lo, va = parity.toy_forward_numpy(x, A)
            return lo * np.float32(scale), va
        return engine_forward, po.make_forward(oracle_forward, po.GAME_OTHELLO)

    e0, o0 = make(1.0)
    e1, o1 = make(0.25)
    agents = [dict(model="net", use_symmetry=True, parent_q=True), dict(model="net", use_symmetry=False, parent_q=True)]
    parity.check_match(emu, "othello", agents, 3, concurrent_games=2, num_traversals=20, forwards=(e0, e1),
                       oracle_forwards=(o0, o1))


def test_match_mixed_network_and_random(emu):
    A = 43

    def engine_forward(planes_ptr, batch, logits_ptr, value_ptr):
        planes = np.ctypeslib.as_array(C.cast(planes_ptr, C.POINTER(C.c_float)), shape=(batch, 3, 6, 7))
        lo, va = parity.toy_forward_numpy(planes, A)
        np.ctypeslib.as_array(C.cast(logits_ptr, C.POINTER(C.c_float)), shape=(batch, A))[:] = lo
        np.ctypeslib.as_array(C.cast(value_ptr, C.POINTER(C.c_float)), shape=(batch,))[:] = va
        return 0
    A = 7
    cb = po.make_forward(lambda x: parity.toy_forward_numpy(x, A), po.GAME_C4)
    agents = [dict(model="random", use_symmetry=True, parent_q=True), dict(model="net", use_symmetry=True, parent_q=False)]
    parity.check_match(emu, "c4", agents, 4, concurrent_games=3, num_traversals=30, forwards=(None, engine_forward),
                       oracle_forwards=(None, cb))


def test_second_run_on_one_engine_continues_streams_and_counters(emu):
    """Run k on an engine plays games with streams stream_base + (games of earlier runs) + g; stats() covers all runs."""
    cfg = E.default_config("c4", emu, concurrent_games=2, num_traversals=30, seed=21, stream_base=5)
    eng = E.Engine(cfg, emu)
    eng.set_model("random")
    ocfg = parity.oracle_config("c4", cfg, po.EVAL_RANDOM)
    rec1 = eng.run(3)
    ora1 = po.selfplay(ocfg, 3, 21, 5, True)
    parity.assert_same_games(rec1, ora1)
    rec2 = eng.run(2)
    ora2 = po.selfplay(ocfg, 2, 21, 5 + 3, True)
    parity.assert_same_games(rec2, ora2)
    st = eng.stats()
    for k in ("games", "plies", "traversals", "expansions", "nodes_created"):
        assert st[k] == ora1["stats"][k] + ora2["stats"][k], k
    eng.close()


# ---- resign option (not in the reference, default off): device == oracle, sample of the resign ply kept ----

@pytest.mark.parametrize("game,kw", [("c4", dict(num_traversals=40)), ("othello", dict(num_traversals=32)),
                                     ("go", dict(num_traversals=24, max_batch=4, max_queue=2)),
                                     ("go9", dict(num_traversals=20, max_batch=4, max_queue=2))])
def test_resign_threshold(emu, game, kw):
    n = 6 if game in ("c4", "othello") else 2
    plain, _ = parity.check_case(emu, game, n, concurrent_games=2, seed=31, **kw)
    rec, st = parity.check_case(emu, game, n, concurrent_games=2, seed=31, resign_threshold=0.05, resign_min_ply=4, **kw)
    assert rec.total_plies < plain.total_plies          # some game was cut short
    assert (rec.ply_offset[1:] - rec.ply_offset[:-1] > 4).all()


@pytest.mark.parametrize("form", ["label", "flood"])
def test_wide_go_legal_move_algorithms_agree_with_oracle(emu, form, monkeypatch):
    """The wide kernel has two exact legal-move computations (group labels / per-lane flood fill) chosen by board size;
    the SPRL_GO_LEGAL hook forces one of them so that each is checked against the oracle on 7x7 and 9x9 whole games
    (19x19 runs the label form by default: test_go19_six_strip_rows)."""
    monkeypatch.setenv("SPRL_GO_LEGAL", form)
    parity.check_case(emu, "go7_wide", 2, concurrent_games=2, num_traversals=24, max_batch=4, max_queue=2, seed=9)
    parity.check_case(emu, "go9", 1, concurrent_games=1, num_traversals=20, max_batch=4, max_queue=2, seed=9)
