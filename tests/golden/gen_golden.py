"""Generate the golden fixtures in tests/golden/ from the reference itself.

Runs ONLY in the container that has /root/reference: it drives oracle/_ref/libsprl_ref*.so (the
reference's own C++ sources compiled in place by `make -C oracle ref ref_torch`) and, for G9 only,
imports the reference's Python network definition.  The outputs are data (inputs + expected outputs);
no reference source text is stored.

    python tests/golden/gen_golden.py
"""
import hashlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))

from oracle import pyref  # noqa: E402

SEED = 12345


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path)} bytes")


def g6_rng():
    L = pyref.lib()
    out = {}
    L.ref_seed(SEED, 1)
    raw = np.zeros(1000, np.uint32)
    L.ref_rng_raw(1000, pyref.vp(raw))
    out["raw_seed12345_stream1"] = raw
    L.ref_seed(987654321987, 77)
    raw2 = np.zeros(64, np.uint32)
    L.ref_rng_raw(64, pyref.vp(raw2))
    out["raw_seed987654321987_stream77"] = raw2
    L.ref_seed(SEED, 2)
    ks = np.array([1, 2, 3, 4, 5, 7, 8, 10, 13, 33, 60, 65, 1000, 2 ** 31 - 1] * 16, np.int32)
    out["uniform_int_k"] = ks
    out["uniform_int"] = np.array([L.ref_uniform_int(0, int(k) - 1) for k in ks], np.int32)
    out["uniform_float"] = np.array([L.ref_uniform_float() for _ in range(256)], np.float32)
    for alpha, k in ((0.3, 10), (0.5, 7), (0.2, 50), (1.0, 5), (2.5, 6), (0.3, 1)):
        rows = []
        for _ in range(8):
            v = np.zeros(k, np.float32)
            L.ref_dirichlet(alpha, k, pyref.vp(v))
            rows.append(v)
        out[f"dirichlet_a{alpha}_k{k}"] = np.stack(rows)
    cdf = np.cumsum(np.array([0, 0, 0.1, 0.2, 0, 0.3, 0.4, 0], np.float32)).astype(np.float32)
    out["cdf"] = cdf
    out["sample_cdf"] = np.array([L.ref_sample_cdf(pyref.vp(cdf), len(cdf)) for _ in range(256)], np.int32)
    out["final_state"] = np.array([L.ref_rng_state()], np.uint64)
    save("g6_rng.npz", **out)


def g1_playouts():
    out = {}
    for game in ("othello", "c4", "go"):
        for i, seed in enumerate((11, 22, 33, 44)):
            r = pyref.playout(game, seed, 1)
            for k, v in r.items():
                out[f"{game}_{i}_{k}"] = v
            out[f"{game}_{i}_seed"] = np.array([seed], np.int64)
    rw = np.zeros(2, np.float32)
    ok = pyref.lib().ref_c4_known_answer(pyref.vp(rw))
    out["c4_known_answer_ok"] = np.array([ok], np.int32)
    out["c4_known_answer_rewards"] = rw
    save("g1_playouts.npz", **out)


def g2_symmetries():
    rng = np.random.default_rng(5)
    out = {}
    for game, cells, A in (("othello", 64, 65), ("c4", 42, 7)):
        board = rng.integers(-1, 2, cells).astype(np.int8)
        dist = rng.random(A).astype(np.float32)
        bo, do, inv = pyref.symmetrize(game, board, 1, dist)
        out[f"{game}_board"] = board
        out[f"{game}_dist"] = dist
        out[f"{game}_boards_out"] = bo
        out[f"{game}_dists_out"] = do
        out[f"{game}_inverse"] = inv
    save("g2_symmetries.npz", **out)


def g4_search():
    out = {}
    for game, alpha in (("othello", 0.3), ("c4", 0.5)):
        for kind in ((0, 1) if game == "othello" else (0,)):
            for (mb, mq) in ((8, 4), (1, 1)):
                st, tr, ch = pyref.search_trace(game, kind, 3, 200, mb, mq, 0.25, alpha, SEED, 1)
                key = f"{game}_k{kind}_b{mb}q{mq}"
                out[key + "_stats"] = st
                out[key + "_trav"] = tr
                out[key + "_chosen"] = ch
    # Q8 demonstration: start-position priors per forced symmetry cannot be forced from outside; instead
    # keep a no-noise trace whose root priors show zeros where the drawn symmetry moved the mask (Q1).
    st, tr, ch = pyref.search_trace("go", 0, 3, 200, 16, 8, 0.25, 0.2, SEED, 1)
    out["go_k0_b16q8_stats"], out["go_k0_b16q8_trav"], out["go_k0_b16q8_chosen"] = st, tr, ch
    st, tr, ch = pyref.search_trace("othello", 0, 1, 16, 8, 4, 0.25, 0.3, SEED, 1, use_sym=1, add_noise=0)
    out["othello_nonoise_stats"] = st
    save("g4_search.npz", **out)


def g5_games():
    out = {}
    r = pyref.selfplay("othello", 0, 2, 60, 8, 4, 0.25, 0.3, SEED, 1, True)
    for k in ("boards", "players", "dists", "outcomes", "offsets"):
        out["oth_random_" + k] = r[k]
    r = pyref.selfplay("othello", 1, 1, 40, 8, 4, 0.25, 0.3, SEED, 5, True)
    for k in ("boards", "players", "dists", "outcomes", "offsets"):
        out["oth_heur_" + k] = r[k]
    r = pyref.selfplay("c4", 0, 4, 100, 8, 4, 0.25, 0.5, SEED, 1, True)
    for k in ("boards", "players", "dists", "outcomes", "offsets"):
        out["c4_random_" + k] = r[k]
    # single global stream over several games (what the reference worker really does)
    r = pyref.selfplay("c4", 0, 3, 100, 8, 4, 0.25, 0.5, SEED, 9, False)
    for k in ("boards", "players", "dists", "outcomes", "offsets"):
        out["c4_single_stream_" + k] = r[k]
    r = pyref.selfplay("othello", 0, 1, 30, 1, 1, 0.25, 0.3, SEED, 3, True, use_sym=0, add_noise=0)
    for k in ("boards", "players", "dists", "outcomes", "offsets"):
        out["oth_nosym_b1q1_" + k] = r[k]
    # Go 7x7 (the size the reference compiles, GoNode.hpp:16): 8-ply history states, superko, Tromp-Taylor + komi
    r = pyref.selfplay("go", 0, 3, 120, 16, 8, 0.25, 0.2, SEED, 1, True)
    for k in ("boards", "players", "sizes", "dists", "outcomes", "offsets"):
        out["go_random_" + k] = r[k]
    save("g5_games.npz", **out)

    # worker byte streams (runWorker + vendored npy writer)
    L = pyref.lib(True)
    w = {}
    with tempfile.TemporaryDirectory() as td:
        L.ref_c4_run_worker(b"gold", td.encode(), 3, 100, 8, 4, 0.25, 0.5, SEED, 1)
        L.ref_othello_run_worker(b"goldoth", td.encode(), 0, 1, 30, 8, 4, 0.25, 0.3, SEED, 1)
        for run in ("gold", "goldoth"):
            for part in ("states", "distributions", "outcomes"):
                b = open(os.path.join(td, f"{run}_iteration_0_{part}.npy"), "rb").read()
                w[f"{run}_{part}"] = np.frombuffer(b, np.uint8)
                print(run, part, len(b), hashlib.sha256(b).hexdigest()[:16])
    save("g5_worker_npy.npz", **w)


def g7_g9_network():
    import torch
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    from src.networks.grid_networks import BasicGridNetwork  # reference definition, this container only

    torch.manual_seed(3)
    ref_net = BasicGridNetwork(8, 8, 65, 1, 1, 8).eval()
    with torch.no_grad():
        for m in ref_net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0.0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
    x = (torch.rand(5, 3, 8, 8) > 0.6).float()
    with torch.no_grad():
        logits, value = ref_net(x)
    out = {"input": x.numpy(), "logits": logits.numpy(), "value": value.numpy()}
    for k, v in ref_net.state_dict().items():
        out["sd::" + k] = v.numpy()
    save("g9_network.npz", **out)

    # G7: the reference GridNetwork::evaluate (LibTorch CPU) through a traced model of OUR module
    from sprl_amd.network import GridResNet, trace_to_file
    net = GridResNet(8, 8, 65, 1, 1, 8)
    net.load_state_dict(ref_net.state_dict())
    net.eval()
    with tempfile.TemporaryDirectory() as td:
        path = trace_to_file(net, os.path.join(td, "tiny.pt"), "othello")
        po = pyref.playout("othello", 77, 1)
        idx = [0, 5, 17, 30, len(po["players"]) - 2]
        boards, players, masks = po["boards"][idx], po["players"][idx], po["masks"][idx].copy()
        masks[3, :] = 0.0
        masks[3, 64] = 1.0            # pass-only mask
        pol, val = pyref.othello_evaluate(0, boards, players, masks, model_path=path)
    save("g7_decode.npz", boards=boards, players=players, masks=masks, policy=pol, value=val)


def g9b_baseline_network():
    """The BASELINE-shape network (BasicGridNetwork(8, 8, 65, 1, 2, 64), scripts/othello_controller.py:297) evaluated by the
    REFERENCE's own module on CPU in fp32: seed-reproducible weights (netfill.py), 64 Othello-like inputs, outputs."""
    import torch
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    from src.networks.grid_networks import BasicGridNetwork  # reference definition, this container only
    sys.path.insert(0, OUT)
    import netfill
    seed = 20261004
    x = netfill.othello_like_inputs(64, seed + 1)
    out = dict(seed=np.array([seed], np.int64), input=x, gains=np.array([1.0, 2.0, 3.0]))
    # default-init scale ("random-init net", |logits| <= 0.1), |logits| <= 3.3, and |logits| <= 16 (a sharply trained policy head)
    for gi, gain in enumerate((1.0, 2.0, 3.0)):
        ref_net = netfill.fill_state_dict(BasicGridNetwork(8, 8, 65, 1, 2, 64), seed + gi, gain).eval()
        with torch.no_grad():
            logits, value = ref_net(torch.from_numpy(x))
            ld, vd = ref_net.double()(torch.from_numpy(x).double())      # float64 forward of the same weights: the exact answer
        out[f"logits{gi}"], out[f"value{gi}"] = logits.numpy(), value.numpy()
        out[f"logits_f64_{gi}"], out[f"value_f64_{gi}"] = ld.numpy(), vd.numpy()
        out["keys"] = np.array(list(ref_net.state_dict().keys()))
    save("g9b_baseline_network.npz", **out)


def g_trainer():
    """f-2 fixture: the REFERENCE controller's train_network (scripts/othello_controller.py:128-241) run on a fixed synthetic
    window - seed-reproducible network weights (netfill.py) and samples - with its group / epoch / batch constants scaled down.
    Recorded: the batches it drew (its DataLoaders are wrapped, nothing else is touched), the epoch it selected as best and
    the outputs of the model it saved, so that sprl_amd/trainer.py can be replayed on the same batches and compared."""
    import contextlib
    import io
    import re
    import torch
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    sys.path.insert(0, OUT)
    import netfill
    import scripts.othello_controller as oc           # reference module, this container only (its main() is not run)
    from src.networks.grid_networks import BasicGridNetwork

    seed, n = 777, 320
    rs = np.random.RandomState(seed)
    states = netfill.othello_like_inputs(n, seed)
    raw = rs.standard_normal((n, 65)).astype(np.float32) * 2.0
    dists = np.exp(raw) / np.exp(raw).sum(1, keepdims=True)
    outcomes = rs.randint(-1, 2, size=(n, 1)).astype(np.float32)
    stamps = (1.0 + np.arange(n, dtype=np.float32) / 1024.0).reshape(n, 1) + rs.randint(0, 3, size=(n, 1)).astype(np.float32)
    assert len(np.unique(stamps)) == n                 # unique weights: a batch identifies its samples
    index_of = {float(v): i for i, v in enumerate(stamps[:, 0])}

    batches = []

    class RecordingLoader(oc.DataLoader):
        def __iter__(self):
            for b in super().__iter__():
                batches.append(np.array([index_of[float(v)] for v in b[3][:, 0]], np.int32))
                yield b

    consts = dict(MAX_GROUPS=3, EPOCHS_PER_GROUP=4, BATCH_SIZE=48, RUN_NAME="fixture", device="cpu")
    for k, v in consts.items():
        setattr(oc, k, v)
    oc.DataLoader = RecordingLoader
    lr = float(os.environ.get('SPRL_GEN_TRAINER_LR', '0.0005'))
    with tempfile.TemporaryDirectory() as td:
        cwd = os.getcwd()
        os.chdir(td)
        try:
            os.makedirs("data/models/fixture")
            torch.manual_seed(seed)
            net = netfill.fill_state_dict(BasicGridNetwork(8, 8, 65, 1, 1, 8), seed)
            buf = io.StringIO()
            import pickle
            with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
                try:
                    oc.train_network(net, lr, 0, torch.from_numpy(states), torch.from_numpy(dists.astype(np.float32)),
                                     torch.from_numpy(outcomes), torch.from_numpy(stamps))
                except pickle.UnpicklingError:
                    # the reference's last step, trace_model (src/interface/tracer.py:17), calls torch.load(path) on a pickled
                    # module, which torch >= 2.6 (weights_only=True by default; the reference pins 2.2.2) refuses: training and
                    # the best-model file are complete at that point; the trace is redone below exactly as tracer.py:18 does it
                    pass
            best_epoch = int(re.search(r"best model was at epoch (\d+)", buf.getvalue()).group(1))
            best = torch.load("data/models/fixture/fixture_iteration_0.pt", weights_only=False).eval()
            traced = torch.jit.trace(best, torch.randn(1, 3, 8, 8)).eval()
            probe = torch.from_numpy(states[:16])
            with torch.no_grad():
                lo, va = best(probe)
                tlo, tva = traced(probe)
                flo, fva = net.eval()(probe)             # the live network after the last epoch (carried to the next iteration)
        finally:
            os.chdir(cwd)
    n_train = int(0.9 * n)
    per_epoch_train = -(-n_train // consts["BATCH_SIZE"])
    per_epoch_val = -(-(n - n_train) // consts["BATCH_SIZE"])
    epochs = len(batches) // (per_epoch_train + per_epoch_val)
    assert epochs * (per_epoch_train + per_epoch_val) == len(batches)
    out = dict(seed=np.array([seed]), n=np.array([n]), lr=np.array([lr]), best_epoch=np.array([best_epoch]), epochs=np.array([epochs]),
               max_groups=np.array([consts["MAX_GROUPS"]]), epochs_per_group=np.array([consts["EPOCHS_PER_GROUP"]]),
               batch_size=np.array([consts["BATCH_SIZE"]]), batches_per_epoch=np.array([per_epoch_train, per_epoch_val]),
               dists=dists.astype(np.float32), outcomes=outcomes, stamps=stamps,
               best_logits=lo.numpy(), best_value=va.numpy(), traced_logits=tlo.numpy(), traced_value=tva.numpy(),
               final_logits=flo.numpy(), final_value=fva.numpy())
    for i, b in enumerate(batches):
        out[f"batch{i}"] = b
    save("g_trainer.npz", **out)


MATCH_CASES = [  # game, kind0, kind1, games, traversals, batch, queue, sym0, parentQ0, sym1, parentQ1, seed
    ("othello", 0, 1, 6, 64, 8, 4, 1, 1, 1, 1, 777),
    ("othello", 0, 1, 4, 100, 8, 4, 0, 0, 1, 1, 778),
    ("othello", 1, 0, 4, 48, 1, 1, 1, 0, 0, 1, 779),
    ("c4", 0, 0, 8, 100, 8, 4, 1, 1, 0, 0, 780),
    ("c4", 0, 0, 6, 40, 4, 2, 0, 1, 1, 0, 781),
]


def g10_matches():
    """Evaluate.cpp-style agent-vs-agent games through the reference's UCTNetworkAgent + playGame."""
    out = {"cases": np.array([[c[1:][i] for i in range(11)] for c in MATCH_CASES], np.int64),
           "games": np.array([c[0] for c in MATCH_CASES])}
    for i, (game, k0, k1, n, trav, mb, mq, s0, p0, s1, p1, seed) in enumerate(MATCH_CASES):
        w, a, npl = pyref.match(game, k0, k1, n, trav, mb, mq, s0, p0, s1, p1, seed, 1, 160)
        out[f"winners{i}"], out[f"actions{i}"], out[f"nplies{i}"] = w, a, npl
    save("g10_matches.npz", **out)


def g_go9():
    """Go at 9x9 (BASELINE config 4): the reference compiled with only GO_BOARD_WIDTH = 9 and GO_KOMI = 7.5 changed
    (oracle/Makefile: ref_go9) - rules play-outs, a search trace and whole games."""
    assert pyref.lib(variant="go9").ref_go_board_width() == 9 and abs(pyref.lib(variant="go9").ref_go_komi() - 7.5) < 1e-6
    out = {}
    for i, seed in enumerate((11, 22, 33, 44)):
        r = pyref.playout("go9", seed, 1)
        for k, v in r.items():
            out[f"playout_{i}_{k}"] = v
        out[f"playout_{i}_seed"] = np.array([seed], np.int64)
    st, tr, ch = pyref.search_trace("go9", 0, 3, 200, 16, 8, 0.25, 0.2, SEED, 1)
    out["trace_stats"], out["trace_trav"], out["trace_chosen"] = st, tr, ch
    r = pyref.selfplay("go9", 0, 2, 64, 16, 8, 0.25, 0.2, SEED, 1, True)
    for k in ("boards", "players", "sizes", "dists", "outcomes", "offsets"):
        out["games_" + k] = r[k]
    r = pyref.selfplay("go9", 0, 1, 40, 4, 2, 0.25, 0.2, SEED, 7, True, use_sym=0, add_noise=0)
    for k in ("boards", "players", "sizes", "dists", "outcomes", "offsets"):
        out["games_nosym_" + k] = r[k]
    save("g_go9.npz", **out)


def g_go19():
    """Go at 19x19 (BASELINE config 5): the reference compiled with GO_BOARD_WIDTH = 19, GO_KOMI = 7.5 and its two index
    types widened to int16_t (games/GoNode.hpp:16,20,36-37; oracle/Makefile: ref_go19, four changed lines) - rules play-outs
    (captures, superko, the 2N depth cap, Tromp-Taylor + komi), a 200-traversal search trace at the worker's 16/8 batching
    and whole games."""
    L = pyref.lib(variant="go19")
    assert L.ref_go_board_width() == 19 and abs(L.ref_go_komi() - 7.5) < 1e-6
    out = {}
    for i, seed in enumerate((11, 22, 33, 44)):
        r = pyref.playout("go19", seed, 1, 800)
        for k, v in r.items():
            out[f"playout_{i}_{k}"] = v
        out[f"playout_{i}_seed"] = np.array([seed], np.int64)
    st, tr, ch = pyref.search_trace("go19", 0, 3, 200, 16, 8, 0.25, 0.2, SEED, 1)
    out["trace_stats"], out["trace_trav"], out["trace_chosen"] = st, tr, ch
    r = pyref.selfplay("go19", 0, 1, 32, 16, 8, 0.25, 0.2, SEED, 1, True)
    for k in ("boards", "players", "sizes", "dists", "outcomes", "offsets"):
        out["games_" + k] = r[k]
    r = pyref.selfplay("go19", 0, 1, 40, 4, 2, 0.25, 0.2, SEED, 7, True, use_sym=0, add_noise=0)
    for k in ("boards", "players", "sizes", "dists", "outcomes", "offsets"):
        out["games_nosym_" + k] = r[k]
    save("g_go19.npz", **out)


def g9c_go_networks():
    """The Go-shape networks of BASELINE configs 4 / 5 (scripts/go_controller.py:44-45: MODEL_NUM_BLOCKS = 6, 64 channels, history 8
    -> 17 input planes): BasicGridNetwork(9, 9, 82, 8, 6, 64) and (19, 19, 362, 8, 6, 64) evaluated by the REFERENCE's own module
    on CPU in fp32 and in float64, netfill weights at gains 1 and 2, Go-like inputs.  Stored: seeds, outputs (inputs and weights
    are reproducible from the seeds alone, netfill.py)."""
    import torch
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    from src.networks.grid_networks import BasicGridNetwork  # reference definition, this container only
    sys.path.insert(0, OUT)
    import netfill
    torch.set_num_threads(8)
    seed = 20261005
    out = dict(seed=np.array([seed], np.int64), gains=np.array([1.0, 2.0]), history=np.array([8], np.int64),
               blocks=np.array([6], np.int64), channels=np.array([64], np.int64))
    for width, actions, n in ((9, 82, 16), (19, 362, 8)):
        x = netfill.go_like_inputs(n, width, 8, seed + width)
        out[f"n{width}"] = np.array([n], np.int64)
        for gi, gain in enumerate((1.0, 2.0)):
            ref_net = netfill.fill_state_dict(BasicGridNetwork(width, width, actions, 8, 6, 64), seed + 100 * width + gi, gain).eval()
            with torch.no_grad():
                logits, value = ref_net(torch.from_numpy(x))
                ld, vd = ref_net.double()(torch.from_numpy(x).double())
            out[f"logits{width}_{gi}"], out[f"value{width}_{gi}"] = logits.numpy(), value.numpy()
            out[f"logits{width}_f64_{gi}"], out[f"value{width}_f64_{gi}"] = ld.numpy(), vd.numpy()
            print(f"  {width}x{width} gain {gain}: |logits| <= {np.abs(logits.numpy()).max():.3f}, fp32 vs f64 {np.abs(logits.numpy() - ld.numpy()).max():.2e}")
    save("g9c_go_networks.npz", **out)


def _cnn_dist_worker(args):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity
    return parity.reference_cnn_games(args)


def g_cnn_dist(games=1024, procs=8):
    """Per-game statistics of `games` Othello games of the REFERENCE's selfPlay + GridNetwork (LibTorch-CPU, oracle/_ref) with a
    traced 2 x 64 network whose weights come from netfill (seed, gain 2: logits of a few units, so the policy head matters):
    200 traversals/move, batch 8 / queue 4, D4, Dirichlet(0.25, 0.3).  Columns: plies, outcome of Player ZERO, network
    evaluations, mean entropy of the root-visit pdfs.  The GPU test plays as many games with the same model and compares."""
    import multiprocessing as mp
    import time
    sys.path.insert(0, OUT)
    import netfill
    from sprl_amd.network import GridResNet, trace_to_file
    seed, gain, trav, rng_seed = 20261006, 2.0, 200, 777
    net = netfill.fill_state_dict(GridResNet(8, 8, 65, 1, 2, 64), seed, gain).eval()
    with tempfile.TemporaryDirectory() as td:
        path = trace_to_file(net, os.path.join(td, "dist.pt"), "othello")
        per = games // procs
        t0 = time.time()
        with mp.get_context("spawn").Pool(procs) as pool:
            parts = pool.map(_cnn_dist_worker, [(path, per, trav, rng_seed, 100000 + per * i) for i in range(procs)])
    rows = np.array([r for part in parts for r in part], np.float64)
    print(f"  {len(rows)} reference games in {time.time() - t0:.0f} s: plies {rows[:, 0].mean():.2f} +- {rows[:, 0].std():.2f}, outcome "
          f"{rows[:, 1].mean():+.3f}, evals {rows[:, 2].mean():.1f} +- {rows[:, 2].std():.1f}, entropy {rows[:, 3].mean():.4f} +- {rows[:, 3].std():.4f}")
    save("g_cnn_dist.npz", stats=rows, net_seed=np.array([seed], np.int64), gain=np.array([gain]), traversals=np.array([trav], np.int64),
         rng_seed=np.array([rng_seed], np.int64), first_stream=np.array([100000], np.int64))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "trainer":
        g_trainer()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g9b":
        g9b_baseline_network()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g9c":
        g9c_go_networks()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g_cnn_dist":
        assert pyref.available(True), "run `make -C oracle ref_torch` first"
        g_cnn_dist()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "go9":          # only the 9x9 fixtures (needs `make -C oracle ref_go9`)
        assert pyref.available(variant="go9"), "run `make -C oracle ref_go9` first"
        g_go9()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "go19":        # only the 19x19 fixtures (needs `make -C oracle ref_go19`)
        assert pyref.available(variant="go19"), "run `make -C oracle ref_go19` first"
        g_go19()
        sys.exit(0)
    assert pyref.available() and pyref.available(True), "run `make -C oracle ref ref_torch` first"
    g6_rng()
    g1_playouts()
    g2_symmetries()
    g4_search()
    g5_games()
    g7_g9_network()
    g9b_baseline_network()
    g9c_go_networks()
    g_cnn_dist()
    g_trainer()
    g10_matches()
    if pyref.available(variant="go9"):
        g_go9()
    if pyref.available(variant="go19"):
        g_go19()
