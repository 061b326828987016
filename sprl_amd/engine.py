"""ctypes binding of libsprl_amd.so (include/sprl_amd.h) — the Python face of the C ABI.

Host-side mirror of the reference's `runIteration` (cpp/src/selfplay/SelfPlay.hpp:204-248) and record
writer (cpp/src/selfplay/GridWorker.hpp:146-196).  The search runs in hand-written gfx950 kernels; there
is no CPU fallback: importing works anywhere, creating an engine without an MI355X raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "libsprl_amd.so")

OTHELLO, CONNECT_FOUR, GO7, GO9, GO19, GO7_WIDE = 0, 1, 2, 3, 4, 5
EVAL_RANDOM, EVAL_HEURISTIC, EVAL_NETWORK = 0, 1, 2
MASK_REFERENCE, MASK_SYMMETRISED = 0, 1
GAME_IDS = {"othello": OTHELLO, "connect_four": CONNECT_FOUR, "c4": CONNECT_FOUR, "go": GO7, "go7": GO7, "go9": GO9, "go19": GO19,
            "go7_wide": GO7_WIDE}


class SprlError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"sprl_amd error {code}: {message}")
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("game", C.c_int32), ("device", C.c_int32), ("concurrent_games", C.c_int32), ("num_traversals", C.c_int32),
        ("max_batch", C.c_int32), ("max_queue", C.c_int32), ("dir_eps", C.c_float), ("dir_alpha", C.c_float),
        ("u_weight", C.c_float), ("early_cutoff", C.c_int32), ("early_exp", C.c_float), ("rest_exp", C.c_float),
        ("use_symmetry", C.c_int32), ("add_noise", C.c_int32), ("mask_frame", C.c_int32), ("node_cap", C.c_int32),
        ("spare_arenas", C.c_int32), ("max_plies", C.c_int32), ("seed", C.c_uint64), ("stream_base", C.c_int32),
        ("profile", C.c_int32), ("own_stream", C.c_int32), ("resign_threshold", C.c_float),
        ("resign_min_ply", C.c_int32), ("no_recycle", C.c_int32), ("alloc_base", C.c_int32),
    ]


class Records(C.Structure):
    _fields_ = [
        ("game", C.c_int32), ("num_games", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32),
        ("cells", C.c_int32), ("actions", C.c_int32), ("nsym", C.c_int32), ("use_symmetry", C.c_int32),
        ("history", C.c_int32), ("planes", C.c_int32), ("total_plies", C.c_int64), ("ply_offset", C.POINTER(C.c_int32)), ("boards", C.POINTER(C.c_int8)),
        ("movers", C.POINTER(C.c_int8)), ("pdfs", C.POINTER(C.c_float)), ("winners", C.POINTER(C.c_int8)),
        ("owner_", C.c_void_p),
    ]


class Stats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "games", "plies", "traversals", "levels", "expansions", "nn_evals", "terminal_hits", "gray_hits",
        "dup_hits", "nodes_created", "compactions", "max_nodes_in_arena", "rounds", "kernel_launches",
        "nn_batches", "nn_rows")] + [("seconds_total", C.c_double), ("kernel_ms", C.c_double), ("nn_ms", C.c_double),
                          ("hbm_bytes", C.c_int64)] + [(n, C.c_int64) for n in (
        "cyc_total", "cyc_finish", "cyc_move", "cyc_select", "cyc_create", "cyc_backup", "cyc_leafio", "cyc_noise",
        "cyc_max_slot_launch", "cyc_lvl_wait", "cyc_lvl_pick", "cyc_lvl_desc")] + [
        ("conv_ms", C.c_double), ("conv_launches", C.c_int64), ("conv_boards", C.c_int64), ("nodes_recycled", C.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


FORWARD_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p)

INITQ_PARENT, INITQ_ZERO = 0, 1


class MatchAgent(C.Structure):
    _fields_ = [("model", C.c_char_p), ("forward", FORWARD_FN), ("forward_user", C.c_void_p),
                ("use_symmetry", C.c_int32), ("init_q", C.c_int32)]

_libs = {}
_hip_runtime = None


def _preload_torch_hip_runtime():
    """The engine and the LibTorch-ROCm evaluator must share ONE HIP/HSA runtime in the process.  The torch wheel
    bundles its own (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7); loading it first makes the dynamic
    linker satisfy libsprl_amd.so's libamdhip64.so.7 dependency with that same copy instead of /opt/rocm's."""
    global _hip_runtime
    if _hip_runtime is not None:
        return
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        _hip_runtime = False
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    _hip_runtime = C.CDLL(cand, mode=C.RTLD_GLOBAL) if os.path.exists(cand) else False


def load_library(path=None):
    """Load the C-ABI library.  `path` defaults to the in-tree gfx950 build; a missing library is an error."""
    path = os.path.abspath(path or DEFAULT_LIB)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise SprlError(-4, f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    _preload_torch_hip_runtime()
    L = C.CDLL(path)
    L.sprl_last_error.restype = C.c_char_p
    L.sprl_config_default.argtypes = [C.c_int32, C.POINTER(Config)]
    L.sprl_engine_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    L.sprl_engine_destroy.argtypes = [C.c_void_p]
    L.sprl_engine_destroy.restype = None
    L.sprl_engine_set_model.argtypes = [C.c_void_p, C.c_char_p]
    L.sprl_engine_set_model_buffer.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.sprl_engine_set_forward.argtypes = [C.c_void_p, FORWARD_FN, C.c_void_p]
    L.sprl_engine_evaluator_info.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
    L.sprl_engine_game_evals.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.sprl_engine_conv_kinds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.sprl_engine_run.argtypes = [C.c_void_p, C.c_int32, C.POINTER(Records)]
    L.sprl_engine_begin.argtypes = [C.c_void_p, C.c_int32]
    L.sprl_engine_step.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.sprl_engine_collect.argtypes = [C.c_void_p, C.POINTER(Records)]
    L.sprl_engine_records_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.sprl_engine_pack_records.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.sprl_engine_expand_records.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.sprl_engine_finish.argtypes = [C.c_void_p]
    L.sprl_profile_busy.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.sprl_profile_busy_reset.restype = None
    L.sprl_records_free.argtypes = [C.POINTER(Records)]
    L.sprl_records_free.restype = None
    L.sprl_engine_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.sprl_records_num_samples.argtypes = [C.POINTER(Records)]
    L.sprl_records_num_samples.restype = C.c_int64
    L.sprl_records_expand.argtypes = [C.POINTER(Records), C.c_void_p, C.c_void_p, C.c_void_p]
    L.sprl_records_expand_boards.argtypes = [C.POINTER(Records), C.c_void_p, C.c_void_p]
    L.sprl_write_npy.argtypes = [C.c_char_p, C.POINTER(Records)]
    L.sprl_write_v2.argtypes = [C.c_char_p, C.POINTER(Records)]
    L.sprl_records_slice.argtypes = [C.POINTER(Records), C.c_int32, C.c_int32, C.POINTER(Records)]
    L.sprl_match_play.argtypes = [C.POINTER(Config), C.POINTER(MatchAgent), C.POINTER(MatchAgent), C.c_int32,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    _libs[path] = L
    return L


def default_config(game, lib=None, **overrides):
    L = lib or load_library()
    cfg = Config()
    gid = GAME_IDS[game] if isinstance(game, str) else game
    rc = L.sprl_config_default(gid, C.byref(cfg))
    if rc:
        raise SprlError(rc, L.sprl_last_error().decode())
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError(f"sprl_config has no field {k}")
        setattr(cfg, k, v)
    return cfg


class SelfPlayRecords:
    """Owns one sprl_records; exposes numpy views/copies and the reference's expanded sample arrays."""

    def __init__(self, lib, rec):
        self._lib, self._rec = lib, rec
        r = rec
        n, ng = r.total_plies, r.num_games
        self.game, self.num_games, self.total_plies = r.game, ng, n
        self.rows, self.cols, self.cells, self.actions, self.nsym = r.rows, r.cols, r.cells, r.actions, r.nsym
        self.use_symmetry = bool(r.use_symmetry)
        self.history, self.planes = r.history, r.planes
        self.ply_offset = np.ctypeslib.as_array(r.ply_offset, shape=(ng + 1,)).copy()
        self.boards = np.ctypeslib.as_array(r.boards, shape=(n, r.cells)).copy()
        self.movers = np.ctypeslib.as_array(r.movers, shape=(n,)).copy()
        self.pdfs = np.ctypeslib.as_array(r.pdfs, shape=(n, r.actions)).copy()
        self.winners = np.ctypeslib.as_array(r.winners, shape=(ng,)).copy()

    @property
    def num_samples(self):
        return int(self._lib.sprl_records_num_samples(C.byref(self._rec)))

    def expand(self):
        """(states float32[N,2H+1,R,C], distributions float32[N,A], outcomes float32[N]) — GridWorker.hpp:146-196."""
        n = self.num_samples
        states = np.zeros((n, self.planes, self.rows, self.cols), np.float32)
        dists = np.zeros((n, self.actions), np.float32)
        outs = np.zeros(n, np.float32)
        rc = self._lib.sprl_records_expand(C.byref(self._rec), states.ctypes.data, dists.ctypes.data, outs.ctypes.data)
        if rc:
            raise SprlError(rc, self._lib.sprl_last_error().decode())
        return states, dists, outs

    def expand_boards(self):
        n = self.num_samples
        boards = np.zeros((n, self.history * self.cells), np.int8)
        players = np.zeros(n, np.int8)
        rc = self._lib.sprl_records_expand_boards(C.byref(self._rec), boards.ctypes.data, players.ctypes.data)
        if rc:
            raise SprlError(rc, self._lib.sprl_last_error().decode())
        return boards, players

    def write_npy(self, path_prefix):
        rc = self._lib.sprl_write_npy(os.fsencode(path_prefix), C.byref(self._rec))
        if rc:
            raise SprlError(rc, self._lib.sprl_last_error().decode())

    def close(self):
        if self._rec is not None:
            self._lib.sprl_records_free(C.byref(self._rec))
            self._rec = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One self-play engine on one GPU (sprl_engine_*)."""

    def __init__(self, cfg, lib=None):
        self._lib = lib or load_library()
        self._h = C.c_void_p()
        self._cb = None
        self.cfg = cfg
        rc = self._lib.sprl_engine_create(C.byref(cfg), C.byref(self._h))
        if rc:
            raise SprlError(rc, self._lib.sprl_last_error().decode())

    def _check(self, rc):
        if rc:
            raise SprlError(rc, self._lib.sprl_last_error().decode())

    def set_model(self, model):
        self._check(self._lib.sprl_engine_set_model(self._h, os.fsencode(model)))

    def set_model_module(self, traced):
        """Hot swap: a traced / scripted torch module goes to the evaluator through memory (no .pt file on disk)."""
        import io
        import torch
        buf = io.BytesIO()
        torch.jit.save(traced, buf)
        raw = buf.getvalue()
        self._check(self._lib.sprl_engine_set_model_buffer(self._h, raw, len(raw)))

    def set_model_bytes(self, raw: bytes):
        """Hot swap from a TorchScript archive already in memory (network.trace_to_bytes)."""
        self._check(self._lib.sprl_engine_set_model_buffer(self._h, raw, len(raw)))

    def set_forward(self, fn):
        """fn(planes_ptr, batch, logits_ptr, value_ptr) -> int, all DEVICE pointers (ints)."""
        def _cb(user, planes, batch, logits, value):
            try:
                return int(fn(planes, batch, logits, value) or 0)
            except Exception as exc:  # never let an exception cross the C boundary
                print("forward callback failed:", exc)
                return -1
        self._cb = FORWARD_FN(_cb)
        self._check(self._lib.sprl_engine_set_forward(self._h, self._cb, None))

    def evaluator_info(self):
        buf = C.create_string_buffer(768)
        self._check(self._lib.sprl_engine_evaluator_info(self._h, buf, 768))
        return buf.value.decode()

    def conv_kinds(self):
        """profile = 2: {kind: (ms, launches)} of the trunk-convolution launches: plain / residual / stem / heads."""
        ms, n = (C.c_double * 4)(), (C.c_int64 * 4)()
        self._check(self._lib.sprl_engine_conv_kinds(self._h, ms, n))
        return {k: (ms[i], n[i]) for i, k in enumerate(("plain", "residual", "stem", "heads"))}

    def game_evals(self, num_games):
        """Network evaluations queued by each of the first `num_games` games of the current / last self-play run."""
        out = np.zeros(num_games, np.uint32)
        self._check(self._lib.sprl_engine_game_evals(self._h, out.ctypes.data, num_games))
        return out

    def run(self, num_games):
        rec = Records()
        self._check(self._lib.sprl_engine_run(self._h, num_games, C.byref(rec)))
        return SelfPlayRecords(self._lib, rec)

    def begin(self, num_games):
        self._check(self._lib.sprl_engine_begin(self._h, num_games))

    def step(self, rounds):
        done, active = C.c_int32(), C.c_int32()
        self._check(self._lib.sprl_engine_step(self._h, rounds, C.byref(done), C.byref(active)))
        return done.value, active.value

    def collect(self):
        rec = Records()
        self._check(self._lib.sprl_engine_collect(self._h, C.byref(rec)))
        return SelfPlayRecords(self._lib, rec)

    def records_info(self):
        """(total plies, samples, packed bytes) of the finished run, from the device (no record copy)."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self._lib.sprl_engine_records_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def pack_records_into(self, dst_ptr, capacity_bytes):
        """Write the run's compact records in the gather's wire format into DEVICE memory at `dst_ptr` (16-byte aligned)."""
        self._check(self._lib.sprl_engine_pack_records(self._h, C.c_void_p(dst_ptr), capacity_bytes))

    def expand_records_into(self, states_ptr, dists_ptr, outcomes_ptr, capacity_samples):
        """Write the reference's training samples (planes, distributions, outcomes) into DEVICE buffers."""
        self._check(self._lib.sprl_engine_expand_records(self._h, C.c_void_p(states_ptr), C.c_void_p(dists_ptr),
                                                         C.c_void_p(outcomes_ptr), capacity_samples))

    def finish(self):
        self._check(self._lib.sprl_engine_finish(self._h))

    def stats(self):
        st = Stats()
        self._check(self._lib.sprl_engine_stats(self._h, C.byref(st)))
        return st.as_dict()

    def close(self):
        if self._h:
            self._lib.sprl_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _match_agent(spec, keep):
    """spec: dict(model="random"|"heuristic"|path | forward=callable, use_symmetry=bool, parent_q=bool)."""
    a = MatchAgent()
    fwd = spec.get("forward")
    if fwd is not None:
        def _cb(user, planes, batch, logits, value, fn=fwd):
            try:
                return int(fn(planes, batch, logits, value) or 0)
            except Exception as exc:  # never let an exception cross the C boundary
                print("forward callback failed:", exc)
                return -1
        a.forward = FORWARD_FN(_cb)
        keep.append(a.forward)
    else:
        a.model = os.fsencode(spec["model"])
    a.use_symmetry = 1 if spec.get("use_symmetry", True) else 0
    a.init_q = INITQ_PARENT if spec.get("parent_q", True) else INITQ_ZERO
    return a


def play_match(cfg, agent0, agent1, num_games, max_plies=256, lib=None):
    """Agent-vs-agent games on the device (the reference's Evaluate.exe, cpp/src/Evaluate.cpp).  `cfg` carries the tree
    options shared by both agents (Evaluate.cpp:94-112: dir_eps 0.25, dir_alpha 0.1, noise on).  Returns
    (winners int8[num_games] by colour, actions int16[num_games, max_plies] (-1 padded), nplies int32[num_games]);
    agent k plays colour k ^ (game & 1)."""
    L = lib or load_library()
    keep = []
    a0, a1 = _match_agent(agent0, keep), _match_agent(agent1, keep)
    winners = np.zeros(num_games, np.int8)
    nplies = np.zeros(num_games, np.int32)
    actions = np.full((num_games, max_plies), -1, np.int16)
    rc = L.sprl_match_play(C.byref(cfg), C.byref(a0), C.byref(a1), num_games, winners.ctypes.data, nplies.ctypes.data,
                           actions.ctypes.data, max_plies)
    if rc:
        raise SprlError(rc, L.sprl_last_error().decode())
    return winners, actions, nplies


def match_score(winners):
    """Per-agent tallies from colour winners: (agent0 wins, agent1 wins, draws) — Evaluate.cpp:141-160."""
    w = np.asarray(winners)
    g = np.arange(w.size)
    a0 = int(np.sum((w >= 0) & ((w ^ (g & 1)) == 0)))
    a1 = int(np.sum((w >= 0) & ((w ^ (g & 1)) == 1)))
    return a0, a1, int(np.sum(w < 0))


def profile_busy(lib, kind):
    """(busy_ms, sum_ms) of sprl_profile_busy: kind 0 = tree kernel, 1 = trunk convolution; (None, None) when unavailable."""
    busy, total = C.c_double(0.0), C.c_double(0.0)
    if lib.sprl_profile_busy(kind, C.byref(busy), C.byref(total)) != 0:
        return None, None
    return busy.value, total.value
