"""sprl_math.h (deterministic log/exp/pow used by the device kernels) vs libm: <= 1 float ulp, and the
two oracle math modes differ by nothing else."""
import ctypes as C

import numpy as np

from oracle import pyoracle as po


def test_dirichlet_portable_close_to_libm():
    L = po.lib()
    worst = 0
    for alpha, k in ((0.3, 10), (0.5, 7), (0.2, 50), (1.0, 5)):
        for stream in range(1, 40):
            a, b = po.RNG(), po.RNG()
            L.orc_rng_seed(C.byref(a), 99, stream)
            L.orc_rng_seed(C.byref(b), 99, stream)
            va, vb = np.zeros(k, np.float32), np.zeros(k, np.float32)
            L.orc_dirichlet(C.byref(a), alpha, k, po.vp(va), po.MATH_LIBM)
            L.orc_dirichlet(C.byref(b), alpha, k, po.vp(vb), po.MATH_PORTABLE)
            if a.state != b.state:      # a 1-ulp logf difference may flip a rejection test (rare)
                continue
            d = np.abs(va.view(np.int32).astype(np.int64) - vb.view(np.int32).astype(np.int64)).max()
            worst = max(worst, int(d))
            assert abs(va.sum() - 1) < 1e-5 and abs(vb.sum() - 1) < 1e-5
    assert worst <= 8                   # a few ulp after sum/normalise


def test_portable_games_are_valid_and_deterministic():
    cfg = po.make_config(po.GAME_OTHELLO, 40, math_mode=po.MATH_PORTABLE)
    a = po.selfplay(cfg, 2, 7, 1, True)
    b = po.selfplay(cfg, 2, 7, 1, True)
    assert (a["dists"].view(np.uint32) == b["dists"].view(np.uint32)).all()
    assert np.allclose(a["dists"].sum(1), 1.0, atol=1e-5)
    assert set(np.unique(a["outcomes"]).tolist()) <= {-1.0, 0.0, 1.0}
