"""Compact on-disk record format — SURVEY §8(f) rank 4 (behind a flag; the v1 `.npy` triple stays the default).

v1 (the reference, cpp/src/selfplay/GridWorker.hpp:146-196) stores every sample expanded: nsym symmetric copies x
(2H+1) float planes + float pdf + float outcome = 1 032 B per Othello sample, 8 256 B per ply.  v2 stores one entry per
ply — the board as two bit planes, the mover, the tempered pdf and the game's winner — and applies the symmetries and
the plane encoding when the file is loaded: 8 + 8 + 1 + 4A bytes per ply (277 B for Othello, ~30x smaller), which is
also exactly what ranks send to rank 0 over RCCL (sprl_amd/distributed.py).
"""
import numpy as np

from .distributed import pack_records, unpack_records

MAGIC = b"SPRLv2\x00\x00"

# D4 maps out[map(r, c)] = in[r, c] (cpp/src/symmetry/D4GridSymmetrizer.hpp:108-117); column mirror for Connect Four
_D4 = [lambda r, c, L: (r, c), lambda r, c, L: (c, L - r), lambda r, c, L: (L - r, L - c), lambda r, c, L: (L - c, r),
       lambda r, c, L: (r, L - c), lambda r, c, L: (L - c, L - r), lambda r, c, L: (L - r, c), lambda r, c, L: (c, r)]


def _cell_maps(rows, cols, nsym):
    maps = np.zeros((nsym, rows * cols), np.int64)
    for s in range(nsym):
        for r in range(rows):
            for c in range(cols):
                if nsym == 2:
                    tr, tc = r, (cols - 1 - c if s == 1 else c)
                else:
                    tr, tc = _D4[s](r, c, cols - 1)
                maps[s, r * cols + c] = tr * cols + tc
    return maps


def write_compact(path, rec):
    """Write one run's SelfPlayRecords in the compact form."""
    payload = pack_records(rec)
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(np.array([payload.size, rec.history], np.int64).tobytes())
        f.write(payload.tobytes())


def load_compact(path, use_symmetry=None):
    """Read a v2 file and expand it to the reference's training arrays (states[N,2H+1,R,C], distributions[N,A],
    outcomes[N]) in the reference's sample order (game-major, ply-major, symmetry-minor)."""
    raw = open(path, "rb").read()
    if raw[:8] != MAGIC:
        raise ValueError("not a sprl v2 record file")
    size, hist = np.frombuffer(raw[8:24], np.int64)
    u = unpack_records(np.frombuffer(raw[24:24 + size], np.uint8))
    rows, cols, cells, A, H = u["rows"], u["cols"], u["cells"], u["actions"], int(hist)
    sym = u["use_symmetry"] if use_symmetry is None else bool(use_symmetry)
    nsym = u["nsym"] if sym else 1
    maps = _cell_maps(rows, cols, u["nsym"])[:nsym]
    if A == cells + 1:                      # board games with a pass action: the pass index is fixed
        amaps = [np.concatenate([m, [cells]]) for m in maps]
    else:                                   # Connect Four: actions are columns (ConnectFourSymmetrizer.cpp:66-100)
        amaps = [np.arange(A) if s == 0 else np.arange(A)[::-1].copy() for s in range(nsym)]
    n = u["total_plies"]
    states = np.zeros((n * nsym, 2 * H + 1, cells), np.float32)
    dists = np.zeros((n * nsym, A), np.float32)
    outcomes = np.zeros(n * nsym, np.float32)
    boards, movers, pdfs, offs = u["boards"], u["movers"], u["pdfs"], u["ply_offset"]
    for g in range(u["num_games"]):
        w = u["winners"][g]
        for p in range(offs[g], offs[g + 1]):
            mover = movers[p]
            reward = 0.0 if w < 0 else (1.0 if w == mover else -1.0)
            for s in range(nsym):
                k = p * nsym + s
                for t in range(H):
                    if p - t < offs[g]:
                        break
                    b = boards[p - t]
                    states[k, 2 * t, maps[s]] = (b == mover)
                    states[k, 2 * t + 1, maps[s]] = (b >= 0) & (b != mover)
                states[k, 2 * H] = 1.0 if mover == 0 else 0.0
                dists[k, amaps[s]] = pdfs[p]
                outcomes[k] = reward
    return states.reshape(n * nsym, 2 * H + 1, rows, cols), dists, outcomes
