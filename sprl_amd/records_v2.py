"""Compact on-disk record format — SURVEY §8(f) rank 4 (behind a flag; the v1 `.npy` triple stays the default).

v1 (the reference, cpp/src/selfplay/GridWorker.hpp:146-196) stores every sample expanded: nsym symmetric copies x
(2H+1) float planes + float pdf + float outcome = 1 032 B per Othello sample, 8 256 B per ply; 26 000 B per Go 19x19
sample.  v2 stores one entry per ply — the board as two bit sets of ceil(cells/64) words, the mover, the tempered pdf
and the game's winner — and applies the symmetries, the history and the plane encoding when the file is loaded:
16 + 1 + 4A bytes per Othello ply (277 B, ~30x smaller), 96 + 1 + 4*362 B per Go 19x19 ply (~135x smaller).  The body
is exactly what ranks send to rank 0 over RCCL (sprl_amd/distributed.py), any board size.
"""
import numpy as np

from .distributed import pack_records, unpack_records

MAGIC = b"SPRLv2\x01\x00"          # \x01: wire format with a 12-word header and 16-byte aligned sections

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
    """Write one run's SelfPlayRecords in the compact form (temp file + rename, like the v1 writer)."""
    payload = pack_records(rec)
    tmp = str(path) + ".tmp"
    with open(tmp, "wb") as f:
        f.write(MAGIC)
        f.write(np.array([payload.size], np.int64).tobytes())
        f.write(payload.tobytes())
    import os
    os.replace(tmp, path)


def expand_unpacked(u, use_symmetry=None):
    """Unpacked shard (distributed.unpack_records) -> the reference's training arrays (states[N,2H+1,R,C],
    distributions[N,A], outcomes[N]) in the reference's sample order (game-major, ply-major, symmetry-minor)."""
    rows, cols, cells, A, H = u["rows"], u["cols"], u["cells"], u["actions"], u["history"]
    sym = u["use_symmetry"] if use_symmetry is None else bool(use_symmetry)
    nsym = u["nsym"] if sym else 1
    maps = _cell_maps(rows, cols, u["nsym"])[:nsym]
    if A == cells + 1:                      # board games with a pass action: the pass index is fixed
        amaps = np.stack([np.concatenate([m, [cells]]) for m in maps])
    else:                                   # Connect Four: actions are columns (ConnectFourSymmetrizer.cpp:66-100)
        amaps = np.stack([np.arange(A) if s == 0 else np.arange(A)[::-1].copy() for s in range(nsym)])
    n = u["total_plies"]
    boards, movers, pdfs, offs = u["boards"], u["movers"], u["pdfs"], u["ply_offset"]
    game_of = np.repeat(np.arange(u["num_games"]), np.diff(offs))
    start = offs[game_of]                                         # first ply of each ply's game
    w = u["winners"][game_of]
    reward = np.where(w < 0, 0.0, np.where(w == movers, 1.0, -1.0)).astype(np.float32)
    states = np.zeros((n, nsym, 2 * H + 1, cells), np.float32)
    dists = np.zeros((n, nsym, A), np.float32)
    ply = np.arange(n)
    for t in range(H):                                            # history ply t of every sample at once
        ok = ply - t >= start
        src = boards[np.where(ok, ply - t, ply)]
        own = ((src == movers[:, None]) & ok[:, None]).astype(np.float32)
        opp = ((src >= 0) & (src != movers[:, None]) & ok[:, None]).astype(np.float32)
        for s in range(nsym):
            states[:, s, 2 * t, maps[s]] = own
            states[:, s, 2 * t + 1, maps[s]] = opp
    states[:, :, 2 * H, :] = (movers == 0).astype(np.float32)[:, None, None]
    for s in range(nsym):
        dists[:, s, amaps[s]] = pdfs
    outcomes = np.repeat(reward, nsym)
    return states.reshape(n * nsym, 2 * H + 1, rows, cols), dists.reshape(n * nsym, A), outcomes


def load_compact(path, use_symmetry=None):
    """Read a v2 file and expand it to the reference's training arrays."""
    raw = open(path, "rb").read()
    if raw[:8] != MAGIC:
        raise ValueError("not a sprl v2 record file")
    size = int(np.frombuffer(raw[8:16], np.int64)[0])
    return expand_unpacked(unpack_records(np.frombuffer(raw[16:16 + size], np.uint8)), use_symmetry)
