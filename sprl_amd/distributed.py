"""Game-sharded multi-GPU self-play: one process per GPU, no traffic during search, one gather of the compact
finished-game records to rank 0 per iteration.  Uses torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests): an all_gather of per-rank byte counts, then a gather of equal-sized
padded uint8 shards.  The reference's equivalent is N independent worker processes and a shared directory
(cpp/src/OTHWorker.cpp:39-49, scripts/othello_controller.py:77-93).

Wire format (also the body of a v2 record file, sprl_amd/records_v2.py) — sections start on 16-byte boundaries:
    head      int64[12]   games, plies, actions, cells, game id, nsym, use_symmetry, rows, cols, words, history, 0
    offsets   int32[games + 1]
    winners   int8[games]
    stones0   uint64[plies][words]    bit c = cell c holds a stone of Player::ZERO   (words = ceil(cells / 64))
    stones1   uint64[plies][words]
    movers    uint8[plies]
    pdfs      float32[plies][actions]
The same bytes are produced on the device by the engine (sprl_engine_pack_records, include/sprl_amd.h) straight
from its record buffers, so ranks hand RCCL a device tensor without a host round trip."""
import numpy as np

HEAD_WORDS = 12


def _align(o):
    return (o + 15) & ~15


def section_offsets(games, plies, actions, words):
    """Byte offsets of the sections of a packed shard and its total size."""
    o = {}
    p = HEAD_WORDS * 8
    for name, size in (("offsets", 4 * (games + 1)), ("winners", games), ("stones0", 8 * words * plies),
                       ("stones1", 8 * words * plies), ("movers", plies), ("pdfs", 4 * plies * actions)):
        o[name] = p
        p = _align(p + size)
    o["total"] = p
    return o


def pack_records(rec):
    """Serialise a SelfPlayRecords into one flat uint8 array (compact form: bit boards, mover, pdf, winner)."""
    g, n, A, cells = rec.num_games, rec.total_plies, rec.actions, rec.cells
    words = (cells + 63) // 64
    padded = np.full((n, words * 64), -1, np.int8)
    padded[:, :cells] = rec.boards
    b0 = np.packbits(padded == 0, axis=1, bitorder="little")          # [n, 8 * words]
    b1 = np.packbits(padded == 1, axis=1, bitorder="little")
    head = np.array([g, n, A, cells, rec.game, rec.nsym, int(rec.use_symmetry), rec.rows, rec.cols, words,
                     int(getattr(rec, "history", 1)), 0], np.int64)
    off = section_offsets(g, n, A, words)
    out = np.zeros(off["total"], np.uint8)
    out[:HEAD_WORDS * 8] = head.view(np.uint8)
    for name, arr in (("offsets", np.ascontiguousarray(rec.ply_offset, np.int32)), ("winners", np.ascontiguousarray(rec.winners, np.int8)),
                      ("stones0", b0), ("stones1", b1), ("movers", np.ascontiguousarray(rec.movers, np.int8)),
                      ("pdfs", np.ascontiguousarray(rec.pdfs, np.float32))):
        raw = arr.reshape(-1).view(np.uint8)
        out[off[name]:off[name] + raw.size] = raw
    return out


def unpack_records(buf):
    """Inverse of pack_records -> dict of numpy arrays (boards as int8 cells)."""
    buf = np.ascontiguousarray(buf, np.uint8)
    head = buf[:HEAD_WORDS * 8].view(np.int64)
    g, n, A, ncells, words = int(head[0]), int(head[1]), int(head[2]), int(head[3]), int(head[9])
    off = section_offsets(g, n, A, words)
    if buf.size < off["total"]:
        raise ValueError(f"packed records truncated: {buf.size} bytes, header describes {off['total']}")

    def sec(name, nbytes):
        return buf[off[name]:off[name] + nbytes]

    ply_offset = sec("offsets", 4 * (g + 1)).view(np.int32).copy()
    winners = sec("winners", g).view(np.int8).copy()
    b0 = np.unpackbits(sec("stones0", 8 * words * n).reshape(n, 8 * words), axis=1, bitorder="little")
    b1 = np.unpackbits(sec("stones1", 8 * words * n).reshape(n, 8 * words), axis=1, bitorder="little")
    movers = sec("movers", n).view(np.int8).copy()
    pdfs = sec("pdfs", 4 * n * A).view(np.float32).reshape(n, A).copy()
    boards = np.full((n, 64 * words), -1, np.int8)
    boards[b0 == 1] = 0
    boards[b1 == 1] = 1
    return dict(num_games=g, total_plies=n, actions=A, cells=ncells, game=int(head[4]), nsym=int(head[5]),
                use_symmetry=bool(head[6]), rows=int(head[7]), cols=int(head[8]), words=words, history=int(head[10]),
                ply_offset=ply_offset, winners=winners, boards=np.ascontiguousarray(boards[:, :ncells]), movers=movers,
                pdfs=pdfs, nbytes=off["total"])


def gather_packed(shard, nbytes, dist, dst=0, unpack=True, to_host=True):
    """Gather every rank's packed shard (a 1-D uint8 torch tensor on the backend's device: CUDA for "nccl" = RCCL,
    CPU for "gloo") to rank `dst`: an all_gather of the byte counts, then one gather of equal-sized padded shards.
    Returns on `dst` the list of unpacked shards (rank order = game-shard order); with unpack=False the raw packed shards as
    HOST uint8 tensors (one device-to-host copy each, no decoding: what a producer that only forwards or writes the bytes
    needs; decode later with unpack_records); with unpack=False and to_host=False the shards as views of the gathered DEVICE
    tensors - nothing leaves the device inside the call (copy them out behind a timed bracket, or on a side stream).
    None on the other ranks."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    size = torch.tensor([int(nbytes)], dtype=torch.int64, device=shard.device)
    sizes_t = torch.zeros(world, dtype=torch.int64, device=shard.device)
    dist.all_gather_into_tensor(sizes_t, size)           # ONE tensor, read back with ONE host synchronisation
    sizes = sizes_t.tolist()
    cap = max(sizes)
    if shard.numel() < cap:                              # pad to the largest shard of this step
        padded = torch.zeros(cap, dtype=torch.uint8, device=shard.device)
        padded[:shard.numel()] = shard
        shard = padded
    else:
        shard = shard[:cap].contiguous()
    outs = [torch.empty_like(shard) for _ in range(world)] if rank == dst else None
    dist.gather(shard, outs, dst=dst)
    if rank != dst:
        return None
    if not unpack and not to_host:
        return [t[:s] for t, s in zip(outs, sizes)]
    raw = [t[:s].cpu() for t, s in zip(outs, sizes)]
    return [unpack_records(t.numpy()) for t in raw] if unpack else raw


def gather_records(rec, dist, device="cpu", dst=0):
    """Host-side records (a SelfPlayRecords already collected) -> packed -> gathered.  The GPU path packs on the
    device instead (Engine.pack_records_device) and calls gather_packed directly."""
    import torch
    payload = pack_records(rec)
    return gather_packed(torch.from_numpy(payload).to(device), payload.size, dist, dst)
