"""Game-sharded multi-GPU self-play: one process per GPU, no traffic during search, one gather of the compact
finished-game records to rank 0 per iteration.  Uses torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests): an all_gather of per-rank byte counts, then a gather of equal-sized
padded uint8 shards.  The reference's equivalent is N independent worker processes and a shared directory
(cpp/src/OTHWorker.cpp:39-49, scripts/othello_controller.py:77-93)."""
import numpy as np


def pack_records(rec):
    """Serialise a SelfPlayRecords into one flat uint8 array (compact form: bitboards, mover, pdf, winner)."""
    g, n, A = rec.num_games, rec.total_plies, rec.actions
    cells = np.zeros((n, 64), np.int8) - 1
    cells[:, :rec.cells] = rec.boards
    b0 = np.packbits(cells == 0, axis=1, bitorder="little")           # [n, 8]
    b1 = np.packbits(cells == 1, axis=1, bitorder="little")
    head = np.array([g, n, A, rec.cells, rec.game, rec.nsym, int(rec.use_symmetry), rec.rows, rec.cols], np.int64)  # 9 x int64
    parts = [head.view(np.uint8), rec.ply_offset.astype(np.int32).view(np.uint8), rec.winners.view(np.uint8),
             b0.reshape(-1), b1.reshape(-1), rec.movers.view(np.uint8),
             np.ascontiguousarray(rec.pdfs, np.float32).view(np.uint8).reshape(-1)]
    return np.concatenate(parts)


def unpack_records(buf):
    """Inverse of pack_records -> dict of numpy arrays (boards as int8 cells)."""
    buf = np.ascontiguousarray(buf, np.uint8)
    head = buf[:72].view(np.int64)
    g, n, A, ncells = int(head[0]), int(head[1]), int(head[2]), int(head[3])
    o = 72
    ply_offset = buf[o:o + 4 * (g + 1)].view(np.int32).copy(); o += 4 * (g + 1)
    winners = buf[o:o + g].view(np.int8).copy(); o += g
    b0 = np.unpackbits(buf[o:o + 8 * n].reshape(n, 8), axis=1, bitorder="little"); o += 8 * n
    b1 = np.unpackbits(buf[o:o + 8 * n].reshape(n, 8), axis=1, bitorder="little"); o += 8 * n
    movers = buf[o:o + n].view(np.int8).copy(); o += n
    pdfs = buf[o:o + 4 * n * A].view(np.float32).reshape(n, A).copy(); o += 4 * n * A
    boards = np.full((n, 64), -1, np.int8)
    boards[b0 == 1] = 0
    boards[b1 == 1] = 1
    return dict(num_games=g, total_plies=n, actions=A, cells=ncells, game=int(head[4]), nsym=int(head[5]),
                use_symmetry=bool(head[6]), rows=int(head[7]), cols=int(head[8]), ply_offset=ply_offset,
                winners=winners, boards=boards[:, :ncells], movers=movers, pdfs=pdfs, nbytes=o)


def gather_records(rec, dist, device="cpu", dst=0):
    """Gather every rank's compact records to rank `dst`.  Returns a list of unpacked shards on `dst` (rank order =
    game-shard order), None elsewhere."""
    import torch
    payload = pack_records(rec)
    world, rank = dist.get_world_size(), dist.get_rank()
    size = torch.tensor([payload.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size)
    cap = int(max(int(s.item()) for s in sizes))
    shard = torch.zeros(cap, dtype=torch.uint8, device=device)
    shard[:payload.size] = torch.from_numpy(payload).to(device)
    outs = [torch.empty_like(shard) for _ in range(world)] if rank == dst else None
    dist.gather(shard, outs, dst=dst)
    if rank != dst:
        return None
    return [unpack_records(t.cpu().numpy()[:int(s.item())]) for t, s in zip(outs, sizes)]
