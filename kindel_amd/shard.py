"""Multi-GPU layer: reference positions shard across the GPUs of one node, one process per GPU.

The per-site tables of the reference are independent across reference positions and contigs
(parse_bam loops per contig, /root/reference/kindel/kindel.py:150-151), so rank r owns a
contiguous interval of "G-space" (all contigs laid back to back, see include/kindel_hip.h) and
commits only table increments that land in its interval (+ one halo site for
aligned_depth_next, kindel.py:405-410).  There is NO collective on the pileup path: a read that
straddles a boundary is simply given to both owners.  The only exchange is the stitch of the
per-rank consensus pieces: one tiny all-gather of payload sizes followed by ONE all-gather of the
padded payload (offsets + depth min/max + change codes + consensus bytes) -- RCCL over xGMI when
the backend is "nccl", gloo in the CPU tests.  Payload is <= (sites + inserted bases) bytes in
total, i.e. latency bound; sharding by reads instead would need a 76 B/site table all-reduce.
"""
import ctypes as C

import numpy as np


def g_layout(contig_lens):
    """-> (base[n] uint64, S) exactly as kd_create lays contigs out (len+1 slots, 64-padded; S 1024-padded)"""
    lens = np.asarray(contig_lens, np.uint64)
    padded = (lens + np.uint64(1) + np.uint64(63)) // np.uint64(64) * np.uint64(64)
    base = np.concatenate([[0], np.cumsum(padded)[:-1]]).astype(np.uint64)
    total = int(padded.sum())
    S = (total + 1023) // 1024 * 1024
    return base, S


def partition(contig_lens, world):
    """Equal split of G-space into `world` intervals [lo, hi), cut points aligned to 2048 sites."""
    _, S = g_layout(contig_lens)
    cuts = [0]
    for r in range(1, world):
        c = int(round(S * r / world / 2048.0)) * 2048
        cuts.append(min(max(c, cuts[-1]), S))
    cuts.append(S)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def reads_of_rank(contig_lens, contig, pos0, pos_end, rank, world, margin=512):
    """Boolean mask (same array type as the inputs: numpy or torch) of the reads rank must see:
    every read whose reference footprint [pos0 - margin, pos_end + margin] touches its interval."""
    base, _ = g_layout(contig_lens)
    lo, hi = partition(contig_lens, world)[rank]
    if type(contig).__module__.startswith("torch"):
        import torch
        b = torch.as_tensor(base.astype(np.int64), device=contig.device)
        g0 = b[contig.long()] + pos0.long()
        g1 = b[contig.long()] + pos_end.long()
    else:
        b = base.astype(np.int64)
        g0 = b[np.asarray(contig, np.int64)] + np.asarray(pos0, np.int64)
        g1 = b[np.asarray(contig, np.int64)] + np.asarray(pos_end, np.int64)
    return (g1 + margin >= lo) & (g0 - margin <= hi)


def owned_mask(contig_lens, contig, pos0, rank, world):
    """Reads whose start lies in rank's interval (each read is owned by exactly one rank)."""
    base, _ = g_layout(contig_lens)
    lo, hi = partition(contig_lens, world)[rank]
    if type(contig).__module__.startswith("torch"):
        import torch
        b = torch.as_tensor(base.astype(np.int64), device=contig.device)
        g0 = b[contig.long()] + pos0.long().clamp(min=0)
    else:
        g0 = base.astype(np.int64)[np.asarray(contig, np.int64)] + np.maximum(np.asarray(pos0, np.int64), 0)
    return (g0 >= lo) & (g0 < hi)


class _DevArray:
    """Expose a raw device pointer to torch through the CUDA array interface (ROCm builds honour it)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = dict(shape=(n,), typestr="|u1", data=(int(ptr), False), version=2)


def _as_tensor(ptr, n, device):
    import torch
    if n == 0:
        return torch.zeros(0, dtype=torch.uint8, device=device)
    if torch.device(device).type == "cpu":
        buf = (C.c_uint8 * n).from_address(ptr)
        return torch.from_numpy(np.frombuffer(buf, np.uint8, n))
    return torch.as_tensor(_DevArray(ptr, n), device=device)


def gather(engine, interval, device, group=None):
    """The exchange step: all-gather the per-rank payloads (offsets + depth min/max + change codes +
    consensus bytes).  Call after engine.consensus_run().  Everything stays on `device`:
    -> (gathered uint8 tensor [world, pad], world).  One tiny all-gather of sizes, ONE data all-gather."""
    import torch
    import torch.distributed as dist

    lens = engine.contig_lens
    lo, hi = interval
    coff, mm = engine.consensus_offsets()
    cptr, cbytes = engine.consensus_device()
    chptr = engine.changes_device()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    head = np.concatenate([coff.view(np.uint8), mm.reshape(-1).view(np.uint8)])
    my_size = head.size + (hi - lo) + cbytes
    sizes = torch.zeros(world, dtype=torch.int64, device=device)
    if world > 1:
        dist.all_gather_into_tensor(sizes, torch.tensor([my_size], dtype=torch.int64, device=device), group=group)
        pad = int(sizes.max().item())
    else:
        pad = my_size
    payload = torch.zeros(pad, dtype=torch.uint8, device=device)
    payload[: head.size] = torch.from_numpy(head).to(device)
    payload[head.size: head.size + (hi - lo)] = _as_tensor(chptr + lo, hi - lo, device)
    payload[head.size + (hi - lo): my_size] = _as_tensor(cptr, cbytes, device)
    if world > 1:
        gathered = torch.empty(world * pad, dtype=torch.uint8, device=device)
        dist.all_gather_into_tensor(gathered, payload, group=group)  # the one data collective
    else:
        gathered = payload
    return gathered.view(world, pad), world


def assemble(rows, contig_lens, world, interval=None):
    """Host side: per-rank payload rows (uint8 numpy [world, pad]) -> (seqs, changes, minmax) per contig."""
    lens = np.asarray(contig_lens, np.uint32)
    n = len(lens)
    base, S = g_layout(lens)
    # every rank lays contigs out identically, so intervals are recomputable locally
    ivs = partition(lens, world) if world > 1 else [interval if interval is not None else (0, S)]
    seq_parts = [[] for _ in range(n)]
    changes_g = np.zeros(S, np.uint8)
    mins = np.full(n, 0xFFFFFFFF, np.uint64)
    maxs = np.zeros(n, np.uint64)
    for r in range(world):
        row = rows[r]
        rcoff = row[: (n + 1) * 8].view(np.uint64)
        rmm = row[(n + 1) * 8: (n + 1) * 8 + n * 8].view(np.uint32).reshape(n, 2)
        rlo, rhi = ivs[r]
        o = (n + 1) * 8 + n * 8
        changes_g[rlo:rhi] = row[o: o + (rhi - rlo)]
        o += rhi - rlo
        for c in range(n):
            if rcoff[c + 1] > rcoff[c]:
                seq_parts[c].append(row[o + int(rcoff[c]): o + int(rcoff[c + 1])].tobytes())
        mins = np.minimum(mins, rmm[:, 0])
        maxs = np.maximum(maxs, rmm[:, 1])
    seqs = [b"".join(p) for p in seq_parts]
    changes = [changes_g[int(base[c]): int(base[c]) + int(lens[c])] for c in range(n)]
    minmax = [(int(mins[c]), int(maxs[c])) for c in range(n)]
    return seqs, changes, minmax


def stitch(engine, interval, device, group=None):
    """gather() + host assembly: -> (seqs, changes, minmax), identical on every rank.
    seqs[c] = bytes of contig c's consensus, changes[c] = uint8[L_c], minmax[c] = (min, max) ACGT depth."""
    gathered, world = gather(engine, interval, device, group)
    rows = np.ascontiguousarray(gathered.cpu().numpy())
    return assemble(rows, engine.contig_lens, world, interval)
