"""Multi-GPU layer: reference positions shard across the GPUs of one node, one process per GPU.

The per-site tables of the reference are independent across reference positions and contigs
(parse_bam loops per contig, /root/reference/kindel/kindel.py:150-151), so rank r owns a
contiguous interval of "G-space" (all contigs laid back to back, see include/kindel_hip.h) and
commits only table increments that land in its interval (+ one halo site for
aligned_depth_next, kindel.py:405-410).  There is NO collective on the pileup path: a read that
straddles a boundary is simply given to both owners.  The only exchange is the stitch of the
per-rank consensus pieces: ONE all-gather of fixed-size payloads (header + offsets + depth min/max +
change codes + consensus bytes; the row size follows from the shard geometry, no size exchange) --
RCCL over xGMI when the backend is "nccl", gloo in the CPU tests.  Payload is <= (sites + inserted bases) bytes in
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


def partition_weighted(contig_lens, contig, pos0, weight, world, align=None, snap=0.1):
    """Split G-space into `world` CONTIGUOUS intervals of (nearly) equal WORK instead of equal sites: `weight` per read
    (e.g. its aligned length), credited to the bin of `align` sites its start falls in.  This is the balanced assignment of
    SURVEY 8e for both shapes of input: many contigs (a cut that comes within `snap` x the mean interval of a contig
    boundary is moved onto it, so whole contigs go to one rank: config 4, "greedy by sum of events") and one long contig
    (config 3, position intervals).  Deterministic in its inputs: every rank computes the same cuts from the same batch.
    contig / pos0 / weight: numpy or torch (moved to the host).  -> [(lo, hi)] * world"""
    def host(a):
        return a.detach().cpu().numpy() if type(a).__module__.startswith("torch") else np.asarray(a)
    base, S = g_layout(contig_lens)
    if world <= 1:
        return [(0, S)]
    if align is None:   # bins of 64 .. 2048 sites, a few hundred per rank
        align = 64
        while align < 2048 and S // (2 * align) >= world * 256:
            align *= 2
    c, p, w = host(contig).astype(np.int64), host(pos0).astype(np.int64), host(weight).astype(np.float64)
    g0 = base.astype(np.int64)[c] + np.maximum(p, 0)
    nb = (S + align - 1) // align
    binw = np.bincount(np.minimum(g0 // align, nb - 1), weights=w, minlength=nb)
    cum = np.cumsum(binw)
    total = float(cum[-1]) if len(cum) else 0.0
    bases = base.astype(np.int64)
    cuts = [0]
    for r in range(1, world):
        if total <= 0:
            cut = int(round(S * r / world / align)) * align
        else:
            cut = int(np.searchsorted(cum, total * r / world, side="left") + 1) * align
        near = bases[np.argmin(np.abs(bases - cut))] if len(bases) else cut
        if abs(int(near) - cut) <= snap * S / world:
            cut = int(near)            # whole contigs on one rank where that costs little balance
        cuts.append(min(max(cut, cuts[-1]), S))
    cuts.append(S)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def footprints(contig_lens, batch):
    """Exact reference footprint of every read of a batch, from its CIGAR: -> (g_lo, g_hi) int64 arrays (numpy or torch,
    like the batch), G-space, g_hi EXCLUSIVE -- every table slot the record loop can touch for the read lies in
    [g_lo, g_hi).  Follows parse_records (/root/reference/kindel/kindel.py:40-81): M/=/X and D advance r (:49-54, :59-62),
    a non-first S advances r while r < L (:74-81), a leading S writes the `len` sites in front of pos (:63-73), I / clip
    counters touch the slot AT r (:55-58, :66, :75; hence +1), H / N / P move nothing.  A read placed at pos 0 (r = -1: Python's
    negative indices wrap, :42) may touch any slot of its contig and gets the whole contig.  This is the routing key of the
    multi-GPU layer -- the same footprint k_prep derives on the device (KdRInfo.gstart / span / lead) -- not a guess from
    the query length: a read with a 2 kb deletion reaches 2 kb further than its bases."""
    tor = type(batch["contig"]).__module__.startswith("torch")
    base, _ = g_layout(contig_lens)
    if tor:
        import torch
        dev = batch["contig"].device
        I64 = torch.int64
        lens = torch.as_tensor(np.asarray(contig_lens, np.int64), device=dev)
        b = torch.as_tensor(base.astype(np.int64), device=dev)
        c = batch["contig"].to(I64); pos = batch["pos0"].to(I64); nc = batch["n_cig"].to(I64); co = batch["cig_off"].to(I64)
        n = int(c.numel())
        rid = torch.repeat_interleave(torch.arange(n, device=dev), nc)
        first = torch.cumsum(nc, 0) - nc                      # index of each read's first op in the gathered op list
        k = torch.arange(int(rid.numel()), device=dev) - first[rid]
        w = batch["cigar"].to(I64)[co[rid] + k]
        ln, op = w >> 4, w & 15
        adv = torch.where((op == 0) | (op == 2) | (op == 7) | (op == 8) | ((op == 4) & (k > 0)), ln, torch.zeros_like(ln))
        span = torch.zeros(n, dtype=I64, device=dev).index_add_(0, rid, adv)
        lead = torch.zeros(n, dtype=I64, device=dev).index_add_(0, rid, torch.where((op == 4) & (k == 0), ln, torch.zeros_like(ln)))
        L = lens[c]
        lo = torch.clamp(pos - lead, min=0)
        hi = torch.minimum(torch.clamp(pos, min=0) + span + 1, L + 1)
        wrap = pos < 0
        lo = torch.where(wrap, torch.zeros_like(lo), lo)
        hi = torch.where(wrap, L + 1, hi)
        return b[c] + lo, b[c] + hi
    lens = np.asarray(contig_lens, np.int64)
    b = base.astype(np.int64)
    c = np.asarray(batch["contig"], np.int64); pos = np.asarray(batch["pos0"], np.int64)
    nc = np.asarray(batch["n_cig"], np.int64); co = np.asarray(batch["cig_off"], np.int64)
    n = len(c)
    rid = np.repeat(np.arange(n), nc)
    first = np.cumsum(nc) - nc
    k = np.arange(len(rid)) - first[rid]
    w = np.asarray(batch["cigar"]).astype(np.int64)[co[rid] + k] if len(rid) else np.zeros(0, np.int64)
    ln, op = w >> 4, w & 15
    adv = np.where((op == 0) | (op == 2) | (op == 7) | (op == 8) | ((op == 4) & (k > 0)), ln, 0)
    span = np.bincount(rid, weights=adv, minlength=n).astype(np.int64) if n else np.zeros(0, np.int64)
    lead = np.bincount(rid, weights=np.where((op == 4) & (k == 0), ln, 0), minlength=n).astype(np.int64) if n else np.zeros(0, np.int64)
    L = lens[c]
    lo = np.maximum(pos - lead, 0)
    hi = np.minimum(np.maximum(pos, 0) + span + 1, L + 1)
    wrap = pos < 0
    lo = np.where(wrap, 0, lo)
    hi = np.where(wrap, L + 1, hi)
    return b[c] + lo, b[c] + hi


def reads_touching(contig_lens, contig, pos0, pos_end, rank, world, margin=512, intervals=None):
    """Mask of the reads whose CALLER-SUPPLIED bounds [pos0 - margin, pos_end + margin] touch rank's interval.  Only for
    producers that know their own reads' reach (the synthetic generator, which draws the deletion lengths itself); routing of
    decoded input goes through footprints() + reads_of_rank()."""
    base, _ = g_layout(contig_lens)
    lo, hi = (intervals if intervals is not None else partition(contig_lens, world))[rank]
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


def reads_of_rank(contig_lens, g_lo, g_hi, rank, world, intervals=None):
    """Boolean mask (numpy or torch, like the inputs) of the reads `rank` must see: those whose footprint [g_lo, g_hi)
    (footprints()) touches its commit range [lo, hi] (hi = the halo site for aligned_depth_next, kindel.py:405-410)."""
    lo, hi = (intervals if intervals is not None else partition(contig_lens, world))[rank]
    return (g_hi > lo) & (g_lo <= hi)


def owned_mask(contig_lens, contig, pos0, rank, world, intervals=None):
    """Reads whose start lies in rank's interval (each read is owned by exactly one rank)."""
    base, _ = g_layout(contig_lens)
    lo, hi = (intervals if intervals is not None else partition(contig_lens, world))[rank]
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


_HDR = 16   # payload header: u64 payload bytes, u64 spare


def gather(engine, interval, device, group=None, pad=None, intervals=None):
    """The exchange step: ONE all-gather of fixed-size payloads -- header (payload bytes) + contig offsets + depth min/max +
    change codes + consensus bytes; RCCL over xGMI when the backend is "nccl".  The row size `pad` is agreed without
    communication: every rank derives the same upper bound from the shard geometry (the largest interval of `intervals`,
    or of the equal-sites split when none are given, twice: change codes + one byte per site, plus room for inserted bases).  Should a rank's consensus not fit (an insertion-heavy
    shard), its header says so, every rank reads that in the gathered rows and all of them repeat the gather with the
    announced size -- a second collective only in that case.  Call after engine.consensus_run(); everything stays on
    `device`.  -> (gathered uint8 tensor [world, pad], world)."""
    import torch
    import torch.distributed as dist

    lo, hi = interval
    coff, mm = engine.consensus_offsets()
    cptr, cbytes = engine.consensus_device()
    chptr = engine.changes_device()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    head = np.concatenate([coff.view(np.uint8), mm.reshape(-1).view(np.uint8)])
    my_size = _HDR + head.size + (hi - lo) + cbytes
    if pad is None:
        if intervals is not None:      # every rank holds the same list: the widest interval bounds every rank's payload
            widest = max(1, max(b - a for a, b in intervals))
        else:
            widest = max(1, -(-engine.total_sites() // max(world, 1)) + 4096) if world > 1 else hi - lo
        pad = _HDR + head.size + 2 * widest + widest // 4 + 65536
        pad = max(pad, my_size) if world == 1 else pad
    for _ in range(2):
        payload = torch.zeros(pad, dtype=torch.uint8, device=device)
        hdr = np.asarray([my_size, 0], np.uint64).view(np.uint8)
        payload[:_HDR] = torch.from_numpy(hdr).to(device)
        if my_size <= pad:
            o = _HDR
            payload[o: o + head.size] = torch.from_numpy(head).to(device)
            o += head.size
            payload[o: o + (hi - lo)] = _as_tensor(chptr + lo, hi - lo, device)
            o += hi - lo
            payload[o: o + cbytes] = _as_tensor(cptr, cbytes, device)
        if world > 1:
            gathered = torch.empty(world * pad, dtype=torch.uint8, device=device)
            dist.all_gather_into_tensor(gathered, payload, group=group)  # the one data collective
        else:
            gathered = payload
        rows = gathered.view(world, pad)
        need = int(rows[:, :8].contiguous().view(torch.int64).max().item())   # every rank sees the same sizes
        if need <= pad:
            return rows, world
        pad = need
    raise RuntimeError("shard.gather: payload sizes changed between two gathers")


def assemble(rows, contig_lens, world, interval=None, intervals=None):
    """Host side: per-rank payload rows (uint8 numpy [world, pad]) -> (seqs, changes, minmax) per contig.
    intervals: the per-rank G-space intervals (default: partition(contig_lens, world))."""
    lens = np.asarray(contig_lens, np.uint32)
    n = len(lens)
    base, S = g_layout(lens)
    if intervals is not None:
        ivs = intervals
    else:   # every rank lays contigs out identically, so the equal-sites intervals are recomputable locally
        ivs = partition(lens, world) if world > 1 else [interval if interval is not None else (0, S)]
    seq_parts = [[] for _ in range(n)]
    changes_g = np.zeros(S, np.uint8)
    mins = np.full(n, 0xFFFFFFFF, np.uint64)
    maxs = np.zeros(n, np.uint64)
    for r in range(world):
        row = rows[r][_HDR:]
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


def stitch(engine, interval, device, group=None, intervals=None, pad=None):
    """gather() + host assembly: -> (seqs, changes, minmax), identical on every rank.
    seqs[c] = bytes of contig c's consensus, changes[c] = uint8[L_c], minmax[c] = (min, max) ACGT depth."""
    gathered, world = gather(engine, interval, device, group, pad=pad, intervals=intervals)
    rows = np.ascontiguousarray(gathered.cpu().numpy())
    return assemble(rows, engine.contig_lens, world, interval, intervals=intervals)
