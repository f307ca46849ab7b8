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


def row_pad(engine, interval, world, intervals=None):
    """The row size every rank derives WITHOUT communication from the shard geometry: the widest interval of `intervals` (or of the
    equal-sites split) twice -- change codes + one byte per site -- plus room for inserted bases (1/16 of the sites + 16 KB: the
    all-gather moves the agreed size whatever a row holds, and a rank with more net insertions than that announces its size:
    gather() / stitch() repeat the collective exactly sized)."""
    n = len(engine.contig_lens)
    head = (n + 1) * 8 + n * 8
    lo, hi = interval
    if intervals is not None:      # every rank holds the same list: the widest interval bounds every rank's payload
        widest = max(1, max(b - a for a, b in intervals))
    else:
        widest = max(1, -(-engine.total_sites() // max(world, 1)) + 4096) if world > 1 else hi - lo
    return -(-(_HDR + head + 2 * widest + widest // 16 + 16384) // 64) * 64


class Exchange:
    """The exchange step with its buffers kept across steps.  The ENGINE writes this rank's row -- header (row bytes) + contig
    offsets + depth min/max + change codes + consensus bytes, kindel_hip.h: kd_exchange_row -- into `row` (device memory): on
    demand (run()), or on the way of kd_finish / kd_step once attach()ed, so that a multi-GPU step is kd_step + ONE collective
    (RCCL over xGMI when the backend is "nccl") and no torch kernel, upload or read-back besides.  Everything stays on `device`."""

    def __init__(self, engine, interval, device, group=None, pad=None, intervals=None, force_collective=False):
        import torch
        import torch.distributed as dist
        self.engine, self.interval, self.device, self.group = engine, interval, device, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force_collective: run the all-gather even in a group of ONE rank (bench.py --rccl-at-1, the GPU suite: the RCCL call on a
        # device row on the one GPU a test box has)
        self.collective = self.world > 1 or (bool(force_collective) and dist.is_initialized())
        self.pad = int(pad) if pad is not None else row_pad(engine, interval, self.world, intervals)
        # the row lives where the ENGINE's memory is (it writes it with device-to-device copies and a kernel); the collective runs on
        # `device`: the same memory under RCCL, host tensors under gloo (a CLI run with KINDEL_DIST_BACKEND=gloo, the CPU tests) --
        # then the row is copied over in front of the collective
        mem = engine.memory_device
        self.row = torch.empty(self.pad, dtype=torch.uint8, device=mem)
        self.staged = self.collective and torch.empty(0, device=mem).device != torch.empty(0, device=device).device
        self.row_coll = torch.empty(self.pad, dtype=torch.uint8, device=device) if self.staged else self.row
        self.rows = torch.empty(self.world * self.pad, dtype=torch.uint8, device=device) if self.collective else self.row
        self.attached = False
        self._cuda = self.collective and torch.device(device).type == "cuda"

    def attach(self):
        """From now on kd_finish / kd_step (Engine.finish / step_device) leave the row behind: collect() is the collective alone."""
        self.engine.set_exchange(self.row.data_ptr(), self.pad)
        self.engine._exchange_keep = self.row      # (the context holds the raw pointer: the tensor lives as long as the registration)
        self.attached = True
        return self

    def detach(self):
        if self.attached:
            self.engine.set_exchange(0, 0)
            self.engine._exchange_keep = None
            self.attached = False

    def collect(self):
        """The ONE data collective over rows the engines have already written -> uint8 tensor [world, pad] (on `device`)."""
        import torch
        import torch.distributed as dist
        if self.collective:
            if self.staged:
                self.row_coll.copy_(self.row)
            dist.all_gather_into_tensor(self.rows, self.row_coll, group=self.group)
            if self._cuda:
                torch.cuda.current_stream().synchronize()     # the step ends when every rank's consensus is in this GPU's HBM
        return self.rows.view(self.world, self.pad)

    def run(self):
        """Row on demand (after Engine.consensus_run / finish) + the collective."""
        self.engine.exchange_row(self.row.data_ptr(), self.pad)
        return self.collect()

    def need(self, rows):
        """The largest row any rank announced (every rank reads the same headers: the same decision everywhere)."""
        import torch
        return int(rows[:, :8].contiguous().view(torch.int64).max().item())


def gather(engine, interval, device, group=None, pad=None, intervals=None):
    """The exchange step, one-shot: ONE all-gather of fixed-size rows (Exchange).  The row size `pad` is agreed without
    communication (row_pad).  Should a rank's consensus not fit (an insertion-heavy shard), its header says so, every rank
    reads that in the gathered rows and all of them repeat the gather with the announced size -- a second collective only
    in that case.  Call after engine.consensus_run() / finish(); everything stays on `device`.
    -> (gathered uint8 tensor [world, pad], world)."""
    lo, hi = interval
    if (lo, hi) != tuple(engine.shard_interval()):
        raise ValueError("shard.gather: interval %r is not the engine's shard %r" % ((lo, hi), tuple(engine.shard_interval())))
    for _ in range(2):
        ex = Exchange(engine, interval, device, group=group, pad=pad, intervals=intervals)
        rows = ex.run()
        need = ex.need(rows)
        if need <= ex.pad:
            return rows, ex.world
        pad = need
    raise RuntimeError("shard.gather: payload sizes changed between two gathers")


def assemble(rows, contig_lens, world, interval=None, intervals=None, with_parts=False):
    """Host side: per-rank payload rows (uint8 numpy [world, pad]) -> (seqs, changes, minmax) per contig.
    intervals: the per-rank G-space intervals (default: partition(contig_lens, world)).
    with_parts: a fourth result, part_lens[c][r] = bytes rank r contributed to contig c (realign splices its patches by them)."""
    lens = np.asarray(contig_lens, np.uint32)
    n = len(lens)
    base, S = g_layout(lens)
    if intervals is not None:
        ivs = intervals
    else:   # every rank lays contigs out identically, so the equal-sites intervals are recomputable locally
        ivs = partition(lens, world) if world > 1 else [interval if interval is not None else (0, S)]
    seq_parts = [[] for _ in range(n)]
    part_lens = [[0] * world for _ in range(n)]
    changes_g = np.zeros(S, np.uint8)
    mins = np.full(n, 0xFFFFFFFF, np.uint64)
    maxs = np.zeros(n, np.uint64)
    for r in range(world):
        announced = int(np.asarray(rows[r][:8]).view(np.uint64)[0])
        if announced > len(rows[r]):      # (gather() / stitch() never hand such rows over: they repeat the collective exactly sized)
            raise ValueError("shard.assemble: rank %d's row did not fit (%d bytes announced, rows of %d)" % (r, announced, len(rows[r])))
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
                part_lens[c][r] = int(rcoff[c + 1]) - int(rcoff[c])
        mins = np.minimum(mins, rmm[:, 0])
        maxs = np.maximum(maxs, rmm[:, 1])
    seqs = [b"".join(p) for p in seq_parts]
    changes = [changes_g[int(base[c]): int(base[c]) + int(lens[c])] for c in range(n)]
    minmax = [(int(mins[c]), int(maxs[c])) for c in range(n)]
    if with_parts:
        return seqs, changes, minmax, part_lens
    return seqs, changes, minmax


def stitch(engine, interval, device, group=None, intervals=None, pad=None, with_parts=False):
    """gather() + host assembly: -> (seqs, changes, minmax[, part_lens]), identical on every rank.
    seqs[c] = bytes of contig c's consensus, changes[c] = uint8[L_c], minmax[c] = (min, max) ACGT depth."""
    gathered, world = gather(engine, interval, device, group, pad=pad, intervals=intervals)
    rows = np.ascontiguousarray(gathered.cpu().numpy())
    return assemble(rows, engine.contig_lens, world, interval, intervals=intervals, with_parts=with_parts)


# ---------------------------------------------------------------------------------------------------------------------
# Product entry: ONE file, N ranks (north_star: "reference positions shard by contig/interval across the 8 GPUs of one node
# with a single RCCL all-gather to stitch the final FASTA").  Ingest is sharded too -- end to end the host decode is the
# long pole, so N GPUs only help if every rank decodes 1/N of the file:
#   * rank r decodes the records that begin in its byte share of the BGZF blocks (kd_decode_open_span; boundaries are a
#     function of the file alone, verified through the record chain);
#   * ONE small all-gather of per-rank facts (first / last start key, sortedness, longest footprint, span offsets): in a
#     coordinate-sorted file the shares are ordered by position, so rank r OWNS the sites from its first read's start to
#     the next rank's -- work-balanced by construction (equal bytes = equal reads);
#   * reads that reach into the interval from the neighbouring shares (at most the longest footprint before it, the longest
#     leading clip behind it) are decoded by the rank itself from the few blocks either side -- no read exchange;
#   * pileup + consensus of the interval (kd_set_shard: shard-local tables, no collective), then the ONE all-gather of
#     stitch().
# Anything that breaks the premises (not BGZF, not sorted, a span whose chain does not close) is agreed on in that first
# all-gather and every rank reads the whole file instead -- same results, no ingest speed-up.  This is what
# parse_bam's loop over the records (/root/reference/kindel/kindel.py:143-151) becomes across ranks.
# ---------------------------------------------------------------------------------------------------------------------
_REC_FIELDS = ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig")


def concat_batches(parts):
    """Host batches (dicts of numpy arrays over the same contig table) -> one batch, offsets rebased, order kept."""
    parts = [p for p in parts if len(p["contig"])]
    if not parts:
        return None
    if len(parts) == 1:
        return parts[0]
    out = {k: np.concatenate([np.asarray(p[k]) for p in parts]) for k in ("contig", "pos0", "flag", "seq_len", "n_cig")}
    so, co, sq, cg = [], [], [], []
    s_at = c_at = 0
    for p in parts:
        so.append(np.asarray(p["seq_off"], np.uint64) + np.uint64(s_at))
        co.append(np.asarray(p["cig_off"], np.uint64) + np.uint64(c_at))
        sq.append(np.asarray(p["seq4"])); cg.append(np.asarray(p["cigar"]))
        s_at += len(p["seq4"]); c_at += len(p["cigar"])
    out["seq_off"], out["cig_off"] = np.concatenate(so), np.concatenate(co)
    out["seq4"], out["cigar"] = np.concatenate(sq), np.concatenate(cg)
    out["contig_names"], out["contig_lens"] = parts[0]["contig_names"], parts[0]["contig_lens"]
    return out


def _subset(batch, keep):
    out = dict(batch)
    for k in _REC_FIELDS:
        out[k] = np.asarray(batch[k])[keep]
    return out


def _start_keys(base, batch):
    return base.astype(np.int64)[np.asarray(batch["contig"], np.int64)] + np.maximum(np.asarray(batch["pos0"], np.int64), 0)


def ingest_sharded(path, rank, world, device="cpu", group=None, threads=0, lib=None):
    """This rank's reads of `path` and its G-space interval.
    -> dict(batch (host, may be None), interval, intervals, names, lens, order (contig ids, first appearance), mode, stats)"""
    import torch
    import torch.distributed as dist
    from . import _native as N

    def gather_rows(vals):
        t = torch.tensor(vals, dtype=torch.int64, device=device)
        if world == 1:
            return t.cpu().numpy().reshape(1, -1)
        out = torch.empty(world * t.numel(), dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(out, t, group=group)
        return out.cpu().numpy().reshape(world, -1)

    def gather_obj(o):
        if world == 1:
            return [o]
        out = [None] * world
        dist.all_gather_object(out, o, group=group)
        return out

    own, ok, blocks = None, 1, None
    try:
        off = N.bgzf_index(path, lib=lib)
        size = __import__("os").path.getsize(path)
        blocks = [0] + [int(np.searchsorted(off, size * p // world)) for p in range(1, world)] + [len(off)]
        own = N.decode_span(path, blocks[rank], blocks[rank + 1], threads=threads, lib=lib)
    except Exception:   # noqa: BLE001 -- ANY failure of this rank's share must reach the collective below, or the others wait for it
        own, ok = None, 0   # not BGZF (SAM text, plain gzip), a span that does not close, an RNAME without @SQ line, no memory:
        #                     every rank then reads the whole file (and a real error is raised by all of them alike)
    n = len(own["contig"]) if own is not None else 0
    first = last = -1
    is_sorted, reach, lead, s0, s1 = 1, 0, 0, 0, 0
    if own is not None:
        lens = own["contig_lens"]
        base, S = g_layout(lens)
        s0, s1 = own["span"][0], own["span"][1]
        if n:
            g0 = _start_keys(base, own)
            g_lo, g_hi = footprints(lens, own)
            first, last = int(g0[0]), int(g0[-1])
            is_sorted = int(bool(np.all(g0[1:] >= g0[:-1])) and bool(np.all(np.asarray(own["pos0"]) >= 0)))
            reach, lead = int((g_hi - g0).max()), int((g0 - g_lo).max())
    rows = gather_rows([ok, n, first, last, is_sorted, reach, lead, s0, s1])       # collective 1 of 2 (72 bytes per rank)
    good = bool(rows[:, 0].all() and rows[:, 4].all())
    if good:
        good = all(rows[r, 8] == rows[r + 1, 7] for r in range(world - 1))        # every span ends where the next begins
        filled = [r for r in range(world) if rows[r, 1] > 0]
        good = good and all(rows[a, 3] <= rows[b, 2] for a, b in zip(filled, filled[1:]))   # shares ordered by position
    if not good:
        # premises broken: every rank reads the whole file and takes its share of an equal-work partition (same results)
        try:
            full = N.decode_file(path, threads=threads, lib=lib)
        except Exception as e:   # noqa: BLE001 -- one rank alone failing here (no memory) would leave the others in the next collective:
            #                      the failure travels in the result and pileup_consensus_sharded's gather raises it on every rank
            return dict(batch=None, interval=(0, 0), intervals=[(0, 0)] * world, names=[], lens=np.zeros(0, np.uint32), order=[],
                        mode="whole-file", stats={}, fail=(type(e).__name__, str(e)))
        lens = full["contig_lens"]
        ivs = partition_weighted(lens, full["contig"], full["pos0"], full["seq_len"], world) if len(full["contig"]) else partition(lens, world)
        batch = None
        if len(full["contig"]):
            g_lo, g_hi = footprints(lens, full)
            batch = _subset(full, reads_of_rank(lens, g_lo, g_hi, rank, world, intervals=ivs))
        c = np.asarray(full["contig"])
        _, fi = np.unique(c, return_index=True)
        order = [int(c[i]) for i in np.sort(fi)]
        return dict(batch=batch, interval=ivs[rank], intervals=ivs, names=[str(x) for x in full["contig_names"]], lens=lens, order=order,
                    mode="whole-file", stats=dict(decoded_records=len(c)))
    lens = own["contig_lens"]
    base, S = g_layout(lens)
    # rank r owns [cut_r, cut_{r+1}): from its first read's start to the next non-empty share's
    cuts = [0] * (world + 1)
    cuts[world] = S
    nxt = S
    for r in range(world - 1, 0, -1):
        if rows[r, 1] > 0:
            nxt = int(rows[r, 2])
        cuts[r] = nxt
    ivs = [(cuts[r], cuts[r + 1]) for r in range(world)]
    lo, hi = ivs[rank]
    M, Lmax = int(rows[:, 5].max()), int(rows[:, 6].max())
    pieces, extra = [own], 0
    fail = None     # an exception of this rank's neighbour decode: carried through the closing collective, raised by every rank
    try:
        if hi > lo:
            # reads of EARLIER shares that reach into [lo, ...): they start after lo - M
            if rank > 0 and lo > 0:
                b_hi, step, back = blocks[rank], 1, []
                while b_hi > 0:
                    b_lo = max(0, b_hi - step)
                    sp = N.decode_span(path, b_lo, b_hi, threads=threads, lib=lib)
                    back.insert(0, sp)
                    b_hi, step = b_lo, step * 2
                    if len(sp["contig"]) and int(_start_keys(base, sp)[0]) + M <= lo:
                        break
                pieces = back + pieces
                extra += sum(len(b["contig"]) for b in back)
            # reads of LATER shares that touch the halo site hi (or reach back over it with a leading clip): they start by hi + Lmax
            if rank + 1 < world and hi < S:
                b_lo, step, fwd = blocks[rank + 1], 1, []
                nblk = blocks[world]
                while b_lo < nblk:
                    b_hi = min(nblk, b_lo + step)
                    sp = N.decode_span(path, b_lo, b_hi, threads=threads, lib=lib)
                    fwd.append(sp)
                    b_lo, step = b_hi, step * 2
                    if len(sp["contig"]) and int(_start_keys(base, sp)[-1]) > hi + Lmax:
                        break
                pieces = pieces + fwd
                extra += sum(len(b["contig"]) for b in fwd)
    except Exception as e:   # noqa: BLE001
        fail = (type(e).__name__, str(e))
        pieces = [own]
    batch = concat_batches(pieces) if hi > lo else None
    if batch is not None:
        g_lo, g_hi = footprints(lens, batch)
        batch = _subset(batch, (g_hi > lo) & (g_lo <= hi))
    used = gather_obj((np.unique(np.asarray(own["contig"])).tolist(), fail))      # (a handful of ints per rank; same collective round)
    failed = next((f for _, f in used if f), None)
    if failed:          # the lowest failing rank's exception, on every rank
        raise {"KeyError": KeyError, "IndexError": IndexError, "MemoryError": MemoryError, "OSError": OSError}.get(failed[0], RuntimeError)(
            "rank-sharded ingest: " + failed[1])
    order = sorted(set(c for u, _ in used for c in u))   # coordinate-sorted file: first appearance = ascending contig id
    return dict(batch=batch, interval=(lo, hi), intervals=ivs, names=[str(x) for x in own["contig_names"]], lens=lens, order=order,
                mode="sharded", stats=dict(decoded_records=n, neighbour_records=extra, blocks=(blocks[rank], blocks[rank + 1])))


def realign_patches(eng, order, lens, interval, device, group, world, min_overlap, clip_decay_threshold, mask_ends, keep=None):
    """--realign across ranks (kindel.py:502-513) WITHOUT a table crossing the links (round 6; round 5 summed the shards' tables with
    one all-reduce per contig: 128 bytes per site, 640 MB per rank on a 5 Mbp contig -- the collective SURVEY 8e rejected).  The
    clip-dominant-region scans (kindel.py:156-275) read a contig through per-site predicates only -- "the clip depth dominates
    here", "an extension runs over this site" -- and a site's counters are complete on the rank that owns it, so every rank
    evaluates the predicates on ITS sites (kindel.cdr_scan_inputs over the rows of its interval), the SPARSE results -- position and
    consensus character of the few sites where a predicate holds: nine bytes each -- are gathered in one small collective for all
    contigs, and every rank runs the same scans over the same concatenated lists (kindel.cdr_start_regions / cdr_end_regions):
    identical patches everywhere, a region that crosses a cut is simply a run of sites whose entries came from two ranks.
    -> {cid: merged patches (kindel.merge_cdrps)}; keep (tests): the gathered scan inputs per contig.
    An exception on one rank alone travels in the gathered object and is raised on every rank (no rank waits in a collective)."""
    import torch.distributed as dist
    from . import _native as N
    from . import kindel as K
    mine, err = {}, None
    try:
        for cid in order:
            L = int(lens[cid])
            b = eng.contig_base(cid)
            a0, a1 = max(0, min(L, int(interval[0]) - b)), max(0, min(L, int(interval[1]) - b))      # this rank's rows of the contig
            if a1 <= a0:
                continue
            t = eng.tables(cid, np.arange(N.KD_CH_CLIP_STARTS, dtype=np.uint32))     # weights, deletions, clip start / end weights: what the scans read
            W = np.ascontiguousarray(t[0:5, a0:a1].T)
            d = t[N.KD_CH_DEL, a0:a1]
            S = np.ascontiguousarray(t[N.KD_CH_CSW:N.KD_CH_CSW + 5, a0:a1].T)
            E = np.ascontiguousarray(t[N.KD_CH_CEW:N.KD_CH_CEW + 5, a0:a1].T)
            mine[cid] = (K.cdr_scan_inputs(W, d, S, clip_decay_threshold, mask_ends, L, offset=a0),
                         K.cdr_scan_inputs(W, d, E, clip_decay_threshold, mask_ends, L, offset=a0))
    except Exception as e:   # noqa: BLE001
        err = (type(e).__name__, str(e))
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, (mine, err), group=group)
    else:
        parts = [(mine, err)]
    first_err = next((e for _, e in parts if e), None)
    if first_err:
        raise {"KeyError": KeyError, "IndexError": IndexError, "MemoryError": MemoryError, "OSError": OSError}.get(first_err[0], RuntimeError)(
            "realign across ranks: " + first_err[1])
    patches = {}
    for cid in order:
        L = int(lens[cid])
        s_in = K.cdr_scan_concat([m[cid][0] for m, _ in parts if cid in m])
        e_in = K.cdr_scan_concat([m[cid][1] for m, _ in parts if cid in m])
        if keep is not None:
            keep[cid] = (s_in, e_in)
        cdrps = K.pair_cdrs(K.cdr_start_regions(L, s_in), K.cdr_end_regions(L, e_in))
        patches[cid] = K.merge_cdrps(cdrps, min_overlap)
    return patches


def pileup_consensus_sharded(path, rank, world, device="cpu", dev_index=0, group=None, min_depth=1, threads=0, lib=None, realign=None):
    """ingest_sharded + shard-local pileup + consensus + stitch.  -> dict(seqs, changes, minmax, names, lens, order, mode, stats
    [, patches]), identical on every rank.  A reference exception raised by any rank's reads (KeyError / IndexError / RuntimeError,
    kindel.py:47-81) is agreed on by all ranks and raised everywhere.
    realign: None, or dict(min_overlap, clip_decay_threshold, mask_ends): clip-dominant regions are patched (realign_patches); every
    rank's consensus run skips the patched sites of its shard and the patch texts are spliced into the stitched contigs."""
    import torch
    import torch.distributed as dist
    from . import _native as N
    ing = ingest_sharded(path, rank, world, device=device, group=group, threads=threads, lib=lib)
    eng = None
    err = ing.get("fail")       # (the whole-file fallback's decode failed on this rank)
    try:
        try:        # ANY exception of this rank (a reference exception, no device memory -- also for the tables themselves --, a native
            #         error) goes through the gather
            if err is None:
                eng = N.Engine(ing["lens"], device=dev_index, lib=lib)
                eng.set_shard(*ing["interval"])
                if ing["batch"] is not None and len(ing["batch"]["contig"]):
                    eng.push(ing["batch"])
                eng.finalize()
        except Exception as e:   # noqa: BLE001
            err = (type(e).__name__, str(e))
        if world > 1:
            errs = [None] * world
            dist.all_gather_object(errs, err, group=group)
        else:
            errs = [err]
        first_err = next((e for e in errs if e), None)
        if first_err:     # (the lowest rank's = the earliest position's, for a sorted file the reference's own choice)
            raise {"KeyError": KeyError, "IndexError": IndexError, "MemoryError": MemoryError, "OSError": OSError}.get(first_err[0], RuntimeError)(first_err[1])
        patches = None
        if realign is None:
            eng.consensus_run(min_depth)
            seqs, changes, minmax = stitch(eng, ing["interval"], device, group, intervals=ing["intervals"])
        else:
            from . import kindel as K
            kept = {} if realign.get("keep_scan_inputs") else None
            patches = realign_patches(eng, ing["order"], ing["lens"], ing["interval"], device, group, world, realign["min_overlap"],
                                      realign["clip_decay_threshold"], realign["mask_ends"], keep=kept)
            plans = {cid: K._patch_plan(int(ing["lens"][cid]), patches.get(cid)) for cid in ing["order"]}
            flat, owner = [], []
            for cid in ing["order"]:
                b = eng.contig_base(cid)
                for s_, e_, _ in plans[cid]:
                    flat.append((b + s_, b + e_)); owner.append(cid)
            eng.consensus_run(min_depth, flat)
            seqs, changes, minmax, part_lens = stitch(eng, ing["interval"], device, group, intervals=ing["intervals"], with_parts=True)
            if flat:
                # where a patch's text goes: the offset its OWNER (the rank whose interval holds the patch's first site) recorded,
                # relative to its part of the contig, behind the parts of the ranks in front
                local = np.full(len(flat), -1, np.int64)
                for cid in set(owner):
                    po = eng.consensus_meta(cid)[2]
                    for k, c in enumerate(owner):
                        if c == cid and po[k] != np.uint64(0xFFFFFFFFFFFFFFFF):
                            local[k] = int(po[k])
                if world > 1:
                    x = torch.from_numpy(local).to(device)
                    allx = torch.empty(world * len(flat), dtype=torch.int64, device=device)
                    dist.all_gather_into_tensor(allx, x, group=group)
                    local_all = allx.cpu().numpy().reshape(world, len(flat))
                else:
                    local_all = local.reshape(1, -1)
                ivs = ing["intervals"]
                seqs = list(seqs)
                for cid in set(owner):
                    raw, parts, prev = seqs[cid], [], 0
                    ks = [k for k, c in enumerate(owner) if c == cid]
                    for (s_, e_, text), k in zip(plans[cid], ks):
                        g = flat[k][0]
                        r = next(r for r in range(world) if ivs[r][0] <= g < ivs[r][1])
                        if local_all[r][k] < 0:
                            raise RuntimeError("realign across ranks: rank %d did not record the offset of the patch at site %d" % (r, g))
                        o = sum(part_lens[cid][:r]) + int(local_all[r][k])
                        parts.append(raw[prev:o]); parts.append(text.encode()); prev = o
                    parts.append(raw[prev:])
                    seqs[cid] = b"".join(parts)
    finally:
        if eng is not None:
            eng.close()
    return dict(seqs=seqs, changes=changes, minmax=minmax, names=ing["names"], lens=ing["lens"], order=ing["order"], mode=ing["mode"],
                stats=ing["stats"], patches=patches, scan_inputs=kept if realign else None)
