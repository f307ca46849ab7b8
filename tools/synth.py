"""Seeded synthetic alignments for the benchmark configs of BASELINE.json / SURVEY.md section 8d.

Generates SoA read batches (the kd_batch layout of include/kindel_hip.h) directly as torch
tensors -- on the GPU for the full-size configs, on the CPU for tests -- plus a minimal
BGZF/BAM writer so end-to-end runs (`kindel consensus x.bam`) have real files without samtools.
torch is used here as plumbing for bulk random data only; nothing of the hot path lives here.

Short-read model (configs C2, C3, C4): uniform starts, 150 bp reads, per-base substitution
0.5 %, N 0.1 %, 7.5 % of reads carry one insertion and 7.5 % one deletion (geometric length),
5 % a leading and 5 % a trailing soft clip of 5-40 bases; reads are coordinate sorted.
Planted features make every branch of consensus_sequence (/root/reference/kindel/kindel.py:413-424)
fire: majority-deletion sites, majority-insertion sites, a zero-coverage gap, and inside the gap
hand-built read pairs producing exact base ties and insertion ties.

Long-read model (config C5): ONT-like reads (log-normal length, median 10 kb), CIGARs with
thousands of ops, I and D each ~7.5 % of op-bases, 20 % of reads soft-clipped at both ends.
"""
import struct
import zlib

import numpy as np
import torch

NIB = torch.tensor([1, 2, 4, 8], dtype=torch.uint8)  # A C G T in BAM nibble codes

#: BASELINE.json configs -> generator arguments
CONFIGS = {
    "C2": dict(kind="short", contig_lens=[10_000], depth=10_000, seed=2),
    "C3": dict(kind="short", contig_lens=[5_000_000], depth=500, seed=3),
    "C4": dict(kind="short", contig_lens=[50_000] * 100, depth=1000, seed=4),
    "C5": dict(kind="long", contig_lens=[1_000_000], depth=200, seed=5),
}

FIELDS = ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig", "seq4", "cigar")


def _geom(gen, n, p, device):
    """1 + Geometric(p) (support 1,2,...)"""
    u = torch.rand(n, generator=gen, device=device).clamp_(1e-12, 1.0)
    return (torch.log(u) / np.log(1.0 - p)).floor().to(torch.int64) + 1


def _pack_nibbles(codes):
    """[n, even] uint8 nibble codes -> [n, even/2] bytes, high nibble first (BAM)."""
    return (codes[:, 0::2] << 4) | codes[:, 1::2]


def short_reads(contig_lens, depth, read_len=150, seed=0, device="cpu", planted=True, chunk=1 << 20,
                shard=None, clip_p=0.05, indel_p=0.075):
    """-> dict of torch tensors (FIELDS + contig_lens) for a coordinate-sorted short-read batch.

    clip_p: probability of a leading (and, independently, trailing) soft clip; indel_p: probability of one
    insertion (and, separately, one deletion) per read -- the SURVEY section 8d error model is the default.

    shard=(rank, world): generate only the reads whose contig/interval belongs to `rank` under
    kindel_amd.shard.partition (used by bench.py --gpus N so every rank synthesises its own part).
    """
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    rl = int(read_len)
    assert rl % 2 == 0
    out = {k: [] for k in ("contig", "pos0", "n_cig", "ops", "seq")}
    nib = NIB.to(dev)
    refs = []
    if shard is not None:
        from kindel_amd import shard as _sh
        _base, _ = _sh.g_layout(contig_lens)
        _lo, _hi = _sh.partition(contig_lens, shard[1])[shard[0]]
    for c, L in enumerate(contig_lens):
        L = int(L)
        if shard is not None and (int(_base[c]) + L + 1024 < _lo or int(_base[c]) - 1024 > _hi):
            continue  # contig entirely outside this rank's interval: another rank synthesises it
        g = torch.Generator(device=dev)
        g.manual_seed(int(seed) * 1000003 + c)
        ref = torch.randint(0, 4, (L,), generator=g, device=dev, dtype=torch.uint8)
        refs.append(ref)
        n = max(1, int(round(L * depth / rl)))
        # planted sites: spaced > 2 reads apart, away from the ends
        n_plant = 0
        if planted and L >= 40 * rl:
            n_plant = min(40, L // (4 * rl))
            psite = torch.linspace(2 * rl, L - 3 * rl, n_plant, device=dev).to(torch.int64)
            ptype = (torch.arange(n_plant, device=dev) % 2)          # 0: deletion site, 1: insertion site
            plen = 1 + (torch.arange(n_plant, device=dev) % 3)
            gap0, gap1 = L // 2, L // 2 + 4 * rl                      # zero-coverage gap (ties live inside)
        for s0 in range(0, n, chunk):
            m = min(chunk, n - s0)
            s1 = torch.where(torch.rand(m, generator=g, device=dev) < clip_p,
                             torch.randint(5, 41, (m,), generator=g, device=dev), torch.zeros(m, dtype=torch.int64, device=dev))
            s2 = torch.where(torch.rand(m, generator=g, device=dev) < clip_p,
                             torch.randint(5, 41, (m,), generator=g, device=dev), torch.zeros(m, dtype=torch.int64, device=dev))
            u = torch.rand(m, generator=g, device=dev)
            ityp = torch.where(u < indel_p, 1, torch.where(u < 2 * indel_p, 2, 0)).to(torch.int64)  # 1 = I, 2 = D
            ilen = torch.minimum(_geom(g, m, 0.7, dev), torch.tensor(20, device=dev))
            qins = torch.where(ityp == 1, ilen, torch.zeros_like(ilen))
            alen = rl - s1 - s2 - qins                                # aligned query bases
            ipos = 10 + (torch.rand(m, generator=g, device=dev) * (alen - 20).clamp(min=1)).to(torch.int64)
            dlen = torch.where(ityp == 2, ilen, torch.zeros_like(ilen))
            span = alen + dlen
            start = (torch.rand(m, generator=g, device=dev, dtype=torch.float64) * (L - span - 1).clamp(min=1)).to(torch.int64)
            ibase = torch.randint(0, 4, (m, 20), generator=g, device=dev, dtype=torch.uint8)  # inserted bases
            if n_plant:
                # reads covering a planted site (with margin) take the planted indel with p = 0.8
                k = torch.searchsorted(psite, start + s1 * 0 + 12).clamp(max=n_plant - 1)
                p = psite[k]
                covers = (p >= start + 12) & (p <= start + alen - 12) & (torch.rand(m, generator=g, device=dev) < 0.8)
                pt, pl = ptype[k], plen[k]
                ityp = torch.where(covers, torch.where(pt == 0, 2, 1), ityp)
                ilen = torch.where(covers, pl, ilen)
                qins_new = torch.where(ityp == 1, ilen, torch.zeros_like(ilen))
                # keep the query length fixed: an insertion eats aligned bases at the 3' end
                alen = rl - s1 - s2 - qins_new
                qins = qins_new
                dlen = torch.where(ityp == 2, ilen, torch.zeros_like(ilen))
                ipos = torch.where(covers, p - start, ipos.clamp(max=(alen - 10).clamp(min=1)))
                # planted insertion text is a function of the site so that reads agree
                pb = ((p.unsqueeze(1) * 7 + torch.arange(20, device=dev).unsqueeze(0) * 3) % 4).to(torch.uint8)
                ibase = torch.where((covers & (ityp == 1)).unsqueeze(1), pb, ibase)
                span = alen + dlen
                keep = ~((start < gap1) & (start + span > gap0))
            else:
                keep = torch.ones(m, dtype=torch.bool, device=dev)
            m1 = torch.where(ityp > 0, ipos, alen)
            m2 = alen - m1
            # ---- bases: ref index (or -1 for random) per query position ----
            j = torch.arange(rl, device=dev).unsqueeze(0)
            a0 = s1.unsqueeze(1); a1 = (s1 + m1).unsqueeze(1); a2 = (s1 + m1 + qins).unsqueeze(1); a3 = (rl - s2).unsqueeze(1)
            st = start.unsqueeze(1)
            ridx = torch.where((j >= a0) & (j < a1), st + j - a0,
                               torch.where((j >= a2) & (j < a3), st + m1.unsqueeze(1) + dlen.unsqueeze(1) + j - a2,
                                           torch.full_like(j, -1)))
            rnd = torch.randint(0, 4, (m, rl), generator=g, device=dev, dtype=torch.uint8)
            code = torch.where(ridx >= 0, ref[ridx.clamp(min=0)], rnd)
            in_ins = (j >= a1) & (j < a2)
            code = torch.where(in_ins, torch.gather(ibase, 1, (j - a1).clamp(0, 19).expand(m, rl)), code)
            sub = torch.rand(m, rl, generator=g, device=dev) < 0.005
            code = torch.where(sub & ~in_ins, rnd, code)
            nibs = nib[code.long()]
            nmask = torch.rand(m, rl, generator=g, device=dev) < 0.001
            nibs = torch.where(nmask & ~in_ins, torch.tensor(15, dtype=torch.uint8, device=dev), nibs)
            # ---- cigar: up to 5 ops  S M I|D M S ----
            oplen = torch.stack([s1, m1, torch.where(ityp == 1, qins, dlen), torch.where(ityp > 0, m2, torch.zeros_like(m2)), s2], 1)
            opcode = torch.stack([torch.full_like(s1, 4), torch.zeros_like(s1), torch.where(ityp == 1, 1, 2),
                                  torch.zeros_like(s1), torch.full_like(s1, 4)], 1)
            ops = torch.where(oplen > 0, (oplen << 4) | opcode, torch.full_like(oplen, -1))
            out["contig"].append(torch.full((int(keep.sum()),), c, dtype=torch.int32, device=dev))
            out["pos0"].append(start[keep].to(torch.int32))
            out["ops"].append(ops[keep])
            out["seq"].append(_pack_nibbles(nibs[keep]))
        if n_plant:
            cr = _crafted_tie_reads(ref, gap0 + rl // 2, rl, nib, dev)
            out["contig"].append(torch.full((cr["pos0"].numel(),), c, dtype=torch.int32, device=dev))
            out["pos0"].append(cr["pos0"]); out["ops"].append(cr["ops"]); out["seq"].append(cr["seq"])
    contig = torch.cat(out["contig"]); pos0 = torch.cat(out["pos0"]); ops = torch.cat(out["ops"]); seq = torch.cat(out["seq"])
    if shard is not None:
        keep = _sh.reads_touching(contig_lens, contig, pos0, pos0 + rl + 64, shard[0], shard[1])
        contig, pos0, ops, seq = contig[keep], pos0[keep], ops[keep], seq[keep]
    key = contig.to(torch.int64) * (1 << 40) + pos0.to(torch.int64)
    order = torch.argsort(key, stable=True)
    contig, pos0, ops, seq = contig[order], pos0[order], ops[order], seq[order]
    R = contig.numel()
    valid = ops >= 0
    n_cig = valid.sum(1).to(torch.int32)
    cig_off = torch.cumsum(n_cig.to(torch.int64), 0) - n_cig
    cigar = ops[valid].to(torch.int32)
    batch = dict(
        contig=contig, pos0=pos0, flag=torch.zeros(R, dtype=torch.int32, device=dev),
        seq_off=torch.arange(R, device=dev, dtype=torch.int64) * (rl // 2),
        seq_len=torch.full((R,), rl, dtype=torch.int32, device=dev), cig_off=cig_off, n_cig=n_cig,
        seq4=torch.cat([seq.reshape(-1), torch.zeros(16, dtype=torch.uint8, device=dev)]),
        cigar=torch.cat([cigar, torch.zeros(4, dtype=torch.int32, device=dev)]),
        contig_lens=np.asarray(contig_lens, np.uint32), seq4_bytes=R * (rl // 2), cigar_words=int(cigar.numel()),
    )
    return batch


def _crafted_tie_reads(ref, at, rl, nib, dev):
    """Inside the zero-coverage gap: 6 pairs with one differing base (1:1 base tie -> 'N') and 6 pairs
    with different 2-base insertions at the same site (insertion tie -> 'N'), kindel.py:377,421,424."""
    pos0, ops, seqs = [], [], []
    for k in range(12):
        st = at + k * (rl // 4) * 0 + (k // 2) * 0  # all pairs stacked on the same window: depth stays tiny
        st = at + (k // 2) * 3
        base = ref[st:st + rl].clone()
        if k < 6:  # base tie at offset 70 + pair index
            code = base.clone()
            if k % 2:
                code[70 + k // 2] = (code[70 + k // 2] + 1 + (k // 2) % 3) % 4
            o = torch.tensor([-1, (rl << 4) | 0, -1, -1, -1], device=dev)
        else:      # insertion tie after 60 aligned bases
            code = torch.cat([base[:60], torch.full((2,), (k % 2) * 2 + ((k // 2) % 2), dtype=torch.uint8, device=dev),
                              base[60:rl - 2]])
            o = torch.tensor([-1, (60 << 4) | 0, (2 << 4) | 1, ((rl - 62) << 4) | 0, -1], device=dev)
        pos0.append(st); ops.append(o); seqs.append(nib[code.long()])
    return dict(pos0=torch.tensor(pos0, dtype=torch.int32, device=dev), ops=torch.stack(ops),
                seq=_pack_nibbles(torch.stack(seqs)))


def long_reads(contig_lens, depth, seed=0, device="cpu", median_len=10_000, min_len=2_000, max_len=30_000,
               shard=None):
    """ONT-like batch: CIGARs alternate M runs (mean ~12) with I/D ops (mean 1.5), ~15 % indel op-bases,
    5 % substitutions, 20 % of reads soft-clipped at both ends.  Coordinate sorted."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    nib = NIB.to(dev)
    L = int(contig_lens[0])
    assert len(contig_lens) == 1, "long-read model: single contig"
    ref = torch.randint(0, 4, (L,), generator=g, device=dev, dtype=torch.uint8)
    n = max(1, int(round(L * depth / (median_len * 1.13))))
    rlen = torch.exp(torch.randn(n, generator=g, device=dev) * 0.5 + np.log(median_len)).clamp(min_len, max_len)
    blocks = (rlen / 13.0).ceil().to(torch.int64).clamp(min=2)   # one block = M run + one indel op
    nb = int(blocks.sum())
    mlen = _geom(g, nb, 1.0 / 12.0, dev)
    ilen = _geom(g, nb, 1.0 / 1.5, dev)
    is_ins = torch.rand(nb, generator=g, device=dev) < 0.5
    rid = torch.repeat_interleave(torch.arange(n, device=dev), blocks)
    first = torch.cumsum(blocks, 0) - blocks
    last_block = torch.zeros(nb, dtype=torch.bool, device=dev)
    last_block[first + blocks - 1] = True
    ilen = torch.where(last_block, torch.zeros_like(ilen), ilen)   # reads end on an M run
    clip = torch.rand(n, generator=g, device=dev) < 0.2
    c1 = torch.where(clip, torch.randint(10, 201, (n,), generator=g, device=dev), torch.zeros(n, dtype=torch.int64, device=dev))
    c2 = torch.where(clip, torch.randint(10, 201, (n,), generator=g, device=dev), torch.zeros(n, dtype=torch.int64, device=dev))
    radv = mlen + torch.where(is_ins, torch.zeros_like(ilen), ilen)
    qadv = mlen + torch.where(is_ins, ilen, torch.zeros_like(ilen))
    rcum = torch.cumsum(radv, 0); qcum = torch.cumsum(qadv, 0)
    rspan = rcum[first + blocks - 1] - (rcum[first] - radv[first])
    qspan = qcum[first + blocks - 1] - (qcum[first] - qadv[first])
    start = (torch.rand(n, generator=g, device=dev, dtype=torch.float64) * (L - rspan - 1).clamp(min=1)).to(torch.int64)
    if shard is not None:
        from kindel_amd import shard as _sh
        keep = _sh.reads_touching(contig_lens, torch.zeros(n, dtype=torch.int32, device=dev), start, start + rspan,
                                 shard[0], shard[1])
    else:
        keep = torch.ones(n, dtype=torch.bool, device=dev)
    order = torch.argsort(torch.where(keep, start, torch.full_like(start, 1 << 60)), stable=True)[: int(keep.sum())]
    # per-block start coordinates relative to the read
    r0 = (rcum - radv) - (rcum[first] - radv[first])[rid]
    q0 = (qcum - qadv) - (qcum[first] - qadv[first])[rid] + c1[rid]
    seq_len = qspan + c1 + c2
    # ---- cigar words: [S] (M, I|D)* M [S]  ----
    n_ops = (c1 > 0).to(torch.int64) + 2 * blocks - 1 + (c2 > 0).to(torch.int64)
    # build in read order `order`
    inv = torch.empty(n, dtype=torch.int64, device=dev); inv[order] = torch.arange(order.numel(), device=dev)
    sel = keep[rid]
    new_first_op = torch.zeros(n, dtype=torch.int64, device=dev)
    no = n_ops[order]
    new_first_op[order] = torch.cumsum(no, 0) - no
    total_ops = int(no.sum())
    cigar = torch.zeros(total_ops + 4, dtype=torch.int64, device=dev)
    bpos = torch.arange(nb, device=dev) - first[rid]      # block index within read
    mslot = new_first_op[rid] + (c1[rid] > 0).to(torch.int64) + 2 * bpos
    cigar[mslot[sel]] = (mlen[sel] << 4)
    isel = sel & ~last_block
    cigar[(mslot + 1)[isel]] = (ilen[isel] << 4) | torch.where(is_ins[isel], 1, 2)
    k1 = keep & (c1 > 0); cigar[new_first_op[k1]] = (c1[k1] << 4) | 4
    k2 = keep & (c2 > 0); cigar[(new_first_op + n_ops - 1)[k2]] = (c2[k2] << 4) | 4
    # ---- bases ----
    sl = seq_len[order]
    sbytes = (sl + 1) // 2
    seq_off = torch.cumsum(sbytes, 0) - sbytes
    total_nib = int(sbytes.sum()) * 2
    nibs = nib[torch.randint(0, 4, (total_nib + 32,), generator=g, device=dev).long()]  # clips + insertions stay random
    # aligned bases: for every M run copy ref[start + r0 .. ] with 5 % substitutions
    msel = sel
    mq = (seq_off[inv[rid]] * 2 + q0)[msel]
    mr = (start[rid] + r0)[msel]
    ml = mlen[msel]
    tot = int(ml.sum())
    run = torch.repeat_interleave(torch.arange(ml.numel(), device=dev), ml)
    within = torch.arange(tot, device=dev) - torch.repeat_interleave(torch.cumsum(ml, 0) - ml, ml)
    code = ref[(mr[run] + within).clamp(max=L - 1)]
    subm = torch.rand(tot, generator=g, device=dev) < 0.05
    code = torch.where(subm, torch.randint(0, 4, (tot,), generator=g, device=dev, dtype=torch.uint8), code)
    nibs[mq[run] + within] = nib[code.long()]
    # zero the pad nibble of odd-length reads
    odd = (sl % 2) == 1
    nibs[(seq_off * 2 + sl)[odd]] = 0
    seq4 = (nibs[0:total_nib:2] << 4) | nibs[1:total_nib:2]
    R = order.numel()
    return dict(
        contig=torch.zeros(R, dtype=torch.int32, device=dev), pos0=start[order].to(torch.int32),
        flag=torch.zeros(R, dtype=torch.int32, device=dev), seq_off=seq_off, seq_len=sl.to(torch.int32),
        cig_off=new_first_op[order], n_cig=no.to(torch.int32),
        seq4=torch.cat([seq4, torch.zeros(16, dtype=torch.uint8, device=dev)]), cigar=cigar.to(torch.int32),
        contig_lens=np.asarray(contig_lens, np.uint32), seq4_bytes=total_nib // 2, cigar_words=total_ops,
    )


def make(config, scale=1.0, device="cpu", shard=None):
    """config: 'C2'..'C5' or a dict like CONFIGS[...]; scale < 1 shrinks depth for tests."""
    cfg = dict(CONFIGS[config]) if isinstance(config, str) else dict(config)
    kind = cfg.pop("kind")
    cfg["depth"] = cfg["depth"] * scale
    fn = short_reads if kind == "short" else long_reads
    return fn(device=device, shard=shard, **cfg)


def shuffled(batch, mode="records", seed=None):
    """The batch with its reads in random order (unsorted input).  mode "records": what a decoder hands over for an unsorted FILE --
    bases and CIGAR words lie in record order too, offsets ascend with the read index; "index": only the per-read arrays are
    permuted, every read's bases / CIGAR stay where the sorted batch had them (a layout no file produces: rounds 1 - 2 measured it)."""
    dev = batch["contig"].device
    n = int(batch["contig"].numel())
    gen = None
    if seed is not None:
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
    perm = torch.randperm(n, device=dev, generator=gen)
    out = dict(batch)
    for k in ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig"):
        out[k] = batch[k][perm].contiguous()
    if mode == "index":
        return out

    def regroup(data, off, cnt, pad):
        cnt = cnt.long()
        new_off = torch.cumsum(cnt, 0) - cnt
        total = int(cnt.sum())
        res = torch.zeros(total + pad, dtype=data.dtype, device=dev)
        step = 1 << 24                      # reads per piece: bounds the index tensors
        for a0 in range(0, n, step):
            c, no, oo = cnt[a0:a0 + step], new_off[a0:a0 + step], off[a0:a0 + step].long()
            seg = torch.repeat_interleave(torch.arange(c.numel(), device=dev), c)
            dst = torch.arange(int(no[0]), int(no[0]) + int(c.sum()), device=dev)
            res[dst] = data[oo[seg] + (dst - no[seg])]
        return res, new_off.to(off.dtype), total

    out["cigar"], out["cig_off"], out["cigar_words"] = regroup(batch["cigar"], out["cig_off"], out["n_cig"], 2)
    out["seq4"], out["seq_off"], out["seq4_bytes"] = regroup(batch["seq4"], out["seq_off"], (out["seq_len"].long() + 1) // 2, 64)
    return out


def to_numpy(batch):
    """torch batch -> dict of numpy arrays in the dtypes of kd_batch (for the oracle / kd_push_batch)."""
    dt = dict(contig=np.uint32, pos0=np.int32, flag=np.uint32, seq_off=np.uint64, seq_len=np.uint32,
              cig_off=np.uint64, n_cig=np.uint32, seq4=np.uint8, cigar=np.uint32)
    out = {k: batch[k].detach().cpu().numpy().astype(dt[k], copy=False) for k in FIELDS}
    out["seq4"] = out["seq4"][: batch["seq4_bytes"] + 8]
    out["cigar"] = out["cigar"][: batch["cigar_words"] + 2]
    out["contig_lens"] = np.asarray(batch["contig_lens"], np.uint32)
    out["contig_names"] = np.asarray(["ctg%d" % i for i in range(len(out["contig_lens"]))])
    return out


def device_ptrs(batch):
    """field -> device address, for kd_push_batch_device.  The library launches on its own stream (include/kindel_hip.h): the
    torch kernels that wrote the arrays are waited for here, so that what the addresses point at is complete."""
    if batch["contig"].is_cuda:
        torch.cuda.synchronize(batch["contig"].device)
    return {k: batch[k].data_ptr() for k in FIELDS}


def counts(batch):
    """(reads, aligned-base events, walked events) of a batch, from its CIGAR words"""
    cg = batch["cigar"][: batch["cigar_words"]].to(torch.int64)
    ln, op = cg >> 4, cg & 15
    aligned = int(ln[(op == 0) | (op == 7) | (op == 8)].sum())
    walked = aligned + int(ln[(op == 1) | (op == 2) | (op == 4)].sum())
    return int(batch["contig"].numel()), aligned, walked


# --------------------------------------------------------------------------------------
# minimal BGZF / BAM writer (SAMv1 4.1, 4.2) -- for tests and end-to-end runs only
# --------------------------------------------------------------------------------------
def _bgzf_block(data):
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = comp.compress(data) + comp.flush()
    bsize = len(body) + 25
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + body +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def write_bam(path, batch, names=None, sort_order="coordinate", block_bytes=0xff00):
    """Write a numpy batch (see to_numpy) as a BAM file (block_bytes: uncompressed bytes per BGZF block; tests use small ones)."""
    lens = [int(x) for x in batch["contig_lens"]]
    names = [str(x) for x in (names if names is not None else batch.get("contig_names", ["ctg%d" % i for i in range(len(lens))]))]
    text = "@HD\tVN:1.6\tSO:%s\n" % sort_order + "".join("@SQ\tSN:%s\tLN:%d\n" % (n, l) for n, l in zip(names, lens))
    buf = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(lens)))
    for n, l in zip(names, lens):
        buf += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    n = len(batch["contig"])
    for i in range(n):
        sl, nc = int(batch["seq_len"][i]), int(batch["n_cig"][i])
        so, co = int(batch["seq_off"][i]), int(batch["cig_off"][i])
        name = b"r%d\0" % i
        cig = batch["cigar"][co:co + nc].astype("<u4")
        aux = b""
        if nc > 65535:   # SAMv1 4.2.2: placeholder <l_seq>S<ref_len>N in the record, the real CIGAR in the CG:B,I tag
            ref_len = int(sum(int(c) >> 4 for c in cig if (int(c) & 15) in (0, 2, 3, 7, 8)))
            aux = b"CGBI" + struct.pack("<i", nc) + cig.tobytes()
            cig = np.asarray([(sl << 4) | 4, (ref_len << 4) | 3], "<u4")
        rec = struct.pack("<iiBBHHHiiii", int(np.int32(batch["contig"][i])), int(batch["pos0"][i]), len(name), 60, 0, len(cig),
                          int(batch["flag"][i]) & 0xffff, sl, -1, -1, 0)
        rec += name + cig.tobytes() + batch["seq4"][so:so + (sl + 1) // 2].tobytes()
        rec += b"\xff" * sl + aux
        buf += struct.pack("<i", len(rec)) + rec
    with open(path, "wb") as fh:
        for o in range(0, len(buf), block_bytes):
            fh.write(_bgzf_block(bytes(buf[o:o + block_bytes])))
        fh.write(_bgzf_block(b""))
