"""Seeded random read batches with arbitrary CIGARs (valid and invalid) for engine-vs-oracle fuzzing."""
import numpy as np

OPS = "MIDNSHP=X"


def random_batch(rng, n_reads, contig_lens=(37, 90), wild=0.15, sort=False, long_ops=None):
    """Small contigs so that overhangs / wrap-arounds / slot-L writes happen often.
    wild: probability that a read is built without any validity constraint.
    long_ops=(lo, hi): a tenth of the reads get lo..hi CIGAR ops (long-read path: checkpoints, segments);
    use with contigs long enough to hold them."""
    contig, pos0, flag, seq_off, seq_len, cig_off, n_cig = [], [], [], [], [], [], []
    seq4, cigar = bytearray(), []
    for _ in range(n_reads):
        c = int(rng.integers(0, len(contig_lens)))
        L = contig_lens[c]
        is_wild = rng.random() < wild
        lo_ops, hi_ops = long_ops if long_ops else (17, 40)
        n_ops = int(rng.integers(1, 7)) if rng.random() < 0.9 else int(rng.integers(lo_ops, hi_ops))
        ops = []
        for k in range(n_ops):
            if is_wild:
                op = int(rng.integers(0, 9)); ln = int(rng.integers(0, 12))
            else:
                r = rng.random()
                if k == 0 and r < 0.15: op, ln = 4, int(rng.integers(1, 8))          # leading S
                elif k == n_ops - 1 and k > 0 and r < 0.15: op, ln = 4, int(rng.integers(1, 8))  # trailing S
                elif r < 0.70: op, ln = [0, 7, 8][int(rng.integers(0, 3))], int(rng.integers(1, 15))
                elif r < 0.82: op, ln = 1, int(rng.integers(1, 4))
                elif r < 0.94: op, ln = 2, int(rng.integers(1, 4))
                else: op, ln = [3, 5, 6][int(rng.integers(0, 3))], int(rng.integers(1, 5))
            ops.append((ln, op))
        qlen = sum(ln for ln, op in ops if op in (0, 1, 4, 7, 8))
        rlen = sum(ln for ln, op in ops if op in (0, 2, 7, 8))
        if is_wild:
            sl = max(0, qlen + int(rng.integers(-3, 4)))
            p = int(rng.integers(-3, L + 4))
        else:
            sl = qlen
            # valid placement, including the extremes (first site, ending exactly on the last site)
            hi = max(0, L - rlen)
            p = [0, hi][int(rng.integers(0, 2))] if rng.random() < 0.2 else int(rng.integers(0, hi + 1))
            if rlen > L:
                ops = [(min(L, 5), 0)]; sl = min(L, 5); p = 0
        alphabet = [1, 2, 4, 8, 15] if (not is_wild or rng.random() < 0.7) else [1, 2, 4, 8, 15, 0, 3, 5, 9, 14]
        nib = [alphabet[int(x)] for x in rng.integers(0, len(alphabet), sl)]
        fl = 4 if rng.random() < 0.04 else [0, 16, 256, 2048][int(rng.integers(0, 4))]
        if is_wild and rng.random() < 0.1:
            ops = []           # CIGAR '*'
        contig.append(c); pos0.append(p); flag.append(fl)
        seq_off.append(len(seq4)); seq_len.append(sl); cig_off.append(len(cigar)); n_cig.append(len(ops))
        nn = nib + [0] * (len(nib) & 1)
        seq4.extend((nn[i] << 4) | nn[i + 1] for i in range(0, len(nn), 2))
        cigar.extend((ln << 4) | op for ln, op in ops)
    b = dict(contig=np.asarray(contig, np.uint32), pos0=np.asarray(pos0, np.int32), flag=np.asarray(flag, np.uint32),
             seq_off=np.asarray(seq_off, np.uint64), seq_len=np.asarray(seq_len, np.uint32),
             cig_off=np.asarray(cig_off, np.uint64), n_cig=np.asarray(n_cig, np.uint32),
             seq4=np.frombuffer(bytes(seq4) + b"\0" * 32, np.uint8).copy(), cigar=np.asarray(cigar + [0, 0], np.uint32),
             contig_names=np.asarray(["f%d" % i for i in range(len(contig_lens))]),
             contig_lens=np.asarray(contig_lens, np.uint32))
    if sort and n_reads:
        key = b["contig"].astype(np.int64) * (1 << 32) + np.maximum(b["pos0"].astype(np.int64), 0)
        o = np.argsort(key, kind="stable")
        for k in ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig"):
            b[k] = b[k][o]
    return b


def mixed_batch(rng, L, n_reads, piled=False):
    """VALID reads of very mixed shapes on one contig of L sites, coordinate-sorted, at whatever depth n_reads gives: plain reads of
    2 .. 400 bases and a few of 3000 .. 6000, leading clips of up to 260 bases, CIGARs of up to 16 ops with deletions of up to 5000
    sites and N skips -- the shapes k_window's tile lists carry or do not carry (kd_window.h: window-relative start, length, reach
    of the leading clip and CIGAR word count have bounded fields), at depths where its deep-tile and queue regimes switch.
    piled: all reads start in the first eighth of the contig."""
    contig, pos0, flag, seq_off, seq_len, cig_off, n_cig = [], [], [], [], [], [], []
    seq4, cigar = bytearray(), []
    starts = np.sort(rng.integers(0, max(1, L // 8) if piled else L, n_reads))
    for i in range(n_reads):
        r = rng.random()
        if r < 0.55:
            ops = [(int(rng.integers(2, 400)) if rng.random() < 0.97 else int(rng.integers(3000, 6000)), 0)]
        elif r < 0.75:
            lead = int(rng.integers(1, 30)) if rng.random() < 0.8 else int(rng.integers(120, 260))
            ops = [(lead, 4), (int(rng.integers(5, 300)), 0)]
            if rng.random() < 0.5:
                ops.append((int(rng.integers(1, 40)), 4))
        else:
            ops, k = [], int(rng.integers(2, 9))
            for j in range(k):
                ops.append((int(rng.integers(1, 120)), [0, 7, 8][int(rng.integers(0, 3))]))
                t = rng.random()
                if j < k - 1:
                    if t < 0.4:
                        ops.append((int(rng.integers(1, 6)), 1))
                    elif t < 0.8:
                        ops.append((int(rng.integers(1, 30)) if rng.random() < 0.9 else int(rng.integers(1000, 5000)), 2))
                    elif t < 0.9:
                        ops.append((int(rng.integers(1, 200)), 3))
            ops = ops[:16]
            if ops[-1][1] in (1, 2, 3):
                ops[-1] = (int(rng.integers(1, 50)), 0)
        qlen = sum(ln for ln, op in ops if op in (0, 1, 4, 7, 8))
        rlen = sum(ln for ln, op in ops if op in (0, 2, 3, 7, 8))
        p = int(starts[i])
        if rlen > L:
            ops, qlen, rlen, p = [(min(L, 50), 0)], min(L, 50), min(L, 50), 0
        if p + rlen > L:
            p = L - rlen
        nib = rng.choice([1, 2, 4, 8, 15], qlen, p=[0.24, 0.24, 0.24, 0.24, 0.04])
        contig.append(0); pos0.append(p); flag.append(0)
        seq_off.append(len(seq4)); seq_len.append(qlen); cig_off.append(len(cigar)); n_cig.append(len(ops))
        nn = [int(x) for x in nib] + [0] * (qlen & 1)
        seq4.extend((nn[k] << 4) | nn[k + 1] for k in range(0, len(nn), 2))
        cigar.extend((ln << 4) | op for ln, op in ops)
    b = dict(contig=np.asarray(contig, np.uint32), pos0=np.asarray(pos0, np.int32), flag=np.asarray(flag, np.uint32),
             seq_off=np.asarray(seq_off, np.uint64), seq_len=np.asarray(seq_len, np.uint32),
             cig_off=np.asarray(cig_off, np.uint64), n_cig=np.asarray(n_cig, np.uint32),
             seq4=np.frombuffer(bytes(seq4) + b"\0" * 32, np.uint8).copy(), cigar=np.asarray(cigar + [0, 0], np.uint32),
             contig_names=np.asarray(["f0"]), contig_lens=np.asarray([L], np.uint32))
    o = np.argsort(np.maximum(b["pos0"].astype(np.int64), 0), kind="stable")
    for k in ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig"):
        b[k] = b[k][o]
    return b


def oracle_outcome(batch):
    """-> ("ok", {cid: OracleAln}) or ("raise", ExceptionType) following the reference's contig-major order"""
    from oracle import oracle as ko
    try:
        return "ok", {cid: ko.parse_records(batch, cid) for cid in ko.contig_order(batch)}
    except (KeyError, IndexError, RuntimeError) as e:
        return "raise", type(e)


def check_engine(lib, batch, mode, window=64, slice_reads=0):
    """Engine vs oracle: identical tables / insertion dicts / consensus, or an exception where the oracle raises.
    When several reads are invalid the engine reports the first in batch order, the reference the first in
    contig-major order, so only the fact of raising is compared then."""
    from tests import parity as P
    kind, val = oracle_outcome(batch)
    if kind == "raise":
        try:
            P.Run(lib, batch, mode=mode, window=window, slice_reads=slice_reads)
        except (KeyError, IndexError, RuntimeError):
            return "raise"
        raise AssertionError("oracle raised %s, engine did not" % val.__name__)
    run = P.Run(lib, batch, mode=mode, window=window, slice_reads=slice_reads)
    P.assert_matches_oracle(run)
    return "ok"
