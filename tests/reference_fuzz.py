"""TEST INFRASTRUCTURE: end-to-end differential fuzz against the UNMODIFIED reference (round 5).  Random SAM files -> the
reference's bam_to_consensus (oracle/refrun.py: /root/reference, or its bytecode under oracle/_ref) vs
kindel_amd.kindel.bam_to_consensus: record names, sequences, change lists, report text, or the exception type -- default and
realign=True (the CDR path, a host port), several min_depth / min_overlap / clip_decay_threshold / mask_ends / trim / uppercase
settings.  `python -m tests.reference_fuzz N SEED0` runs a local campaign on the kernel emulator (3 300 files ran clean in round 5; `... N SEED0 structured`: files with clip-dominant regions, 1 460 ran clean);
tests/test_reference_fuzz.py keeps a few dozen seeds in the CPU suite."""
import os
import random
import tempfile


def rand_sam(rng, n_contigs, realistic):
    lens = [rng.randint(60, 400) for _ in range(n_contigs)]
    refs = ["".join(rng.choice("ACGT") for _ in range(L)) for L in lens]
    txt = "@HD\tVN:1.6\tSO:unsorted\n" + "".join("@SQ\tSN:c%d\tLN:%d\n" % (i, L) for i, L in enumerate(lens))
    n_reads = rng.randint(5, 400)
    hot = [(rng.randrange(n_contigs), rng.randint(10, 50)) for _ in range(3)]    # places where many reads clip (CDRs for realign)
    for n in range(n_reads):
        c = rng.randrange(n_contigs); L = lens[c]
        rl = rng.randint(20, 90)
        pos = rng.randint(0, max(0, L - rl))
        ops = []; seq = []; r = pos; left = rl
        if rng.random() < 0.35:      # leading clip
            k = rng.randint(1, 25); ops.append("%dS" % k); seq.append("".join(rng.choice("ACGT") for _ in range(k)))
        while left > 0 and r < L:
            m = min(left, rng.randint(3, 40), L - r)
            if m <= 0: break
            s = list(refs[c][r:r + m])
            for j in range(len(s)):
                if rng.random() < 0.04: s[j] = rng.choice("ACGTN")
            ops.append("%d%s" % (m, rng.choice("M=X") if not realistic else "M")); seq.append("".join(s)); r += m; left -= m
            u = rng.random()
            if u < 0.15 and r < L - 3:
                k = rng.randint(1, 4); ops.append("%dI" % k); seq.append("".join(rng.choice("ACGT") for _ in range(k)))
            elif u < 0.3 and r < L - 6:
                k = rng.randint(1, 5); ops.append("%dD" % k); r += k
            elif u < 0.33: ops.append("%dN" % rng.randint(1, 9))
        if rng.random() < 0.35:      # trailing clip
            k = rng.randint(1, 25); ops.append("%dS" % k); seq.append("".join(rng.choice("ACGT") for _ in range(k)))
        s = "".join(seq)
        if len(s) < 2: continue
        flag = rng.choice([0, 16, 0, 0, 256, 2048, 4 if rng.random() < 0.3 else 0])
        txt += "r%d\t%d\tc%d\t%d\t60\t%s\t*\t0\t0\t%s\t*\n" % (n, flag, c, pos + 1, "".join(ops), s)
    return txt

def structured_sam(rng, n_contigs=None, sort=True):
    """Files on which --realign DOES something: each contig's sample carries one to three novel segments (an insertion or a
    replacement of 0 - 60 reference bases by 15 - 90 new ones); reads tile the SAMPLE, and a read across a junction is aligned by
    its larger flank with the rest soft-clipped, as a local aligner reports it -- clip-dominant regions with overlapping
    clip consensuses from both sides (kindel.py:283-478)."""
    n_contigs = n_contigs or rng.randint(1, 3)
    txt_sq, recs = "", []
    for c in range(n_contigs):
        L = rng.randint(600, 5000)
        ref = "".join(rng.choice("ACGT") for _ in range(L))
        txt_sq += "@SQ\tSN:c%d\tLN:%d\n" % (c, L)
        # sample = list of (sample_start, length, ref_start or None)
        cuts = sorted(rng.sample(range(150, L - 150), rng.randint(1, 3)))
        cuts = [p for i, p in enumerate(cuts) if i == 0 or p - cuts[i - 1] > 250]
        segs, sample, r = [], [], 0
        for p in cuts:
            segs.append((len(sample), p - r, r)); sample.extend(ref[r:p])
            m = rng.randint(15, 90)
            segs.append((len(sample), m, None)); sample.extend(rng.choice("ACGT") for _ in range(m))
            r = p + rng.choice([0, 0, rng.randint(1, 60)])
        segs.append((len(sample), L - r, r)); sample.extend(ref[r:])
        sample = "".join(sample)
        depth, rl = rng.randint(8, 40), rng.randint(60, 150)
        n_reads = max(5, len(sample) * depth // rl)
        for n in range(n_reads):
            a = rng.randint(0, max(0, len(sample) - rl)); b = min(len(sample), a + rl)
            parts = []          # (length, ref_start or None) of the read's pieces
            for s0, ln, rs in segs:
                lo, hi = max(a, s0), min(b, s0 + ln)
                if lo < hi:
                    parts.append((hi - lo, None if rs is None else rs + lo - s0))
            best = max((k for k in range(len(parts)) if parts[k][1] is not None), key=lambda k: parts[k][0], default=None)
            if best is None or parts[best][0] < 12:
                continue
            lead = sum(x for x, _ in parts[:best]); trail = sum(x for x, _ in parts[best + 1:])
            seq = list(sample[a:b])
            for j in range(len(seq)):
                if rng.random() < 0.01: seq[j] = rng.choice("ACGT")
            cigar = ("%dS" % lead if lead else "") + "%dM" % parts[best][0] + ("%dS" % trail if trail else "")
            recs.append((c, parts[best][1], "r%d_%d\t0\tc%d\t%d\t60\t%s\t*\t0\t0\t%s\t*\n" % (c, n, c, parts[best][1] + 1, cigar, "".join(seq))))
    if sort:
        recs.sort(key=lambda t: (t[0], t[1]))
    else:
        rng.shuffle(recs)
    return "@HD\tVN:1.6\tSO:%s\n" % ("coordinate" if sort else "unsorted") + txt_sq + "".join(t[2] for t in recs)


def check_structured_seed(R, K, seed):
    """structured_sam through the reference and kindel_amd with realign=True (and the default) -> None or the first difference;
    also says whether realign changed the sequence (so a campaign can count the files that exercised the patches)."""
    rng = random.Random(seed)
    txt = structured_sam(rng, sort=bool(seed % 3))
    with tempfile.NamedTemporaryFile("w", suffix=".sam", delete=False) as fh:
        fh.write(txt)
        path = fh.name
    try:
        outs = []
        for kw in (dict(), dict(realign=True), dict(realign=True, min_overlap=rng.choice([5, 9, 12]), clip_decay_threshold=rng.choice([0.1, 0.05]), mask_ends=rng.choice([0, 50]))):
            a, b = outcome(R.bam_to_consensus, path, kw), outcome(K.bam_to_consensus, path, kw)
            if a != b:
                return "seed %d %r: differ (reference %s, kindel_amd %s)" % (seed, kw, a[0] if a[0] == "ok" else a, b[0] if b[0] == "ok" else b), False
            outs.append(a)
        return None, outs[0][0] == "ok" and outs[1][0] == "ok" and outs[0][1] != outs[1][1]
    finally:
        os.unlink(path)


def wild_sam(rng):
    nc = rng.randint(1, 3)
    lens = [rng.randint(40, 120) for _ in range(nc)]
    txt = "@HD\tVN:1.6\tSO:unsorted\n" + "".join("@SQ\tSN:c%d\tLN:%d\n" % (i, L) for i, L in enumerate(lens))
    wildness = rng.choice([0.0, 0.0, 0.02, 0.1])
    for n in range(rng.randint(1, 40)):
        c = rng.randrange(nc); L = lens[c]
        ops = []
        for _ in range(rng.randint(1, 7)):
            ops.append((rng.choice([0, 1, 1, 2, 3, 5, 8]), rng.choice("MMMMMIDSHNP=X")))
        q = sum(l for l, o in ops if o in "MIS=X")
        cigar = "".join("%d%s" % x for x in ops) if rng.random() > wildness * 0.3 else "*"
        qlen = q if rng.random() > wildness else rng.randint(0, 30)
        seq = "".join(rng.choice("ACGTNacgtRY=") if rng.random() < wildness * 0.3 else rng.choice("ACGT") for _ in range(qlen)) or "*"
        pos = rng.randint(1, max(1, L - 30)) if rng.random() > wildness else rng.choice([0, 1, L, L + 1, L + 5, rng.randint(1, L)])
        flag = rng.choice([0, 0, 0, 16, 4, 256, 2048])
        rname = "c%d" % c if rng.random() > wildness * 0.3 else "*"
        txt += "r%d\t%d\t%s\t%d\t60\t%s\t*\t0\t0\t%s\t*\n" % (n, flag, rname, pos, cigar, seq)
    return txt


def check_wild_seed(R, K, seed):
    """One WILD file (ops of length 0, H / N / P anywhere, clips in the middle, POS 0 / at and behind the contig's end, SEQ shorter or
    longer than the CIGAR, '*' CIGARs and RNAMEs, IUPAC and '=' bases) under three settings: the same result or the same exception type.
    The one documented divergence is tolerated: an insertion with a letter outside the BAM alphabet is refused (OSError)."""
    rng = random.Random(seed)
    txt = wild_sam(rng)
    with tempfile.NamedTemporaryFile("w", suffix=".sam", delete=False) as fh:
        fh.write(txt)
        path = fh.name
    try:
        for kw in (dict(), dict(min_depth=2, trim_ends=True), dict(realign=True, mask_ends=rng.choice([0, 3, 50]), min_overlap=rng.choice([3, 7]))):
            a, b = outcome(R.bam_to_consensus, path, kw), outcome(K.bam_to_consensus, path, kw)
            if a != b and not (b == ("raise", "OSError") and a[0] == "ok"):
                return "seed %d %r: reference %s, kindel_amd %s" % (seed, kw, a if a[0] == "raise" else "ok", b if b[0] == "raise" else "ok")
    finally:
        os.unlink(path)
    return None


def outcome(fn, path, kw):
    try:
        res = fn(path, **kw)
        return ("ok", [(c.name, c.sequence) for c in res.consensuses], {k: list(v) for k, v in res.refs_changes.items()}, dict(res.refs_reports))
    except Exception as e:      # noqa: BLE001 -- the exception TYPE is what is compared
        return ("raise", type(e).__name__)


def settings(rng, seed):
    return (dict(), dict(realign=True),
            dict(realign=True, min_depth=2, min_overlap=rng.choice([5, 7, 9]), clip_decay_threshold=rng.choice([0.1, 0.2, 0.05]),
                 mask_ends=rng.choice([0, 5, 50]), trim_ends=True, uppercase=True),
            dict(min_depth=rng.choice([0, 2, 5]), trim_ends=bool(seed & 2), uppercase=bool(seed & 4)))


def check_seed(R, K, seed, keep_dir=None):
    """One random file through both implementations under four settings -> None, or a description of the first difference."""
    rng = random.Random(seed)
    txt = rand_sam(rng, rng.randint(1, 3), realistic=bool(seed & 1))
    with tempfile.NamedTemporaryFile("w", suffix=".sam", delete=False) as fh:
        fh.write(txt)
        path = fh.name
    try:
        for kw in settings(rng, seed):
            a, b = outcome(R.bam_to_consensus, path, kw), outcome(K.bam_to_consensus, path, kw)
            if a != b:
                if keep_dir:
                    open(os.path.join(keep_dir, "reference_fuzz_fail_%d.sam" % seed), "w").write(txt)
                what = "outcome"
                if a[0] == "ok" and b[0] == "ok":
                    what = "sequences" if a[1] != b[1] else "changes" if a[2] != b[2] else "report"
                return "seed %d %r: %s differ (reference %s, kindel_amd %s)" % (seed, kw, what, a[0] if a[0] == "ok" else a, b[0] if b[0] == "ok" else b)
    finally:
        os.unlink(path)
    return None


def _df_diff(a, b):
    import numpy as np
    if list(a.columns) != list(b.columns) or len(a) != len(b):
        return "shape / columns"
    for c in a.columns:
        x, y = a[c].to_numpy(), b[c].to_numpy()
        if x.dtype.kind == "f" or y.dtype.kind == "f":
            ok = np.allclose(x.astype(float), y.astype(float), rtol=0, atol=1e-12, equal_nan=True)     # (float columns: 1e-12 absolute)
        elif x.dtype == object or y.dtype == object:
            ok = (x.astype(str) == y.astype(str)).all()
        else:
            ok = np.array_equal(x, y)
        if not ok:
            return "column %s" % c
    return None


def check_api_seed(R, K, seed):
    """The rest of the Python API on one random file: weights() (three option sets; integer columns exact, float columns to 1e-12),
    features() (single-contig files: the reference's handles no more), parse_bam()'s alignment fields -> None or the first difference."""
    import numpy as np
    rng = random.Random(seed)
    single = bool(seed % 3)
    txt = rand_sam(rng, 1 if single else rng.randint(1, 3), realistic=bool(seed & 1))
    with tempfile.NamedTemporaryFile("w", suffix=".sam", delete=False) as fh:
        fh.write(txt)
        path = fh.name

    def both(name, *a, **kw):
        out = []
        for M in (R, K):
            try:
                out.append(("ok", getattr(M, name)(*a, **kw)))
            except Exception as e:      # noqa: BLE001
                out.append(("raise", type(e).__name__))
        return out
    try:
        for kw in (dict(), dict(relative=True), dict(confidence=False)):
            a, b = both("weights", path, **kw)
            if a[0] != b[0] or (a[0] == "raise" and a != b):
                return "seed %d weights %r: %s vs %s" % (seed, kw, a[:2] if a[0] == "raise" else "ok", b[:2] if b[0] == "raise" else "ok")
            if a[0] == "ok":
                d = _df_diff(a[1], b[1])
                if d:
                    return "seed %d weights %r: %s" % (seed, kw, d)
        if single:
            a, b = both("features", path)
            if a[0] != b[0] or (a[0] == "raise" and a != b):
                return "seed %d features: outcomes differ" % seed
            if a[0] == "ok" and _df_diff(a[1], b[1]):
                return "seed %d features: %s" % (seed, _df_diff(a[1], b[1]))
        a, b = both("parse_bam", path)
        if a[0] != b[0] or (a[0] == "raise" and a != b):
            return "seed %d parse_bam: outcomes differ" % seed
        if a[0] == "ok":
            if list(a[1]) != list(b[1]):
                return "seed %d parse_bam: contigs %r vs %r" % (seed, list(a[1]), list(b[1]))
            for cid in a[1]:
                x, y = a[1][cid], b[1][cid]
                for f in x._fields:
                    u, v = getattr(x, f), getattr(y, f)
                    if f == "ref_id":
                        same = u == v
                    elif f in ("weights", "clip_start_weights", "clip_end_weights", "insertions"):
                        same = [dict(d) for d in u] == [dict(d) for d in v]
                    else:
                        same = np.asarray(u).tolist() == np.asarray(v).tolist()
                    if not same:
                        return "seed %d parse_bam %s.%s differs" % (seed, cid, f)
    finally:
        os.unlink(path)
    return None


if __name__ == "__main__":
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    from kindel_amd import _native as N
    N._default = N.Library(g.build_emu())
    from kindel_amd import kindel as K
    from oracle import refrun
    R = refrun.load_reference()
    n, s0 = (int(sys.argv[1]) if len(sys.argv) > 1 else 300), (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    if len(sys.argv) > 3 and sys.argv[3] == "structured":
        import logging
        logging.disable(logging.WARNING)
        patched = 0
        for seed in range(s0, s0 + n):
            d, did = check_structured_seed(R, K, seed)
            patched += bool(did)
            if d:
                bad += 1
                print("DIFF", d, flush=True)
        print("structured reference fuzz done:", n, "files,", patched, "changed by realign, differences:", bad, flush=True)
        sys.exit(0)
    for seed in range(s0, s0 + n):
        d = check_seed(R, K, seed, keep_dir=tempfile.gettempdir())
        if d:
            bad += 1
            print("DIFF", d, flush=True)
    print("reference fuzz done:", n, "files, differences:", bad, flush=True)
