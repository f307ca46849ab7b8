"""Shared parity helpers: run a batch through a C-ABI library and compare with the oracle / goldens."""
import hashlib
import json
import os
import tempfile

import numpy as np

from kindel_amd import _native as N
from oracle import oracle as ko
from oracle import samio_py

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF_TESTS = "/root/reference/tests"


def golden_outputs():
    with open(os.path.join(GOLD, "reference_outputs.json")) as fh:
        return json.load(fh)


def golden_quirks():
    with open(os.path.join(GOLD, "quirks.json")) as fh:
        return json.load(fh)


def fixture_keys():
    return sorted(k[:-4] for k in os.listdir(os.path.join(GOLD, "fixtures")) if k.endswith(".npz"))


def load_fixture(key):
    return dict(np.load(os.path.join(GOLD, "fixtures", key + ".npz")))


def subset(batch, n0, n1):
    out = dict(batch)
    for k in ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig"):
        out[k] = batch[k][n0:n1]
    return out


def sam_to_batch(sam_text):
    with tempfile.NamedTemporaryFile("w", suffix=".sam", delete=False) as fh:
        fh.write(sam_text)
        path = fh.name
    try:
        _, refs, recs = samio_py.read_alignment_file(path)
    finally:
        os.unlink(path)
    return samio_py.records_to_batch(refs, recs)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sha_s(s):
    return hashlib.sha256(s.encode()).hexdigest()


def ins_digest(items):
    canon = sorted((int(p), s, int(c)) for p, s, c in items)
    return hashlib.sha256(json.dumps(canon).encode()).hexdigest()


def changes_str(ch):
    return "".join("." if c == 0 else chr(c) for c in ch)


class Run:
    """One engine pass over a batch: tables, insertion dicts, consensus per contig."""

    def __init__(self, lib, batch, mode=N.KD_MODE_AUTO, window=0, slice_reads=0, min_depth=1, device=0,
                 n_pushes=1, shard=None):
        self.batch = batch
        self.order = ko.contig_order(batch)
        eng = N.Engine(batch["contig_lens"], device=device, lib=lib, mode=mode)
        try:
            if window or slice_reads:
                eng.set_tuning(window, slice_reads)
            if shard is not None:
                eng.set_shard(*shard)
            n = len(batch["contig"])
            cuts = np.linspace(0, n, n_pushes + 1).astype(int)
            for a, b in zip(cuts[:-1], cuts[1:]):
                eng.push(subset(batch, a, b))
            self.info = eng.batch_info()
            eng.finalize()
            self.stats = eng.stats()
            eng.consensus_run(min_depth)
            self.tables, self.ins, self.cns = {}, {}, {}
            for cid in self.order:
                self.tables[cid] = eng.tables(cid)
                site, count, strings = eng.insertions(cid)
                self.ins[cid] = [(int(p), s, int(c)) for p, c, s in zip(site, count, strings)]
                self.cns[cid] = eng.consensus_fetch(cid)
        finally:
            eng.close()


def _same(got, want, tag):
    """np.array_equal with a failure message that NAMES the cells: an intermittent difference must leave its address behind."""
    got, want = np.asarray(got), np.asarray(want)
    if got.shape == want.shape and np.array_equal(got, want):
        return
    if got.shape != want.shape:
        raise AssertionError("%s: shape %s vs %s" % (tag, got.shape, want.shape))
    idx = np.argwhere(got != want)
    cells = [(tuple(int(x) for x in i), int(got[tuple(i)]), int(want[tuple(i)])) for i in idx[:16]]
    raise AssertionError("%s: %d of %d cells differ; (index, engine, oracle): %s; index range %s .. %s" % (
        tag, len(idx), got.size, cells, idx.min(axis=0).tolist(), idx.max(axis=0).tolist()))


def assert_matches_oracle(run, min_depth=1, what=""):
    for cid in run.order:
        oa = ko.parse_records(run.batch, cid)
        t, L = run.tables[cid], oa.L
        tag = "%s contig %d: " % (what, cid)
        _same(t[0:5, :L].T, oa.weights, tag + "weights [site, A T G C N]")
        assert not t[0:5, L].any(), tag + "weights slot L must stay empty"
        _same(t[5], oa.deletions, tag + "deletions")
        _same(t[6:11, :L].T, oa.clip_start_weights, tag + "clip_start_weights")
        _same(t[11:16, :L].T, oa.clip_end_weights, tag + "clip_end_weights")
        _same(t[16], oa.clip_starts, tag + "clip_starts")
        _same(t[17], oa.clip_ends, tag + "clip_ends")
        _same(t[18], oa.ins_totals, tag + "insertion totals")
        if sorted(run.ins[cid]) != sorted(oa.insertions):
            a, b = set(run.ins[cid]), set(oa.insertions)
            raise AssertionError(tag + "insertion dicts: engine only %s, oracle only %s" % (sorted(a - b)[:8], sorted(b - a)[:8]))
        seq, ch, mm, _ = run.cns[cid]
        oseq, och = oa.consensus_sequence(min_depth=min_depth)
        if seq.decode() != oseq:
            s_ = seq.decode()
            k = next((i for i, (x, y) in enumerate(zip(s_, oseq)) if x != y), min(len(s_), len(oseq)))
            raise AssertionError(tag + "consensus: lengths %d / %d, first difference at byte %d: %r vs %r" % (len(s_), len(oseq), k, s_[k:k + 12], oseq[k:k + 12]))
        assert [None if c == 0 else chr(c) for c in ch] == och, tag + "changes"
        assert mm == oa.depth_minmax(), tag + "depth min/max %s vs %s" % (mm, oa.depth_minmax())
    return True


def assert_matches_golden(run, key, golden):
    """Compare with what the unmodified reference produced for fixture `key` (tests/golden)."""
    names = [str(x) for x in run.batch["contig_names"]]
    recs = {c["name"]: c for c in golden[key]["contigs"]}
    assert [names[c] for c in run.order] == [c["name"] for c in golden[key]["contigs"]], "contig order"
    for cid in run.order:
        g = recs[names[cid]]
        t, L = run.tables[cid], g["L"]
        assert sha(np.ascontiguousarray(t[0:5, :L].T)) == g["sha"]["weights"], key + " weights sha"
        assert sha(np.ascontiguousarray(t[6:11, :L].T)) == g["sha"]["clip_start_weights"]
        assert sha(np.ascontiguousarray(t[11:16, :L].T)) == g["sha"]["clip_end_weights"]
        assert sha(t[16]) == g["sha"]["clip_starts"] and sha(t[17]) == g["sha"]["clip_ends"]
        assert sha(t[5]) == g["sha"]["deletions"], key + " deletions sha"
        assert ins_digest(run.ins[cid]) == g["sha"]["insertions"], key + " insertions"
        assert int(t[0:5].sum()) == g["sums"]["weights"]
        seq, ch, mm, _ = run.cns[cid]
        assert len(seq) == g["consensus_len"] and sha_s(seq.decode()) == g["consensus_sha"], key + " consensus"
        assert sha_s(changes_str(ch)) == g["changes_sha"], key + " changes"
        assert list(mm) == g["depth_minmax"]


def quirk_expect(entry):
    return {"KeyError": KeyError, "IndexError": IndexError, "RuntimeError": RuntimeError}.get(entry.get("raises"))


def expected_variants(batch, abs_threshold=1, rel_threshold=0.01, only_variants=True):
    """Brute-force restatement of kindel_amd.kindel.variants() over the ORACLE's tables for `batch`
    -> list of (chrom, pos, ref, alt, type, count, depth) in output order."""
    out = []
    names = [str(x) for x in batch["contig_names"]]
    for cid in ko.contig_order(batch):
        oa = ko.parse_records(batch, cid)
        ins = {}
        for p_, s_, c in oa.insertions:
            ins.setdefault(int(p_), {})[s_] = ins.get(int(p_), {}).get(s_, 0) + int(c)
        for i in range(oa.L):
            five = dict(zip("ATGCN", (int(x) for x in oa.weights[i])))
            d = int(oa.deletions[i])
            depth = sum(five.values()) + d
            ref = "N"
            if sum(five.values()):
                ref = [k for k in "ATGCN" if five[k] == max(five.values())][0]
            rows = []
            for nt in "ACGT":
                k = five[nt]
                if depth > 0 and k >= max(abs_threshold, 1) and k >= rel_threshold * depth and not (only_variants and nt == ref):
                    rows.append((names[cid], i + 1, ref, nt, "snv", k, depth))
            if d >= max(abs_threshold, 1) and d >= rel_threshold * max(depth, 1):
                rows.append((names[cid], i + 1, ref, "-", "del", d, depth))
            for s_, k in ins.get(i, {}).items():
                if k >= max(abs_threshold, 1) and k >= rel_threshold * max(depth, 1):
                    rows.append((names[cid], i + 1, ref, "+" + s_, "ins", k, depth))
            out += sorted(rows, key=lambda r: (r[4], r[3]))
    return out


def variants_rows(df):
    return [(r.chrom, int(r.pos), r.ref, r.alt, r.type, int(r.count), int(r.depth)) for r in df.itertuples(index=False)]


def check_fetch_all(lib, batch, min_depth=1):
    eng = N.Engine(batch["contig_lens"], lib=lib)
    try:
        eng.push(batch)
        eng.finalize()
        eng.consensus_run(min_depth)
        n = len(batch["contig_lens"])
        off0, _ = eng.consensus_offsets()
        buf = np.zeros(int(off0[-1]) + 8, np.uint8)
        chg = np.zeros(eng.total_sites(), np.uint8)
        off = eng.consensus_fetch_all_into(buf, chg)
        assert np.array_equal(off, off0 - off0[0])
        for c in range(n):
            seq, ch, _, _ = eng.consensus_fetch(c)
            assert buf[int(off[c]): int(off[c + 1])].tobytes() == seq, c
            b = eng.contig_base(c)
            assert np.array_equal(chg[b: b + len(ch)], ch), c
        with np.testing.assert_raises(Exception):
            eng.consensus_fetch_all_into(np.zeros(max(int(off[-1]) - 1, 0), np.uint8))
    finally:
        eng.close()


def check_as_shards(lib, host, world, dev="cpu", tb=None, window=0):
    """The multi-GPU decomposition on ONE device: `world` work-balanced contiguous intervals (partition_weighted), the
    shards run one after another -- each with its own shard-local context and the reads routed to it by their CIGAR
    footprints -- every shard's payload goes through gather() / assemble() exactly as the ranks' rows do, and the stitched
    consensus, change codes and depth ranges as well as every shard's tables inside its interval (and at its halo site)
    must equal the oracle's, bit for bit.  tb: the device-resident batch (torch) to push instead of the host arrays.
    -> the intervals."""
    from kindel_amd import shard
    from tools import synth
    src = tb if tb is not None else host
    lens = host["contig_lens"]
    base, S = shard.g_layout(lens)
    ivs = shard.partition_weighted(lens, src["contig"], src["pos0"], src["seq_len"], world)
    assert ivs[0][0] == 0 and ivs[-1][1] == S and all(a[1] == b[0] for a, b in zip(ivs, ivs[1:]))
    g_lo, g_hi = shard.footprints(lens, src)
    n_reads = len(host["contig"])
    stitched = {cid: np.zeros((N.KD_NCH, int(lens[cid]) + 1), np.uint32) for cid in range(len(lens))}
    halo, rows, seen = [], [], 0
    for r in range(world):
        keep = shard.reads_of_rank(lens, g_lo, g_hi, r, world, intervals=ivs)
        seen += int(keep.sum())
        sub = dict(src)
        for k in ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig"):
            sub[k] = src[k][keep].contiguous() if tb is not None else src[k][keep]
        eng = N.Engine(lens, lib=lib)
        try:
            if window:
                eng.set_tuning(window, 0)
            eng.set_shard(*ivs[r])
            if tb is not None:
                eng.push_device(synth.device_ptrs(sub), int(sub["contig"].numel()), tb["seq4_bytes"], tb["cigar_words"])
            else:
                eng.push(sub)
            eng.finalize()
            eng.consensus_run(1)
            lo, hi = ivs[r]
            for cid in range(len(lens)):
                c0, c1 = int(base[cid]), int(base[cid]) + int(lens[cid]) + 1
                a, b = max(c0, lo), min(c1, hi)
                if a >= b:
                    continue
                t = eng.tables(cid)
                stitched[cid][:, a - c0: b - c0] = t[:, a - c0: b - c0]
                if b < c1 and r + 1 < world:      # the halo site of this shard lies in the same contig: committed here as well
                    halo.append((cid, b - c0, t[:, b - c0].copy()))
            payload, _ = shard.gather(eng, ivs[r], dev)     # world 1: this rank's row, as the all-gather would carry it
            rows.append(np.ascontiguousarray(payload.cpu().numpy()[0]))
        finally:
            eng.close()
    assert seen <= 1.05 * n_reads + world * 4096      # a read is seen by the shards it touches: boundary reads twice, no more
    seqs, changes, minmax = shard.assemble(rows, lens, world, intervals=ivs)
    tot_w = 0
    for cid in ko.contig_order(host):
        oa = ko.parse_records(host, cid)
        t, L = stitched[cid], oa.L
        assert np.array_equal(t[0:5, :L].T, oa.weights) and np.array_equal(t[5], oa.deletions), cid
        assert np.array_equal(t[6:11, :L].T, oa.clip_start_weights) and np.array_equal(t[11:16, :L].T, oa.clip_end_weights)
        assert np.array_equal(t[16], oa.clip_starts) and np.array_equal(t[17], oa.clip_ends)
        assert np.array_equal(t[18], oa.ins_totals)
        oseq, och = oa.consensus_sequence()
        assert seqs[cid].decode() == oseq, cid
        assert [None if c == 0 else chr(c) for c in changes[cid]] == och
        assert minmax[cid] == oa.depth_minmax()
        for hc, site, col in halo:
            if hc == cid and site < L:
                assert np.array_equal(col[0:5], oa.weights[site]) and col[5] == oa.deletions[site], ("halo", cid, site)
        tot_w += int(t[0:5].sum())
    assert tot_w > 0
    return ivs


def long_read_cases():
    """name -> (SAM text, expected exception or None): oracle/quirk_cases.py: long_cases(), pinned against the unmodified
    reference in tests/golden/long_quirks.json (oracle/make_golden.py long)"""
    from oracle import quirk_cases
    return quirk_cases.long_cases()


def golden_long_quirks():
    with open(os.path.join(GOLD, "long_quirks.json")) as fh:
        return json.load(fh)


def assert_matches_long_golden(run, entry, what=""):
    """engine output == what the unmodified reference returned for the case (digests, oracle/make_golden.py: do_long_quirks)"""
    names = [str(x) for x in run.batch["contig_names"]]
    for rec in entry["contigs"]:
        cid = names.index(rec["name"])
        t, L = run.tables[cid], rec["L"]
        tag = "%s %s: " % (what, rec["name"])
        g = rec["sha"]
        assert sha(np.ascontiguousarray(t[0:5, :L].T)) == g["weights"], tag + "weights"
        assert sha(t[5]) == g["deletions"], tag + "deletions"
        assert sha(np.ascontiguousarray(t[6:11, :L].T)) == g["clip_start_weights"], tag + "clip_start_weights"
        assert sha(np.ascontiguousarray(t[11:16, :L].T)) == g["clip_end_weights"], tag + "clip_end_weights"
        assert sha(t[16]) == g["clip_starts"] and sha(t[17]) == g["clip_ends"], tag + "clip_starts / clip_ends"
        assert ins_digest(run.ins[cid]) == g["insertions"], tag + "insertion dicts"
        assert int(t[18].sum()) == rec["sums"]["ins_events"], tag + "insertion totals"
        seq, ch, _, _ = run.cns[cid]
        assert seq.decode() == rec["consensus"], tag + "consensus"
        assert changes_str(ch) == rec["changes"], tag + "changes"


def many_contigs_batch(n, seed=5):
    """More contigs than a 16-bit field or a 65 535-wide launch dimension holds (a fragmented assembly, a metagenome): n contigs of
    64 - 119 sites with one to three reads each -- 40M, 20M2I18M and 5S20M3D15M in turn, positions at random, sorted by (contig, pos).
    Built with numpy in one go (tools/synth.py loops over the contigs)."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(64, 120, n).astype(np.uint32)
    contig = np.repeat(np.arange(n, dtype=np.uint32), rng.integers(1, 4, n))
    m = len(contig)
    kind = rng.integers(0, 3, m)
    pos0 = (rng.random(m) * (lens[contig] - np.array([40, 38, 38])[kind] + 1)).astype(np.int32)
    order = np.lexsort((pos0, contig))
    contig, kind, pos0 = contig[order], kind[order], pos0[order]
    M, I, D, S = 0, 1, 2, 4
    shapes = [[40 << 4 | M], [20 << 4 | M, 2 << 4 | I, 18 << 4 | M], [5 << 4 | S, 20 << 4 | M, 3 << 4 | D, 15 << 4 | M]]
    n_cig = np.array([1, 3, 4], np.uint32)[kind]
    cig_off = np.concatenate([[0], np.cumsum(n_cig)[:-1]]).astype(np.uint64)
    cigar = np.zeros(int(n_cig.sum()) + 2, np.uint32)
    for k, words in enumerate(shapes):
        sel = np.nonzero(kind == k)[0]
        for j, w in enumerate(words):
            cigar[cig_off[sel] + np.uint64(j)] = w
    nib = np.array([1, 2, 4, 8], np.uint8)[rng.integers(0, 4, (m, 40))]
    seq4 = np.zeros(m * 20 + 24, np.uint8)
    seq4[: m * 20] = (nib[:, 0::2] << 4 | nib[:, 1::2]).reshape(-1)
    return dict(contig=contig, pos0=pos0, flag=np.zeros(m, np.uint32), seq_off=np.arange(m, dtype=np.uint64) * np.uint64(20),
                seq_len=np.full(m, 40, np.uint32), cig_off=cig_off, n_cig=n_cig, seq4=seq4, cigar=cigar, contig_lens=lens,
                seq4_bytes=m * 20, cigar_words=int(n_cig.sum()), contig_names=np.asarray(["c%d" % i for i in range(n)]))


def check_many_contigs(lib, n, world=3, sample=60):
    """many_contigs_batch(n) through kd_finish and the per-contig read-outs: a sample of contigs (the first, the last, the ones either
    side of 65 535 / 65 536, a random set) against the oracle -- tables, insertion dicts, consensus, change codes, depth range -- and
    the same batch as `world` shards stitched from their exchange rows against the one context's FASTA, every contig."""
    from kindel_amd import shard
    b = many_contigs_batch(n)
    lens = b["contig_lens"]
    eng = N.Engine(lens, lib=lib)
    try:
        out = np.zeros(int(lens.sum()) * 2 + 4096, np.uint8)
        eng.push(b)
        off = eng.finish(out)
        rng = np.random.default_rng(1)
        pick = sorted(set([0, n - 1] + [c for c in (65534, 65535, 65536, 65537) if c < n] + [int(x) for x in rng.integers(0, n, sample)]))
        for cid in pick:
            oa = ko.parse_records(b, cid)
            L, t = oa.L, eng.tables(cid)
            assert np.array_equal(t[0:5, :L].T, oa.weights) and np.array_equal(t[5], oa.deletions) and np.array_equal(t[18], oa.ins_totals), cid
            assert np.array_equal(t[6:11, :L].T, oa.clip_start_weights) and np.array_equal(t[16], oa.clip_starts), cid
            site, count, strings = eng.insertions(cid)
            assert sorted((int(p), s, int(c)) for p, c, s in zip(site, count, strings)) == sorted(oa.insertions), cid
            oseq, och = oa.consensus_sequence(min_depth=1)
            assert out[int(off[cid]): int(off[cid + 1])].tobytes().decode() == oseq, cid
            seq, ch, mm, _ = eng.consensus_fetch(cid)
            assert seq.decode() == oseq and [None if c == 0 else chr(c) for c in ch] == och and mm == oa.depth_minmax(), cid
        whole = [out[int(off[c]): int(off[c + 1])].tobytes() for c in range(n)]
    finally:
        eng.close()
    ivs = shard.partition_weighted(lens, b["contig"], b["pos0"], b["seq_len"], world)
    g_lo, g_hi = shard.footprints(lens, b)
    rows = []
    for r in range(world):
        keep = shard.reads_of_rank(lens, g_lo, g_hi, r, world, intervals=ivs)
        sub = dict(b)
        for k in ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig"):
            sub[k] = b[k][keep]
        eng = N.Engine(lens, lib=lib)
        try:
            eng.set_shard(*ivs[r])
            ex = shard.Exchange(eng, ivs[r], eng.memory_device, intervals=ivs).attach()
            eng.push(sub)
            eng.finish(out)
            rows.append(np.ascontiguousarray(ex.collect().cpu().numpy()[0]))
            ex.detach()
        finally:
            eng.close()
    seqs, _changes, _minmax = shard.assemble(rows, lens, world, intervals=ivs)
    assert seqs == whole
    return len(pick)


def check_huge_g_space(lib, depth=12, region=60000, seed=4):
    """Sites beyond 2^31 in G-space (a human-sized reference, 4.2 G sites in four contigs, can only be run as shards -- its tables are
    319 GB -- but every shard must be right wherever it lies): the reads of a `region`-site contig are moved to three places of the big
    layout -- straddling G-site 2^31, in the middle of the third contig (G ~ 3.3 G) and at the very end of the last one (the last sites
    of G-space) -- and each place is run as a shard of exactly that interval.  The oracle cannot hold a 1.1 G-site contig's tables;
    the check is translation: the shard's exchange row (consensus bytes, change codes, depth range) must equal the small contig's own
    consensus as the oracle computes it."""
    from kindel_amd import shard
    from tools import synth
    small = synth.to_numpy(synth.short_reads([region], depth, seed=seed))
    oa = ko.parse_records(small, 0)
    oseq, och = oa.consensus_sequence(min_depth=1)
    omm = oa.depth_minmax()
    lens = np.asarray([1_100_000_000, 1_100_000_000, 1_100_000_000, 900_000_000], np.uint32)
    base, S = shard.g_layout(lens)
    assert S > (1 << 32) - (1 << 27) and S < (1 << 32)
    places = [(1, (1 << 31) - int(base[1]) - region // 2), (2, 1_050_000_000), (3, int(lens[3]) - region)]
    for cid, off in places:
        lo = int(base[cid]) + off
        hi = lo + region + (1 if cid == 3 else 0)        # (the last contig's slot `len`: the end of the contig is the region's end)
        if cid == 1:
            assert lo < (1 << 31) < hi
        b = dict(small)
        b["contig_lens"] = lens
        b["contig"] = np.full_like(small["contig"], cid)
        b["pos0"] = (small["pos0"].astype(np.int64) + off).astype(np.int32)
        eng = N.Engine(lens, lib=lib)
        try:
            eng.set_shard(lo, hi)
            ex = shard.Exchange(eng, (lo, hi), eng.memory_device, pad=(2 * region + region // 8 + 65536) // 64 * 64).attach()
            out = np.zeros(2 * region + 4096, np.uint8)
            eng.push(b)
            coff = eng.finish(out)
            row = np.ascontiguousarray(ex.collect().cpu().numpy()[0])
            ex.detach()
        finally:
            eng.close()
        n = len(lens)
        assert out[int(coff[cid]): int(coff[cid + 1])].tobytes().decode() == oseq, ("bytes", cid)
        assert int(coff[n]) == len(oseq)
        need = int(row[:8].view(np.uint64)[0])
        rcoff = row[16: 16 + (n + 1) * 8].view(np.uint64)
        rmm = row[16 + (n + 1) * 8: 16 + (n + 1) * 8 + n * 8].view(np.uint32).reshape(n, 2)
        o = 16 + (n + 1) * 8 + n * 8
        sites = min(hi, S) - lo
        assert need == o + sites + len(oseq) and need <= len(row)
        ch = row[o: o + region]
        assert [None if c == 0 else chr(c) for c in ch] == och, ("changes", cid)
        assert row[o + sites + int(rcoff[cid]): o + sites + int(rcoff[cid + 1])].tobytes().decode() == oseq, ("row bytes", cid)
        assert int(rmm[cid][1]) == omm[1] and int(rmm[cid][0]) == omm[0], ("depth range", cid, rmm[cid], omm)
    return len(places)
