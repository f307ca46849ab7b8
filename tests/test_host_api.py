"""Host-side logic of kindel_amd.kindel / cli (report text, patch splicing, realign scans, DataFrames),
driven through the kernel emulator on CPU and compared with the reference's golden outputs."""
import io
import os
from contextlib import redirect_stderr, redirect_stdout

import numpy as np
import pytest

from tools import synth
from tests import parity as P

ROOT = P.ROOT

GOLD = P.golden_outputs()
QUIRKS = P.golden_quirks()


def _bam(tmp_path, key, n=None):
    b = P.load_fixture(key)
    p = str(tmp_path / (key + ".bam"))
    synth.write_bam(p, b if n is None else P.subset(b, 0, n), sort_order="unknown")
    return p


def test_consensus_helper_matches_reference_unit_test():
    """/root/reference/tests/test_kindel.py:25-32"""
    from kindel_amd import kindel as K
    w = {"A": 1, "C": 2, "G": 3, "T": 4, "N": 5}
    assert K.consensus(w) == ("N", 5, 0.33, False)
    assert K.consensus({"A": 5, "C": 5, "G": 3, "T": 4, "N": 1})[3] is True
    assert K.consensus({"A": 0, "T": 0, "G": 0, "C": 0, "N": 0}) == ("N", 0, 0, False)


def test_merge_by_lcs_matches_reference_unit_test():
    """/root/reference/tests/test_kindel.py:35-53"""
    from kindel_amd import kindel as K
    one = ("AACTGCCGCTAGGGGCGCGTTCGGGCTCGCCAACATCTTCAGTCCGGG",
           "GCCGCTAGGGGCGCGTTCGGGCTCGCCAACATCTTCAGTCCGGGCGCTAAGCAGAACA")
    two = ("AACTGCCGCTAGGGGCGCGTTCGGGCTCGCCAACATCTTCAGTCCGGGCGCTAAGCAGAACATC",
           "GCAGATACCTACACCACCGGGGGAACTGCCGCTAGGGGCGCGTTCGGGCTCGCCAACATCTTCAGTCCGGGCGCTAAGCAGAACA")
    want = "AACTGCCGCTAGGGGCGCGTTCGGGCTCGCCAACATCTTCAGTCCGGGCGCTAAGCAGAACA"
    assert K.merge_by_lcs(*one, min_overlap=7) == want
    assert K.merge_by_lcs(*two, min_overlap=7) == want
    assert K.merge_by_lcs("AT", "CG", min_overlap=7) is None


@pytest.mark.parametrize("key", ["ext__3.issue23.bc75", "ext__2.issue23.bc63", "minimap2__1.1.multi"])
def test_bam_to_consensus_default_and_realign(api_on_emu, tmp_path, key):
    from kindel_amd import kindel as K
    path = _bam(tmp_path, key)
    res = K.bam_to_consensus(path)
    res_r = K.bam_to_consensus(path, realign=True, min_overlap=7)
    assert [c.name for c in res.consensuses] == [g["name"] + "_cns" for g in GOLD[key]["contigs"]]
    for i, g in enumerate(GOLD[key]["contigs"]):
        assert res.consensuses[i].sequence == g["consensus"]
        assert "".join("." if c is None else c for c in res.refs_changes[g["name"]]) == g["changes"]
        assert res.refs_reports[g["name"]] == g["report"].replace("{bam_path}", path)
        assert res_r.consensuses[i].sequence == g["realign_consensus"]
        assert res_r.refs_reports[g["name"]] == g["realign_report"].replace("{bam_path}", path)


def test_realign_regions_on_issue23_debug(api_on_emu, tmp_path):
    """CDR scan over device tables vs the reference's cdrp_consensuses (golden, two mask_ends values)."""
    from kindel_amd import kindel as K
    key = "ext__1.issue23.debug"
    aln = list(K.parse_bam(_bam(tmp_path, key)).values())[0]
    g = GOLD[key]["contigs"][0]
    for me in (50, 10):
        got = K.cdrp_consensuses(aln.weights, aln.deletions, aln.clip_start_weights, aln.clip_end_weights,
                                 aln.clip_start_depth, aln.clip_end_depth, 0.1, me)
        assert [[list(r) for r in pair] for pair in got] == g["cdrps_0.1_%d" % me]
    res_r = K.bam_to_consensus(_bam(tmp_path, key), realign=True, min_overlap=7)
    assert res_r.consensuses[0].sequence == g["realign_consensus"]


def test_patch_splicing_matches_reference(api_on_emu, tmp_path):
    """consensus_sequence with cdr_patches (kindel.py:393-401), incl. overlapping / None / empty patches."""
    from kindel_amd import kindel as K
    pc = QUIRKS["__patches__"]
    p = tmp_path / "p.sam"
    p.write_text(pc["sam"])
    aln = list(K.parse_bam(str(p)).values())[0]
    for name, v in pc["sets"].items():
        patches = [K.Region(s, e, q, None) for s, e, q in v["patches"]]
        seq, ch = K.consensus_sequence(aln.weights, aln.insertions, aln.deletions, patches, False, 1, False)
        assert seq == v["consensus"], name
        assert "".join("." if c is None else c for c in ch) == v["changes"], name


@pytest.mark.parametrize("name", ["clip_both_ends_pair", "insertion_tie", "deletion_simple", "unmapped_placed_only_contig_all_N",
                                  "contig_order_first_appearance", "base_tie", "pos0_wraps_to_last_site"])
def test_quirk_reports_and_options(api_on_emu, tmp_path, name):
    from kindel_amd import kindel as K
    entry = QUIRKS[name]
    p = tmp_path / "q.sam"
    p.write_text(entry["sam"])
    res = K.bam_to_consensus(str(p))
    assert [c.name for c in res.consensuses] == entry["names"]
    assert list(res.refs_reports.values()) == [r.replace("{bam_path}", str(p)) for r in entry["reports"]]
    alns = K.parse_bam(str(p))
    for g in entry["contigs"]:
        aln = alns[g["name"]]
        assert [[d[c] for c in "ATGCN"] for d in aln.weights] == g["weights"]
        assert list(aln.deletions) == g["deletions"] and list(aln.clip_starts) == g["clip_starts"]
        assert [int(x) for x in aln.consensus_depth] == g["consensus_depth"]
        assert list(aln.clip_start_depth) == g["clip_start_depth"] and list(aln.clip_end_depth) == g["clip_end_depth"]
        got_ins = sorted([p_, s, c] for p_, d in enumerate(aln.insertions) for s, c in d.items())
        assert got_ins == sorted([p_, s.upper(), c] for p_, s, c in g["insertions"])
        for run in g["runs"]:
            seq, ch = K.consensus_sequence(aln.weights, aln.insertions, aln.deletions, None, run["trim_ends"],
                                           run["min_depth"], run["uppercase"])
            assert seq == run["consensus"]


@pytest.mark.parametrize("name", ["ERR_M_overhang_past_L", "ERR_iupac_in_M", "ERR_cigar_star_mapped"])
def test_reference_exceptions_surface(api_on_emu, tmp_path, name):
    from kindel_amd import kindel as K
    p = tmp_path / "e.sam"
    p.write_text(QUIRKS[name]["sam"])
    with pytest.raises(P.quirk_expect(QUIRKS[name])):
        K.bam_to_consensus(str(p))


def test_weights_dataframe_multi_contig(api_on_emu, tmp_path):
    from kindel_amd import kindel as K
    key = "minimap2__1.1.multi"
    path = _bam(tmp_path, key)
    for rel, tag in ((False, "abs"), (True, "rel")):
        df = K.weights(path, relative=rel)
        g = np.load(os.path.join(P.GOLD, "weights_%s_%s.npz" % (key, tag)), allow_pickle=True)
        assert list(df.columns) == [str(c) for c in g["columns"]]
        for c in df.columns:
            a, b = df[c].to_numpy(), g[c]
            if a.dtype.kind == "f":
                assert np.allclose(a, b, rtol=0, atol=1e-12, equal_nan=True), (tag, c)
            elif a.dtype == object:
                assert (a.astype(str) == b.astype(str)).all()
            else:
                assert np.array_equal(a, b), (tag, c)
    K.features(path)   # smoke, like the reference's test (and: no multi-contig crash)


@pytest.mark.parametrize("key", ["minimap2__1.1.multi", "ext__3.issue23.bc75"])
def test_variants_extension(api_on_emu, tmp_path, key):
    """variants() has no reference implementation (README.md:106 only): checked against a brute-force filter
    over the oracle's tables."""
    from kindel_amd import kindel as K
    path = _bam(tmp_path, key)
    for a, r, only in ((1, 0.01, True), (3, 0.2, True), (1, 0.0, False)):
        df = K.variants(path, abs_threshold=a, rel_threshold=r, only_variants=only)
        assert P.variants_rows(df) == P.expected_variants(P.load_fixture(key), a, r, only), (a, r, only)
        assert list(df.columns) == ["chrom", "pos", "ref", "alt", "type", "count", "depth", "frequency"]
        assert np.allclose(df["frequency"], np.round(df["count"] / df["depth"].clip(lower=1), 4))
    assert "frequency" not in K.variants(path, absolute=True).columns


def test_cli_in_process(api_on_emu, tmp_path):
    from kindel_amd import cli
    key = "ext__3.issue23.bc75"
    path = _bam(tmp_path, key)
    out, err = io.StringIO(), io.StringIO()
    with redirect_stdout(out), redirect_stderr(err):
        assert cli.main(["consensus", path]) == 0
    g = GOLD[key]["contigs"][0]
    assert out.getvalue() == ">%s_cns\n%s\n" % (g["name"], g["consensus"])
    assert err.getvalue().startswith("========================= REPORT")
    out = io.StringIO()
    with redirect_stdout(out):
        assert cli.main(["version"]) == 0
    assert out.getvalue().strip() == "kindel 1.2.1"
    a = cli.build_parser().parse_args(["consensus", "x.bam"])
    assert (a.realign, a.min_depth, a.min_overlap, a.clip_decay_threshold, a.mask_ends, a.trim_ends, a.uppercase) == \
        (False, 1, 7, 0.1, 50, False, False)   # cli.py:11-18 defaults


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` without a launcher spawns its own ranks -- and refuses (non-zero, nothing on stdout) when
    fewer than N GPUs are visible instead of silently benchmarking one."""
    import subprocess
    import sys
    import torch
    n = torch.cuda.device_count() + 1 if torch.cuda.is_available() else 2
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(max(n, 2)), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0 and p.stdout.strip() == "" and "GPU(s) visible" in p.stderr


def test_parse_records_without_countable_records(api_on_emu):
    """parse_records('ref', 10, []) and an all-unmapped record list: the reference returns an all-zero alignment of ref_len
    sites (kindel.py:29-39, :43-46), not an error."""
    from kindel_amd import kindel as K

    class Rec:
        def __init__(self, pos, mapped, seq, cigars):
            self.pos, self.mapped, self.seq, self.cigars = pos, mapped, seq, cigars

    for recs in ([], [Rec(3, False, "ACGT", ((4, "M"),)), Rec(1, False, "*", ())]):
        aln = K.parse_records("ref", 10, recs)
        assert aln.ref_id == "ref" and len(aln.weights) == 10
        assert all(sum(w.values()) == 0 for w in aln.weights)
        assert aln.deletions == [0] * 11 and aln.clip_starts == [0] * 11 and aln.clip_ends == [0] * 11
        assert len(aln.insertions) == 11 and all(d == {} for d in aln.insertions)
        assert aln.clip_depth == [0] * 10 and list(aln.consensus_depth) == [0] * 10
        seq, changes = K.consensus_sequence(aln.weights, aln.insertions, aln.deletions, None, False, 1, False)
        assert seq == "N" * 10 and changes == ["N"] * 10


def test_cli_refuses_more_ranks_than_gpus(tmp_path):
    """`kindel consensus --gpus N` spawns one process per GPU -- and refuses when there are fewer GPUs than ranks."""
    import subprocess
    import sys
    import torch
    n = torch.cuda.device_count() + 1 if torch.cuda.is_available() else 2
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "KINDEL_DIST_BACKEND")}
    p = subprocess.run([sys.executable, "-m", "kindel_amd", "consensus", "--gpus", str(max(n, 2)), str(tmp_path / "x.bam")],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 2 and p.stdout == "" and "GPU(s) visible" in p.stderr


@pytest.mark.parametrize("header", ["@SQ\tSN:a\tLN:30\n@SQ\tSN:b\tLN:12\n", "@HD\tVN:1.6\n"])
def test_header_only_input_gives_an_empty_result(api_on_emu, tmp_path, header):
    """A header and no records (with and without @SQ lines), as a file and through a pipe: parse_bam returns {} (kindel.py:143-152) and
    bam_to_consensus an empty result -- there is no context to close (round 5's `finally: pl.engine.close()` raised AttributeError
    here), and weights() / variants() hand back empty frames."""
    import threading
    from kindel_amd import kindel as K
    p = tmp_path / "h.sam"
    p.write_text(header)
    for realign in (False, True):
        r = K.bam_to_consensus(str(p), realign=realign)
        assert r.consensuses == [] and r.refs_changes == {} and r.refs_reports == {}
    assert K.parse_bam(str(p)) == {}
    assert len(K.weights(str(p))) == 0 and len(K.variants(str(p))) == 0
    fifo = str(tmp_path / "in.fifo")
    os.mkfifo(fifo)

    def feed():
        with open(fifo, "w") as out:
            out.write(header)
    th = threading.Thread(target=feed, daemon=True)
    th.start()
    r = K.bam_to_consensus(fifo)
    th.join(timeout=30)
    assert not th.is_alive() and r.consensuses == []


def test_contexts_are_released_without_the_cycle_collector(api_on_emu, tmp_path):
    """weights() / features() / variants() / bam_to_consensus() close their context before they return; parse_bam()'s alignments keep
    theirs alive exactly as long as one of them lives -- by reference count, no cycle for the collector to find (round 5: the
    alignments referred back to a Pileup that cached them)."""
    import gc
    import weakref
    from kindel_amd import kindel as K
    from kindel_amd import _native as N
    p = _bam(tmp_path, "bwa_mem__1.1.sub_test", 400)
    made = []
    real_init = N.Engine.__init__

    def spy(self, *a, **k):
        real_init(self, *a, **k)
        made.append(weakref.ref(self))
    N.Engine.__init__ = spy
    gc.disable()
    try:
        K.weights(p); K.features(p); K.variants(p); K.bam_to_consensus(p, realign=True)
        assert len(made) == 4 and all(r() is None or r()._h is None for r in made)      # closed (or gone) on return
        alns = K.parse_bam(p)
        eng = made[-1]()
        assert eng is not None and eng._h is not None
        one = next(iter(alns.values()))
        seq, _ = K.consensus_sequence(one.weights, one.insertions, one.deletions, None, False, 1, False)     # needs the device tables
        assert len(seq) > 0
        del eng, one, alns
        assert made[-1]() is None                                                       # freed by reference count alone
    finally:
        gc.enable()
        N.Engine.__init__ = real_init


def _six_contig_bam(tmp_path):
    """Six contigs, two of them without a record, the others met in the file in the order 4, 0, 5, 2 (not the header's)."""
    from tools import synth
    lens = [5000, 3000, 9000, 2500, 4000, 7000]
    b = synth.to_numpy(synth.short_reads(lens, 12, seed=21))
    keep = np.flatnonzero((b["contig"] != 1) & (b["contig"] != 3))
    rank = np.asarray([1, 9, 3, 9, 0, 2])[b["contig"][keep]]
    keep = keep[np.argsort(rank, kind="stable")]
    sub = dict(b)
    for k in ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig"):
        sub[k] = b[k][keep]
    p = str(tmp_path / "six.bam")
    synth.write_bam(p, sub, sort_order="unsorted")
    return p, lens


@pytest.mark.parametrize("realign", [False, True])
def test_reference_that_does_not_fit_the_device_goes_in_groups_of_contigs(api_on_emu, tmp_path, monkeypatch, realign):
    """bam_to_consensus on a reference whose tables the device cannot hold at once (here: the emulator refuses any single allocation
    above KD_EMU_ALLOC_CAP, as hipMalloc refuses tables beyond the GPU's memory): the contigs in use go in groups, one streamed pass
    over the file each (the records of the other groups dropped by the stream's contig map), halved until a group fits -- the same
    result tuple as the one pass, contigs in order of first appearance; a contig that does not fit by itself is the MemoryError it
    was."""
    from kindel_amd import kindel as K
    from kindel_amd import _native as N
    path, lens = _six_contig_bam(tmp_path)
    want = K.bam_to_consensus(path, realign=realign, min_overlap=7)
    assert [c.name for c in want.consensuses] == ["ctg4_cns", "ctg0_cns", "ctg5_cns", "ctg2_cns"]
    table_bytes = lambda sites: N.KD_NCH * 4 * sites
    for cap_sites, passes in ((16000, 2), (9500, 4)):      # 25 000 sites in use: halves of ~12 500 fit the first cap, single contigs the second
        calls = []
        real = K.pileup_file
        monkeypatch.setattr(K, "pileup_file", lambda *a, **k: (calls.append(k.get("contigs")), real(*a, **k))[1])
        monkeypatch.setenv("KD_EMU_ALLOC_CAP", str(table_bytes(cap_sites)))
        got = K.bam_to_consensus(path, realign=realign, min_overlap=7)
        monkeypatch.delenv("KD_EMU_ALLOC_CAP")
        monkeypatch.setattr(K, "pileup_file", real)
        assert [(c.name, c.sequence) for c in got.consensuses] == [(c.name, c.sequence) for c in want.consensuses]
        assert got.refs_changes == want.refs_changes and got.refs_reports == want.refs_reports
        assert list(got.refs_reports) == list(want.refs_reports)          # (dict order: first appearance)
        assert calls[0] is None and len([c for c in calls if c is not None]) >= passes, calls
    monkeypatch.setenv("KD_EMU_ALLOC_CAP", str(table_bytes(8000)))      # ctg2 (9 000 sites) does not fit by itself
    with pytest.raises(MemoryError):
        K.bam_to_consensus(path, realign=realign)


@pytest.mark.parametrize("chunk", [0, 4096])
def test_pileup_of_chosen_contigs_drops_the_other_records(api_on_emu, tmp_path, chunk):
    """pileup_file(contigs=...): only those @SQ entries are laid out, the stream drops the records of the others (contig map value
    0xfffffffe) -- also when whole chunks of the file hold nothing but dropped records; the tables are those of the one pass."""
    from kindel_amd import kindel as K
    from tools import synth
    lens = [5000, 3000, 9000, 2500, 4000, 7000]
    p = str(tmp_path / "six_sorted.bam")
    synth.write_bam(p, synth.to_numpy(synth.short_reads(lens, 12, seed=21)), block_bytes=2000)
    with K.pileup_file(p) as full:
        for grp in ([5], [0], [2, 3], [1, 4, 5]):
            with K.pileup_file(p, contigs=grp, chunk_bytes=chunk) as pl:
                assert list(pl.names) == ["ctg%d" % c for c in grp] and (chunk == 0 or pl.ingest["batches"] > 20)
                for k, c in enumerate(grp):
                    assert np.array_equal(pl.tables(k), full.tables(c)), (chunk, grp, c)
