"""`-m gpu`: the HIP kernels on a real MI355X, through the C-ABI, against the oracle and the reference's
golden outputs.  Bit-exact: every integer table, every insertion dict, consensus bytes, change codes."""
import os
import subprocess
import sys

import numpy as np
import pytest

from kindel_amd import _native as N
from kindel_amd import shard
from tools import synth
from oracle import oracle as ko
from tests import parity as P

pytestmark = pytest.mark.gpu
QUIRKS = P.golden_quirks()
GOLD = P.golden_outputs()
MODES = [N.KD_MODE_GLOBAL, N.KD_MODE_AUTO, N.KD_MODE_COOP, N.KD_MODE_STRIP]   # AUTO = k_window (lane per read), COOP = k_window_coop
ROOT = P.ROOT


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", sorted(k for k in QUIRKS if not k.startswith("__")))
def test_quirk_case(hip_lib, name, mode):
    entry = QUIRKS[name]
    batch = P.sam_to_batch(entry["sam"])
    exc = P.quirk_expect(entry)
    if exc:
        with pytest.raises(exc):
            P.Run(hip_lib, batch, mode=mode, window=64)
        return
    for md in (1, 0, 2):
        P.assert_matches_oracle(P.Run(hip_lib, batch, mode=mode, window=64, min_depth=md), min_depth=md, what=name)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("key", P.fixture_keys())
def test_reference_fixture(hip_lib, key, mode):
    batch = P.load_fixture(key)
    run = P.Run(hip_lib, batch, mode=mode)
    P.assert_matches_golden(run, key, GOLD)   # what the unmodified reference produced
    P.assert_matches_oracle(run, what=key)    # full tables, element by element


@pytest.mark.parametrize("window,slice_reads", [(64, 16), (256, 0), (640, 64), (1024, 0), (2048, 1000), (4096, 0)])
def test_window_tunings(hip_lib, window, slice_reads):
    for key in ("bwa_mem__3.1.sub_test", "segemehl__6.1.sub_test", "minimap2__1.1.multi"):
        run = P.Run(hip_lib, P.load_fixture(key), window=window, slice_reads=slice_reads)
        # (the 3-contig minimap2 fixture is not coordinate sorted: it goes through the device bucket sort)
        assert run.info["windowed"] == 1, (key, run.info)
        assert (run.info["unsorted"] > 0) == (key == "minimap2__1.1.multi")
        P.assert_matches_golden(run, key, GOLD)


def test_unsorted_input_and_multiple_pushes(hip_lib):
    b = P.load_fixture("minimap2__hxb2-gp120-mutated")      # SO:unsorted in the reference's own fixture
    run = P.Run(hip_lib, b)
    assert run.info["windowed"] == 1 and run.info["unsorted"] > 0   # bucket-sorted on the device
    P.assert_matches_golden(run, "minimap2__hxb2-gp120-mutated", GOLD)
    b = P.load_fixture("segemehl__2.1.sub_test")
    P.assert_matches_golden(P.Run(hip_lib, b, n_pushes=5), "segemehl__2.1.sub_test", GOLD)
    rng = np.random.default_rng(1)
    perm = rng.permutation(len(b["contig"]))
    sh = dict(b)
    for k in ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig"):
        sh[k] = b[k][perm]
    P.assert_matches_golden(P.Run(hip_lib, sh), "segemehl__2.1.sub_test", GOLD)   # order independence


SYN = {
    "C2-small": lambda dev: synth.short_reads([10_000], 1500, seed=2, device=dev),
    "C3-small": lambda dev: synth.short_reads([5_000_000], 8, seed=3, device=dev),
    "C3-mid": lambda dev: synth.short_reads([1_000_000], 120, seed=33, device=dev),
    "C4-small": lambda dev: synth.short_reads([50_000] * 100, 25, seed=4, device=dev),
    "C5-small": lambda dev: synth.long_reads([400_000], 25, seed=5, device=dev),
}


def _check_device_resident(hip_lib, tb, mode, what):
    """Device-resident batch through the C-ABI: every table, insertion dict, consensus byte and change code vs the oracle."""
    host = synth.to_numpy(tb)
    eng = N.Engine(host["contig_lens"], lib=hip_lib, mode=mode)
    try:
        eng.push_device(synth.device_ptrs(tb), tb["contig"].numel(), tb["seq4_bytes"], tb["cigar_words"])
        eng.finalize()
        info, stats = eng.batch_info(), eng.stats()
        assert info["windowed"] == (0 if mode == N.KD_MODE_GLOBAL else 1)
        reads, aligned, walked = synth.counts(tb)
        assert stats["aligned"] == aligned and stats["walked"] == walked and stats["reads"] == reads
        eng.consensus_run(1)
        tot_w = 0
        for cid in ko.contig_order(host):
            oa = ko.parse_records(host, cid)
            t, L = eng.tables(cid), oa.L
            assert np.array_equal(t[0:5, :L].T, oa.weights) and np.array_equal(t[5], oa.deletions), what
            assert np.array_equal(t[6:11, :L].T, oa.clip_start_weights) and np.array_equal(t[11:16, :L].T, oa.clip_end_weights)
            assert np.array_equal(t[16], oa.clip_starts) and np.array_equal(t[17], oa.clip_ends)
            assert np.array_equal(t[18], oa.ins_totals)
            site, count, strings = eng.insertions(cid)
            assert sorted((int(p), s, int(c)) for p, c, s in zip(site, count, strings)) == sorted(oa.insertions)
            seq, ch, mm, _ = eng.consensus_fetch(cid)
            oseq, och = oa.consensus_sequence()
            assert seq.decode() == oseq and [None if c == 0 else chr(c) for c in ch] == och, what
            assert mm == oa.depth_minmax()
            tot_w += int(t[0:5].sum())
        assert tot_w == aligned   # size-independent property: every aligned base lands in exactly one counter
    finally:
        eng.close()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("cfg", sorted(SYN))
def test_synthetic_config_device_resident(hip_lib, cfg, mode):
    """BASELINE.json configs (scaled so the oracle finishes in seconds), inputs resident in HBM."""
    _check_device_resident(hip_lib, SYN[cfg]("cuda:0"), mode, cfg)


@pytest.mark.parametrize("cfg,mode", [("C2", N.KD_MODE_AUTO), ("C2", N.KD_MODE_STRIP), ("C3", N.KD_MODE_AUTO), ("C3", N.KD_MODE_COOP), ("C3", N.KD_MODE_STRIP),
                                      ("C4", N.KD_MODE_AUTO), ("C5", N.KD_MODE_AUTO)])
def test_full_size_config_device_resident(hip_lib, cfg, mode):
    """BASELINE.json configs 2-5 at FULL size (C2 10 kb x 10^4, C3 5 Mbp x 500, C4 100 x 50 kb x 1000, C5 1 Mbp x 200
    long reads): every table of every contig, the insertion dicts and the consensus against the oracle, bit for bit."""
    import torch
    tb = synth.make(cfg, device="cuda:0")
    _check_device_resident(hip_lib, tb, mode, cfg + " full size")
    del tb
    torch.cuda.empty_cache()


@pytest.mark.parametrize("cfg,layout,sort", [("C3-mid", "records", "lds"), ("C3-mid", "index", "lds"), ("C3-mid", "records", "global"),
                                             ("C3-mid", "index", "global"), ("C3", "records", "lds"), ("C3", "index", "global")])
def test_unsorted_input_at_scale(hip_lib, monkeypatch, cfg, layout, sort):
    """Unsorted input on the hardware, at sizes where every stage of the bucket sort has more than one unit of work: the
    LDS-bin counting sort with 512 sort workgroups (16 scan segments of the column scan) and the global-counter fall-back
    (the path of a genome whose bin table exceeds the LDS), both batch layouts (payload in record order = what a decoder hands
    over for an unsorted file; payload left in place), C3-mid and FULL C3: every table of every contig, the insertion dicts, the
    consensus and change codes vs the oracle (sums are order independent: the oracle walks the same shuffled batch)."""
    import torch
    if sort == "global":
        monkeypatch.setenv("KD_SORT_GLOBAL", "1")
    tb = synth.make("C3", device="cuda:0") if cfg == "C3" else SYN[cfg]("cuda:0")
    sh = synth.shuffled(tb, mode=layout, seed=7)
    del tb
    torch.cuda.synchronize()     # (the library launches on its own stream: the permuted arrays must be complete, include/kindel_hip.h)
    eng = N.Engine(sh["contig_lens"], lib=hip_lib)
    eng.push_device(synth.device_ptrs(sh), sh["contig"].numel(), sh["seq4_bytes"], sh["cigar_words"])
    info = eng.batch_info()
    eng.close()
    assert info["windowed"] == 1 and info["unsorted"] > 0
    _check_device_resident(hip_lib, sh, N.KD_MODE_AUTO, "%s shuffled (%s, %s sort)" % (cfg, layout, sort))
    del sh
    torch.cuda.empty_cache()


def test_host_push_equals_device_push(hip_lib):
    tb = synth.short_reads([200_000], 40, seed=12, device="cuda:0")
    host = synth.to_numpy(tb)
    a = P.Run(hip_lib, host)
    eng = N.Engine(host["contig_lens"], lib=hip_lib)
    eng.push_device(synth.device_ptrs(tb), tb["contig"].numel(), tb["seq4_bytes"], tb["cigar_words"])
    eng.finalize()
    assert np.array_equal(eng.tables(0), a.tables[0])
    eng.reset()   # idempotence: a second pass after reset gives the same tables
    eng.push_device(synth.device_ptrs(tb), tb["contig"].numel(), tb["seq4_bytes"], tb["cigar_words"])
    eng.finalize()
    assert np.array_equal(eng.tables(0), a.tables[0])
    eng.close()


def test_virtual_shards_and_stitch(hip_lib):
    tb = synth.short_reads([60_000, 45_000, 30_000], 30, seed=13, device="cuda:0")
    batch = synth.to_numpy(tb)
    full = P.Run(hip_lib, batch)
    world = 4
    ivs = shard.partition(batch["contig_lens"], world)
    pieces = {c: [] for c in full.order}
    for r in range(world):
        keep = shard.reads_of_rank(batch["contig_lens"], *shard.footprints(batch["contig_lens"], batch), r, world)
        sub = dict(batch)
        for k in ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig"):
            sub[k] = batch[k][keep]
        eng = N.Engine(batch["contig_lens"], lib=hip_lib)
        eng.set_shard(*ivs[r])
        eng.push(sub)
        eng.finalize()
        eng.consensus_run(1)
        seqs, changes, mm = shard.stitch(eng, ivs[r], "cuda:0")   # world 1: exercises the device-pointer path
        for c in full.order:
            pieces[c].append(seqs[c])
        eng.close()
    for c in full.order:
        assert b"".join(pieces[c]) == full.cns[c][0]


def _bam_from_fixture(tmp_path, key):
    b = P.load_fixture(key)
    p = str(tmp_path / (key + ".bam"))
    synth.write_bam(p, b, sort_order="unknown")
    return p


@pytest.mark.parametrize("key", ["bwa_mem__1.1.sub_test", "bwa_mem__5.1.sub_test", "minimap2__1.1.multi",
                                 "ext__1.issue23.debug", "ext__2.issue23.bc63", "minimap2__hxb2-gp120-mutated"])
def test_python_api_end_to_end(hip_lib, tmp_path, key):
    """decode (native) -> HIP pileup -> HIP consensus -> host splice/report, vs the reference's outputs."""
    from kindel_amd import kindel as K
    path = _bam_from_fixture(tmp_path, key)
    res = K.bam_to_consensus(path)
    res_r = K.bam_to_consensus(path, realign=True, min_overlap=7)
    for i, g in enumerate(GOLD[key]["contigs"]):
        assert res.consensuses[i].name == g["name"] + "_cns"
        assert res.consensuses[i].sequence == g["consensus"]
        assert res.refs_reports[g["name"]] == g["report"].replace("{bam_path}", path)
        assert "".join("." if c is None else c for c in res.refs_changes[g["name"]]) == g["changes"]
        assert res_r.consensuses[i].sequence == g["realign_consensus"]
        assert res_r.refs_reports[g["name"]] == g["realign_report"].replace("{bam_path}", path)


def test_known_answers_through_parse_bam(hip_lib, tmp_path):
    """/root/reference/tests/test_kindel.py:63-111 restated against kindel_amd.parse_bam."""
    from kindel_amd import kindel as K
    aln = list(K.parse_bam(_bam_from_fixture(tmp_path, "bwa_mem__1.1.sub_test")).values())[0]
    aln2 = list(K.parse_bam(_bam_from_fixture(tmp_path, "ext__3.issue23.bc75")).values())[0]
    assert aln.ref_id == "ENA|EU155341|EU155341.2" and len(aln.weights) == 9306
    assert aln.weights[0]["A"] == 22 and aln.weights[23]["A"] == 57
    assert aln2.weights[68]["G"] == 1 and aln2.weights[2368]["T"] == 13
    assert [aln2.deletions[i] for i in (399, 402, 411, 1048, 1049, 1050)] == [14, 14, 15, 14, 14, 14]
    assert aln2.clip_ends[1748] == 12
    assert aln.clip_starts[525] == 16 and aln.clip_starts[1437] == 84
    assert sum(aln2.insertions[453].values()) == 14 and sum(aln2.insertions[457].values()) == 14
    cdrps = K.cdrp_consensuses(aln.weights, aln.deletions, aln.clip_start_weights, aln.clip_end_weights,
                               aln.clip_start_depth, aln.clip_end_depth, 0.1, 10)
    assert cdrps[0][0].seq == "AACTGCCGCTAGGGGCGCGTTCGGGCTCGCCAACATCTTCAGTCCGGGCGCTAAGCAGAACATCCAGCTGATCAACA"
    assert cdrps[0][1].seq == ("AGCGTCGATGCAGATACCTACACCACCGGGGGAACTGCCGCTAGGGGCGCGTTCGGGCTCGCCAACATCTTCAGTCCGGGCGCTAAG"
                               "CAGAACA")


def test_weights_dataframe(hip_lib, tmp_path):
    from kindel_amd import kindel as K
    for key in ("bwa_mem__1.1.sub_test", "minimap2__1.1.multi"):
        path = _bam_from_fixture(tmp_path, key)
        for rel, tag in ((False, "abs"), (True, "rel")):
            df = K.weights(path, relative=rel)
            g = np.load(os.path.join(P.GOLD, "weights_%s_%s.npz" % (key, tag)), allow_pickle=True)
            assert list(df.columns) == [str(c) for c in g["columns"]]
            for c in df.columns:
                a, b = df[c].to_numpy(), g[c]
                if a.dtype.kind == "f":
                    assert np.allclose(a, b, rtol=0, atol=1e-12, equal_nan=True), (key, tag, c)
                else:
                    assert (a.astype(str) == b.astype(str)).all() if a.dtype == object else np.array_equal(a, b), (key, tag, c)


def test_variants_extension(hip_lib, tmp_path):
    from kindel_amd import kindel as K
    for key in ("minimap2__1.1.multi", "bwa_mem__1.1.sub_test"):
        path = _bam_from_fixture(tmp_path, key)
        for a, r, only in ((1, 0.01, True), (5, 0.1, False)):
            df = K.variants(path, abs_threshold=a, rel_threshold=r, only_variants=only)
            assert P.variants_rows(df) == P.expected_variants(P.load_fixture(key), a, r, only), (key, a, r, only)
    r = subprocess.run([sys.executable, "-m", "kindel_amd", "variants", "-a", "5", path], cwd=ROOT, capture_output=True,
                       text=True)
    assert r.returncode == 0 and r.stdout.startswith("chrom\tpos\tref\talt\ttype\tcount\tdepth\tfrequency")


def test_cli_consensus_stdout(hip_lib, tmp_path):
    key = "ext__3.issue23.bc75"
    path = _bam_from_fixture(tmp_path, key)
    r = subprocess.run([sys.executable, "-m", "kindel_amd", "consensus", path], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    g = GOLD[key]["contigs"][0]
    assert r.stdout == ">%s_cns\n%s\n" % (g["name"], g["consensus"])
    assert "REPORT" in r.stderr
    r = subprocess.run([sys.executable, "-m", "kindel_amd", "consensus", "-r", "-t", "-u", path], cwd=ROOT,
                       capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.split("\n")[1] == g["realign_consensus"].strip("N").upper()


@pytest.mark.parametrize("key,tag", __import__("tests.refcheck", fromlist=["x"]).FASTA_CASES)
def test_reference_fasta(hip_lib, tmp_path, key, tag):
    """The 21 FASTA files the reference's own CLI tests compare with (tests/golden/reference_fasta.json)."""
    from kindel_amd import kindel as K
    from tests import refcheck as RC
    RC.check_reference_fasta(K, tmp_path, key, tag)


def test_features_and_derived_arrays(hip_lib, tmp_path):
    from kindel_amd import kindel as K
    from tests import refcheck as RC
    for key in RC.FEATURE_KEYS:
        RC.check_features(K, tmp_path, key)
    for key in ("bwa_mem__2.1.sub_test", "minimap2__1.1.multi", "ext__1.issue23.debug", "segemehl__4.1.sub_test"):
        RC.check_derived_arrays(K, tmp_path, key, GOLD)


def test_fetch_all_equals_per_contig_fetch(hip_lib):
    P.check_fetch_all(hip_lib, P.load_fixture("minimap2__1.1.multi"))
    P.check_fetch_all(hip_lib, synth.to_numpy(synth.make("C4", scale=0.02)))


def test_multi_tile_items_carry_rows_between_tiles(hip_lib):
    batch = synth.to_numpy(synth.short_reads([2000, 900], 3000, seed=9, planted=False))   # ~58000 reads, 3 windows of 1024
    for window, slice_reads in ((1024, 32768), (640, 4096), (256, 3000)):
        P.assert_matches_oracle(P.Run(hip_lib, batch, window=window, slice_reads=slice_reads), what="w%d s%d" % (window, slice_reads))


@pytest.mark.parametrize("mode", [N.KD_MODE_AUTO, N.KD_MODE_COOP, N.KD_MODE_STRIP])
def test_mostly_clipped_reads_and_deep_sites(hip_lib, mode):
    """k_prep writes a compact record for every clipped / inserted read (ballot-compacted per wavefront, one region of the
    record array per wavefront), k_cold_lane walks the regions (more than one round of 256 where most reads are clipped) and
    lets neighbouring lanes that aim at one site add once (as does k_ins_insert).  Deep, clip-heavy input exercises all of it.
    (One test per mode and a failure message that names the cells: this test failed ONCE in a plain suite run of round 6's last
    session -- the only failure in a dozen runs, plain and fenced, of that library, not reproduced in 1 000 repeats:
    profiles/r06_one_unreproduced_test_failure.txt.)"""
    batch = synth.to_numpy(synth.short_reads([2500, 1200], 2500, seed=13, clip_p=0.6, indel_p=0.3))
    assert len(batch["contig"]) > 3 * 8192
    P.assert_matches_oracle(P.Run(hip_lib, batch, mode=mode), what="mode %d" % mode)


def test_insertion_hash_collision_is_detected_and_reseeded(hip_lib, monkeypatch):
    """KD_TEST_INS_COLLIDE leaves two possible insertion keys in the first attempt: the verification must notice, kd_finalize
    must clean up, re-seed and redo the reduction (incl. the speculatively picked winners)."""
    batch = synth.to_numpy(synth.short_reads([20000], 60, seed=21, indel_p=0.5))
    monkeypatch.setenv("KD_TEST_INS_COLLIDE", "1")
    P.assert_matches_oracle(P.Run(hip_lib, batch))


def test_profile_modes(hip_lib):
    """kd_profile_enable: 1 = hipEvents around every launch, 2 = only around k_window (what bench.py times with)."""
    batch = synth.to_numpy(synth.short_reads([20000], 30, seed=4))
    eng = N.Engine(batch["contig_lens"], lib=hip_lib)
    try:
        for mode, check in ((1, lambda rows: {"k_prep", "k_window", "k_cns_emit"} <= set(rows)),
                            (2, lambda rows: set(rows) == {"k_window"}),
                            (0, lambda rows: not rows)):
            eng.reset()
            eng.profile_enable(mode)
            eng.profile_reset()
            eng.push(batch)
            eng.finalize()
            eng.consensus_run(1)
            rows = eng.profile()
            assert check(rows), (mode, sorted(rows))
            assert all(n >= 1 and ms > 0 for n, ms in rows.values())
    finally:
        eng.close()


LONG_CASES = P.long_read_cases()
LONG_GOLD = P.golden_long_quirks()


@pytest.mark.parametrize("mode", [N.KD_MODE_AUTO, N.KD_MODE_GLOBAL])
@pytest.mark.parametrize("name", sorted(LONG_CASES))
def test_long_read_case(hip_lib, name, mode):
    """kd_long.h on the GPU: rows, "+ins" symbols, several insertions on one site, tile boundaries, clips, bad bases --
    against the oracle and against what the unmodified reference returned (tests/golden/long_quirks.json)"""
    sam, exc = LONG_CASES[name]
    batch = P.sam_to_batch(sam)
    if exc:
        with pytest.raises(exc):
            P.Run(hip_lib, batch, mode=mode, window=64)
        return
    for window, sl in ((64, 0), (448, 16)):
        run = P.Run(hip_lib, batch, mode=mode, window=window, slice_reads=sl)
        assert run.info["long_cigar"] >= 1
        P.assert_matches_oracle(run, what=name)
        P.assert_matches_long_golden(run, LONG_GOLD[name], what=name)


def test_virtual_shards_with_long_reads(hip_lib):
    """Interval shards + the long-read segment pass (second k_window launch) on the GPU."""
    batch = synth.to_numpy(synth.long_reads([120_000], 12, seed=12, median_len=6000, min_len=1500, max_len=15000))
    full = P.Run(hip_lib, batch)
    assert full.info["long_cigar"] > 0
    P.assert_matches_oracle(full)
    world = 4
    ivs = shard.partition(batch["contig_lens"], world)
    base, S = shard.g_layout(batch["contig_lens"])
    pieces = []
    for r in range(world):
        keep = shard.reads_of_rank(batch["contig_lens"], *shard.footprints(batch["contig_lens"], batch), r, world)
        sub = dict(batch)
        for k in ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig"):
            sub[k] = batch[k][keep]
        run = P.Run(hip_lib, sub, shard=ivs[r])
        lo = max(int(base[0]), ivs[r][0]) - int(base[0])
        hi = min(int(base[0]) + int(batch["contig_lens"][0]), ivs[r][1]) - int(base[0])
        assert np.array_equal(run.tables[0][:, lo:hi], full.tables[0][:, lo:hi]), r
        pieces.append(run.cns[0][0])
    assert b"".join(pieces) == full.cns[0][0]


@pytest.mark.parametrize("cfg", ["C4", "C3"])
def test_full_size_config_as_eight_shards(hip_lib, cfg):
    """north_star's multi-GPU decomposition at FULL size on ONE GPU: config 4 (100 contigs x 50 kb x 1000x) -> 8 work-balanced
    shards of whole contigs, config 3 (5 Mbp x 500x) -> 8 position intervals (tests/parity.py: check_as_shards)."""
    import torch
    tb = synth.make(cfg, device="cuda:0")
    ivs = P.check_as_shards(hip_lib, synth.to_numpy(tb), 8, dev="cuda:0", tb=tb)
    if cfg == "C4":   # 100 equal contigs: the cuts snap onto contig boundaries, 12 - 13 whole contigs per shard
        base, _ = shard.g_layout(tb["contig_lens"])
        assert all(iv[0] in set(int(b) for b in base) for iv in ivs)
    del tb
    torch.cuda.empty_cache()


def test_cli_two_ranks_on_one_gpu(hip_lib, tmp_path):
    """`python -m kindel_amd consensus --gpus 2 x.bam` end to end on the real library: two processes (here sharing the one GPU,
    gloo instead of RCCL), each decoding its share of the file, rank 0 prints -- stdout equals the single-process run and the
    reference's golden FASTA."""
    key = "bwa_mem__3.1.sub_test"
    path = str(tmp_path / "x.bam")
    synth.write_bam(path, P.load_fixture(key), sort_order="unknown", block_bytes=3000)
    env = dict(os.environ, KINDEL_DIST_BACKEND="gloo", PYTHONPATH=ROOT)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    one = subprocess.run([sys.executable, "-m", "kindel_amd", "consensus", path], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    two = subprocess.run([sys.executable, "-m", "kindel_amd", "consensus", "--gpus", "2", path], capture_output=True, text=True, timeout=900,
                         env=env, cwd=ROOT)
    assert one.returncode == 0 and two.returncode == 0, two.stderr[-2000:]
    two_out = two.stdout[two.stdout.index(">"):] if ">" in two.stdout else two.stdout   # (gloo's C++ side greets on stdout; RCCL does not)
    two = subprocess.CompletedProcess(two.args, two.returncode, two_out, two.stderr)
    assert two.stdout == one.stdout
    g = GOLD[key]["contigs"][0]
    assert two.stdout == ">%s_cns\n%s\n" % (g["name"], g["consensus"])
    assert one.stderr.strip() and one.stderr.strip() in two.stderr     # the same report (the CLI's own defaults, e.g. min_overlap 7)


def test_cli_two_ranks_realign_on_one_gpu(hip_lib, tmp_path):
    """`kindel consensus --realign --gpus 2` on the real library (round 5): the shards' tables summed over the two ranks, the same
    clip-dominant regions found by both, each patching its part -- stdout is the FASTA the unmodified reference wrote with
    realign=True (a fixture whose consensus realign changes), the report on stderr lists the region."""
    key = "bwa_mem__1.1.sub_test"
    g = GOLD[key]["contigs"][0]
    assert g["realign_consensus"] != g["consensus"]
    path = str(tmp_path / "x.bam")
    synth.write_bam(path, P.load_fixture(key), sort_order="unknown", block_bytes=3000)
    env = dict(os.environ, KINDEL_DIST_BACKEND="gloo", PYTHONPATH=ROOT)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    two = subprocess.run([sys.executable, "-m", "kindel_amd", "consensus", "--realign", "--gpus", "2", path], capture_output=True, text=True,
                         timeout=900, env=env, cwd=ROOT)
    assert two.returncode == 0, two.stderr[-2000:]
    out = two.stdout[two.stdout.index(">"):] if ">" in two.stdout else two.stdout
    assert out == ">%s_cns\n%s\n" % (g["name"], g["realign_consensus"])
    region = [l for l in g["realign_report"].split("\n") if l.startswith("- clip-dominant regions")][0]
    assert region.split(": ", 1)[1] and region in two.stderr


def test_bench_two_ranks_on_one_gpu_same_fasta():
    """bench.py's N > 1 product path on hardware as far as one GPU allows: `bench.py --gpus 2` under torch.distributed.run (the
    driver's launch line, gloo instead of RCCL because both ranks share the GPU) -- one input cut into two work-balanced position
    intervals, shard-local tables, the engine-written exchange row, ONE all-gather -- leaves the same FASTA (sha256 over all
    contigs) as the single-GPU run of the same workload; weak scaling (two config-sized intervals) measured in the same call."""
    import json
    import socket
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    common = ["--steps", "2", "--warmup", "1", "--scale", "0.05", "--no-cpu-baseline", "--e2e-scale", "0"]
    one = subprocess.run([sys.executable, "bench.py"] + common, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), "bench.py", "--gpus", "2", "--backend", "gloo", "--scaling", "strong"] + common,
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert two.returncode == 0, two.stderr[-2000:]
    a = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    b = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2 and b["scaling"] == "strong"
    assert b["fasta_sha256"] == a["fasta_sha256"] and b["consensus_len"] == a["consensus_len"]
    assert b["config"]["aligned_events"] == a["config"]["aligned_events"]        # the ONE input, every read counted once
    assert b["other_scaling"]["scaling"] == "weak" and b["other_scaling"]["value"] > 0


@pytest.mark.parametrize("key,chunk", [("bwa_mem__2.1.sub_test", 5000), ("minimap2__1.1.multi", 900), ("ext__1.issue23.debug", 30000)])
def test_streamed_ingest_many_small_batches(hip_lib, tmp_path, key, chunk):
    """kd_push_stream on the GPU with chunks small enough that batch boundaries cut windows (decode thread + pushing thread,
    dozens of batches into the same tables) == one whole-file batch == the reference's golden consensus."""
    from kindel_amd import kindel as K
    path = str(tmp_path / "p.bam")
    synth.write_bam(path, P.load_fixture(key), sort_order="unknown")
    a = K.pileup_file(path, stream=False)
    b = K.pileup_file(path, stream=True, chunk_bytes=chunk)
    assert b.ingest["batches"] > 3
    assert [a.names[c] for c in a.order] == [b.names[c] for c in b.order]
    for ca, cb in zip(a.order, b.order):
        assert np.array_equal(a.tables(ca), b.tables(cb))
    ra = K.bam_to_consensus(path)
    assert [c.sequence for c in ra.consensuses] == [g["consensus"] for g in GOLD[key]["contigs"]]


@pytest.mark.parametrize("cold_tail", ["1", "0"])
def test_cold_records_riding_in_the_window_launch_gpu(hip_lib, monkeypatch, cold_tail):
    """KD_COLD_TAIL (default 1 since round 6): k_cold_lane's work as workgroups behind k_window's persistent ones, or (0) as a launch
    of its own -- every table, insertion dict and consensus of a clip- and indel-rich batch and of C2 against the oracle either
    way; the reference's exceptions still surface."""
    monkeypatch.setenv("KD_COLD_TAIL", cold_tail)
    batch = synth.to_numpy(synth.short_reads([90000, 25000], 300, seed=15, clip_p=0.3, indel_p=0.3))
    run = P.Run(hip_lib, batch)
    assert run.info["windowed"] == 1
    P.assert_matches_oracle(run)
    P.assert_matches_oracle(P.Run(hip_lib, synth.to_numpy(synth.make("C2", scale=0.2))))
    for key in QUIRKS:
        exc = P.quirk_expect(QUIRKS[key])
        if exc and not key.startswith("__"):
            with pytest.raises(exc):
                P.Run(hip_lib, P.sam_to_batch(QUIRKS[key]["sam"]), window=64)


def _step_sequence(lib):
    """kd_step over one resident batch, step after step, every step's consensus and tables against the oracle: the same batch
    again, bases changed in place under the same pointers, a CIGAR changed in place so that the event counts move."""
    import torch
    tb = synth.short_reads([120_000, 30_000], 40, seed=31, device="cuda:0")
    eng = N.Engine(tb["contig_lens"], lib=lib)
    out = torch.empty(400_000, dtype=torch.uint8, pin_memory=True).numpy()

    def check():
        torch.cuda.synchronize()      # (the library runs on its own stream: torch's writes to the batch must have landed)
        off = eng.step_device(synth.device_ptrs(tb), tb["contig"].numel(), tb["seq4_bytes"], tb["cigar_words"], out)
        host = synth.to_numpy(tb)
        for cid in ko.contig_order(host):
            oa = ko.parse_records(host, cid)
            assert out[int(off[cid]): int(off[cid + 1])].tobytes().decode() == oa.consensus_sequence()[0], cid
            t = eng.tables(cid)
            assert np.array_equal(t[0:5, :oa.L].T, oa.weights) and np.array_equal(t[18], oa.ins_totals)
        return off

    try:
        for _ in range(4):
            check()
        # new bases under the same pointers
        nib = torch.tensor([1, 2, 4, 8], dtype=torch.uint8, device="cuda:0")
        r = torch.randint(0, 4, (tb["seq4_bytes"],), device="cuda:0")
        tb["seq4"][: tb["seq4_bytes"]] = (nib[r] << 4) | nib[(r + 1) % 4]
        check()
        check()
        # one insertion less (the I of a 5-op read S M I M S becomes an M): the event count changes
        ncig = tb["n_cig"].cpu().numpy()
        i = int(np.flatnonzero(ncig == 5)[0]) if (ncig == 5).any() else None
        if i is not None:
            co = int(tb["cig_off"][i])
            w = int(tb["cigar"][co + 2])
            ln, op = w >> 4, w & 15
            if op != 2:         # (a deletion -> insertion of the same length would leave a read whose query no longer adds up: skipped)
                tb["cigar"][co + 2] = (ln << 4) | 0
                check()
                check()
    finally:
        eng.close()


def test_step_raises_the_reference_exception_and_the_next_step_is_clean(hip_lib):
    """kd_step launches the batch's error classification (k_errors) only when its status words come back flagged (round 6): a base
    outside A,C,G,T,N planted in place raises KeyError through kd_step, a read moved behind its contig's end IndexError, and with the
    inputs repaired the SAME context steps clean, tables and consensus against the oracle."""
    import torch
    tb = synth.short_reads([60_000, 20_000], 30, seed=37, device="cuda:0")
    eng = N.Engine(tb["contig_lens"], lib=hip_lib)
    out = torch.empty(200_000, dtype=torch.uint8, pin_memory=True).numpy()

    def step():
        torch.cuda.synchronize()
        return eng.step_device(synth.device_ptrs(tb), tb["contig"].numel(), tb["seq4_bytes"], tb["cigar_words"], out)

    def clean():
        off = step()
        host = synth.to_numpy(tb)
        for cid in ko.contig_order(host):
            oa = ko.parse_records(host, cid)
            assert out[int(off[cid]): int(off[cid + 1])].tobytes().decode() == oa.consensus_sequence()[0], cid
            t = eng.tables(cid)
            assert np.array_equal(t[0:5, :oa.L].T, oa.weights) and np.array_equal(t[18], oa.ins_totals)

    try:
        clean()
        ncig = tb["n_cig"].cpu().numpy()
        i = int(np.flatnonzero(ncig == 1)[len(ncig) // 3])          # a plain read somewhere in the middle
        so = int(tb["seq_off"][i])
        keep = int(tb["seq4"][so + 3])
        tb["seq4"][so + 3] = 0x35                                   # 'M' / 'R': neither in the reference's weight dict
        with pytest.raises(KeyError):
            step()
        with pytest.raises(KeyError):                               # ... and again: nothing of the flagged step sticks
            step()
        tb["seq4"][so + 3] = keep
        clean()
        pos_keep = int(tb["pos0"][i])
        tb["pos0"][i] = int(tb["contig_lens"][int(tb["contig"][i])]) - 10      # runs off the reference: list index out of range
        with pytest.raises(IndexError):
            step()
        tb["pos0"][i] = pos_keep
        clean()
    finally:
        eng.close()


def test_step_repeats_and_inputs_changed_in_place(hip_lib):
    """kd_step: see _step_sequence.  (Rounds 3 - 5 also had an opt-in hipGraph replay of a repeated step and its test here; removed in
    round 6 with the replay: DESIGN section 3.)"""
    _step_sequence(hip_lib)


def test_bench_step_through_rccl_at_world_size_one():
    """The N-GPU step on the one GPU a box has: `bench.py --gpus 1 --rccl-at-1` initialises torch.distributed with backend nccl (= RCCL)
    at world size 1, registers the exchange row with the engine (kd_set_exchange) and ends every step with
    dist.all_gather_into_tensor on that device row (shard.Exchange.collect) -- the collective of kindel_amd/shard.py executed by RCCL on
    an MI355X, its result assembled into the same FASTA (sha256) as the plain single-GPU run."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    common = ["--steps", "3", "--warmup", "1", "--scale", "0.05", "--no-cpu-baseline", "--e2e-scale", "0"]
    one = subprocess.run([sys.executable, "bench.py"] + common, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    env["NCCL_DEBUG"] = "INFO"
    two = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--rccl-at-1"] + common, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert two.returncode == 0, two.stderr[-3000:]
    def line(out):      # (RCCL's INFO lines share the stream: the bench line may not start its line)
        hits = [l[l.index('{"metric"'):] for l in out.splitlines() if '{"metric"' in l]
        assert hits, out[-3000:]
        return json.loads(hits[-1])
    a, b = line(one.stdout), line(two.stdout + "\n" + two.stderr)
    assert b["n_gpus"] == 1 and "rccl_at_1" in b and "rccl_at_1" not in a
    assert b["fasta_sha256"] == a["fasta_sha256"] and b["consensus_len"] == a["consensus_len"]
    log = two.stdout + two.stderr
    assert "NCCL INFO" in log and ("RCCL" in log or "rccl" in log or "nranks 1" in log), log[-1500:]      # the library that ran is RCCL


def test_guard_subprocess_fence_faults_on_an_overread_and_clean_runs_pass():
    """KD_GUARD=1 (kindel_hip.hip, round 6: every device buffer mapped by itself so that its last byte is the last mapped byte of its
    address range, a caller's device batch copied into fenced buffers of exactly the promised sizes): (1) the fence WORKS on this box --
    a batch whose seq4_bytes is declared 256 bytes short makes k_window read past its copy and the process dies of a GPU memory fault
    whose report names the buffer and the kernel (scripts/exp/guard_selftest.py); (2) a clean run -- smoke()'s two paths, a step
    sequence with inputs changed in place -- passes under the fence.  In processes of their own: a GPU fault takes its process down."""
    env = dict(os.environ, PYTHONPATH=ROOT, KD_GUARD="1")
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "exp", "guard_selftest.py")], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert bad.returncode != 0, bad.stdout[-500:]
    assert "Memory access fault" in bad.stderr and "[kd guard] SIGABRT; last kernel launched: k_window" in bad.stderr and "b_gin[k]" in bad.stderr, bad.stderr[-1500:]
    ok = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke(); from tests import test_gpu_parity as T; from kindel_amd import _native as N; "
                         "T._step_sequence(N.default_library()); print('GUARDED-OK')"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert ok.returncode == 0 and "GUARDED-OK" in ok.stdout and "[kd guard] on" in ok.stderr, (ok.returncode, ok.stdout[-800:], ok.stderr[-1500:])
    env.pop("KD_GUARD")
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "exp", "guard_selftest.py")], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert plain.returncode == 0, plain.stderr[-800:]      # (without the fence the same call reads the caller's larger tensor: nothing happens)


def test_reads_in_the_middle_of_a_long_contig_over_dirty_device_memory(hip_lib):
    """Round 6: k_prep writes k_window's boundary table only between the batch's first and last read, k_window's look-ups are clamped to
    that range (kd_prep.h, kd_engine.h).  hipMalloc does not hand out zero pages in a long-lived process: device memory is dirtied first
    (allocated, filled with 0xA5, freed), then reads that cover only the middle of a long contig -- a third of nothing in front, a gap, a
    third behind -- run alone, in three pushes, as four shards and in random order.  The CPU twin with a poisoned emulator:
    tests/test_emu_kernels.py::test_poisoned_device_memory_changes_nothing."""
    import torch
    dirt = [torch.full((64 << 20,), 0xA5, dtype=torch.uint8, device="cuda") for _ in range(8)]
    torch.cuda.synchronize()
    del dirt
    torch.cuda.empty_cache()
    tb = synth.to_numpy(synth.short_reads([600000], 30, seed=78))
    keep = np.flatnonzero(((tb["pos0"] >= 210000) & (tb["pos0"] < 290000)) | ((tb["pos0"] >= 350000) & (tb["pos0"] < 390000)))
    mid = dict(tb)
    for k in ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig"):
        mid[k] = tb[k][keep]
    for window, n_pushes in ((0, 1), (448, 3)):
        P.assert_matches_oracle(P.Run(hip_lib, mid, window=window, n_pushes=n_pushes), what="middle of a contig, window %d" % window)
    P.check_as_shards(hip_lib, mid, 4, dev="cuda")
    shuf = dict(mid)
    perm = np.random.default_rng(6).permutation(len(keep))
    for k in ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig"):
        shuf[k] = mid[k][perm]
    P.assert_matches_oracle(P.Run(hip_lib, shuf), what="the same reads in random order")


def test_more_contigs_than_sixteen_bits_hold(hip_lib):
    """200 001 short contigs on the GPU (tests/parity.py: check_many_contigs): a sample against the oracle, four shards against one context."""
    assert P.check_many_contigs(hip_lib, 200001, world=4, sample=200) >= 150


def test_sites_beyond_two_to_the_thirty_one(hip_lib):
    """4.2 G sites of G-space in four contigs on the GPU: shards straddling G-site 2^31, at G ~ 3.3 G and at the last sites of G-space
    equal the same reads on a small contig of their own (tests/parity.py: check_huge_g_space)."""
    assert P.check_huge_g_space(hip_lib, depth=40, region=300000) == 3

