"""Device-side ingest (kindel_amd/csrc/kd_gpu_inflate.h + kd_ingest.h, kd_push_bam_gpu): BGZF inflate, BAM record walk and the batch
arrays ON the device, against the host decoder feeding the same engine -- every table of every contig, the insertion dicts, the record
counts; through the Python API against the reference's goldens; the files the path must hand back to the host decoder; corrupt files.
Here on the CPU emulator (kernel logic); tests/test_gpu_parity.py runs the same comparison on the GPU."""
import os
import struct

import numpy as np
import pytest

from kindel_amd import _native as N
from tools import synth

REF = "/root/reference/tests"


@pytest.fixture(autouse=True, params=["1", "2"], ids=["inflate_one_pass", "inflate_two_pass"])
def inflate_kernel(request, monkeypatch):
    """Every test of this module under both GPU inflaters (the engine reads KD_INFLATE when a context is created): the one-pass kernel
    of rounds 3 - 5 (kd_gpu_inflate.h: a wavefront per BGZF block) and round 6's two-pass pair (kd_gpu_inflate2.h: a lane per block
    records the matches, a wavefront per block resolves them)."""
    monkeypatch.setenv("KD_INFLATE", request.param)
    return request.param


def both_ways(lib, path):
    with N.BgzfPlan(path, lib=lib) as plan:
        e1 = N.Engine(plan.contig_lens, lib=lib)
        info = e1.push_bam_gpu(plan)
        e1.finalize()
        d = N.decode_file(path, lib=lib)
        assert [str(x) for x in d["contig_names"]] == plan.contig_names and np.array_equal(d["contig_lens"], plan.contig_lens)
        e2 = N.Engine(plan.contig_lens, lib=lib)
        e2.push(d)
        e2.finalize()
        assert info["kept"] == d["contig"].size
        for c in range(len(plan.contig_lens)):
            assert np.array_equal(e1.tables(c), e2.tables(c)), (path, c)
            i1, i2 = e1.insertions(c), e2.insertions(c)
            assert sorted(zip(i1[0].tolist(), i1[1].tolist(), i1[2])) == sorted(zip(i2[0].tolist(), i2[1].tolist(), i2[2])), (path, c)
        assert np.array_equal(e1.contig_first(), e2.contig_first())
        e1.close(); e2.close()
        return info


def test_synthetic_bams_small_blocks_and_unmapped_reads(emu_lib, tmp_path):
    batch = synth.to_numpy(synth.short_reads([9000, 2500, 700], 40, seed=5, clip_p=0.2, indel_p=0.2))
    flag = batch["flag"].copy()
    flag[::17] |= 4                                     # unmapped by flag: kept as records, skipped by the record loop
    batch = dict(batch, flag=flag)
    for block_bytes in (0xff00, 4096, 700, 150):        # records across block boundaries; blocks in which no record starts
        p = str(tmp_path / ("b%d.bam" % block_bytes))
        synth.write_bam(p, batch, block_bytes=block_bytes)
        info = both_ways(emu_lib, p)
        assert info["records"] == len(batch["contig"])
    p = str(tmp_path / "native.bam")
    os.environ["KD_WRITE_BAM_QUAL"] = "phred"
    try:
        N.write_bam(p, batch, lib=emu_lib)
    finally:
        os.environ.pop("KD_WRITE_BAM_QUAL", None)
    both_ways(emu_lib, p)


def test_the_file_uploaded_in_many_small_pieces(emu_lib, tmp_path, monkeypatch):
    # the file goes to the device in pieces and every piece's complete BGZF blocks are inflated while the next piece is copied:
    # here pieces of 3000 bytes against blocks of ~1 - 2 KB (a block is launched only when its trailer has arrived)
    monkeypatch.setenv("KD_UPLOAD_CHUNK", "3000")
    batch = synth.to_numpy(synth.short_reads([9000, 2500], 30, seed=6, clip_p=0.2, indel_p=0.2))
    p = str(tmp_path / "pieces.bam")
    synth.write_bam(p, batch, block_bytes=4096)
    both_ways(emu_lib, p)


def test_long_reads_and_odd_lengths(emu_lib, tmp_path):
    batch = synth.to_numpy(synth.long_reads([60000], 6, seed=3, median_len=4000, min_len=501, max_len=9001))
    assert (batch["seq_len"] & 1).any() and (batch["n_cig"] > 16).any()
    p = str(tmp_path / "long.bam")
    synth.write_bam(p, batch, block_bytes=3000)         # most records longer than a block
    both_ways(emu_lib, p)


def test_refid_minus_one_records_are_dropped_like_the_host_decoder_drops_them(emu_lib, tmp_path):
    batch = synth.to_numpy(synth.short_reads([5000], 20, seed=9))
    contig = batch["contig"].astype(np.int64).copy()
    contig[5::7] = -1
    b2 = dict(batch, contig=contig.astype(np.int32))
    p = str(tmp_path / "unplaced.bam")
    synth.write_bam(p, b2, block_bytes=2000)
    info = both_ways(emu_lib, p)
    assert info["records"] == len(contig) and info["kept"] == int((contig >= 0).sum())


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference's BAM files live on the build container only")
def test_reference_bams(emu_lib):
    import glob
    paths = sorted(glob.glob(os.path.join(REF, "data_*", "*.bam")))
    assert len(paths) >= 10
    for p in paths:
        if os.path.getsize(p) < 3_000_000:              # (bact.tiny: 6.1 Mbp of tables per engine on the emulator -- covered on the GPU)
            both_ways(emu_lib, p)


def test_api_through_the_gpu_ingest_equals_the_host_path(api_on_emu, tmp_path, monkeypatch):
    from kindel_amd import kindel
    batch = synth.to_numpy(synth.short_reads([7000, 6200], 30, seed=5))
    p = str(tmp_path / "x.bam")
    synth.write_bam(p, batch, block_bytes=5000)
    host = kindel.bam_to_consensus(p)
    monkeypatch.setenv("KINDEL_INGEST", "gpu")
    pl = kindel.pileup_file(p)
    assert pl.ingest["path"] == "gpu"
    gpu = kindel.bam_to_consensus(p)
    assert [str(s.sequence) for s in gpu.consensuses] == [str(s.sequence) for s in host.consensuses]
    assert gpu.refs_reports == host.refs_reports and gpu.refs_changes == host.refs_changes
    a = kindel.parse_bam(p)
    monkeypatch.delenv("KINDEL_INGEST")
    b = kindel.parse_bam(p)
    assert list(a) == list(b)
    for k in a:
        for x, y in zip(a[k], b[k]):
            if isinstance(x, np.ndarray):
                assert np.array_equal(x, y)


def test_files_for_the_host_decoder(emu_lib, api_on_emu, tmp_path, monkeypatch):
    from kindel_amd import kindel
    # SAM text: refused when the plan is made
    sam = str(tmp_path / "x.sam")
    open(sam, "w").write("@HD\tVN:1.6\n@SQ\tSN:c\tLN:100\nr1\t0\tc\t5\t60\t10M\t*\t0\t0\tACGTACGTAC\t*\n")
    with pytest.raises(N.UnsupportedByGpuIngest):
        N.BgzfPlan(sam, lib=emu_lib)
    # a CIGAR of more than 65535 operations travels in a CG:B,I tag: the device walk follows the placeholder to the tag's array
    n_ops = 66000
    cig = np.empty(n_ops, np.uint32)
    cig[0::2] = (1 << 4) | 0
    cig[1::2] = (1 << 4) | 2
    sl = n_ops // 2
    batch = dict(contig=np.zeros(1, np.uint32), pos0=np.zeros(1, np.int32), flag=np.zeros(1, np.uint32), seq_off=np.zeros(1, np.uint64),
                 seq_len=np.asarray([sl], np.uint32), cig_off=np.zeros(1, np.uint64), n_cig=np.asarray([n_ops], np.uint32),
                 seq4=np.full((sl + 1) // 2 + 8, 0x11, np.uint8), cigar=np.concatenate([cig, np.zeros(2, np.uint32)]),
                 contig_lens=np.asarray([200000], np.uint32), contig_names=np.asarray(["c"]))
    p = str(tmp_path / "cg.bam")
    synth.write_bam(p, batch)
    both_ways(emu_lib, p)                             # (the same batch, tables and insertions as the host decoder's)
    monkeypatch.setenv("KINDEL_INGEST", "gpu")
    pl = kindel.pileup_file(p)
    assert getattr(pl, "ingest", {}).get("path") == "gpu"
    assert int(np.asarray(pl.tables(0))[:5].sum()) == sl        # every base of the one read tallied (A / deleted site alternating)


def test_a_block_with_a_wrong_crc_is_refused(emu_lib, tmp_path):
    # htslib: "CRC32 checksum mismatch"; the host reader checks every block's trailer, and so does the device-side path (k_bgzf_crc)
    batch = synth.to_numpy(synth.short_reads([4000], 30, seed=2))
    p = str(tmp_path / "x.bam")
    synth.write_bam(p, batch, block_bytes=3000)
    raw = bytearray(open(p, "rb").read())
    o = 0
    for _ in range(3):                                   # the third block's trailer
        o += struct.unpack_from("<H", raw, o + 16)[0] + 1
    raw[o - 8] ^= 0x01
    q = str(tmp_path / "bad_crc.bam")
    open(q, "wb").write(bytes(raw))
    with N.BgzfPlan(q, lib=emu_lib) as plan:
        eng = N.Engine(plan.contig_lens, lib=emu_lib)
        with pytest.raises(OSError, match="CRC-32"):
            eng.push_bam_gpu(plan)
        eng.close()
    both_ways(emu_lib, p)                                # (the untouched file is fine)


def test_corrupt_files_are_refused_not_crashed_on(emu_lib, tmp_path):
    batch = synth.to_numpy(synth.short_reads([4000], 30, seed=2))
    p = str(tmp_path / "x.bam")
    synth.write_bam(p, batch, block_bytes=3000)
    raw = bytearray(open(p, "rb").read())
    rng = np.random.default_rng(4)
    outcomes = set()
    for t in range(40):
        bad = bytearray(raw)
        for _ in range(int(rng.integers(1, 4))):
            bad[int(rng.integers(200, len(bad) - 30))] ^= 1 << int(rng.integers(0, 8))
        q = str(tmp_path / ("bad%d.bam" % t))
        open(q, "wb").write(bytes(bad))
        try:
            with N.BgzfPlan(q, lib=emu_lib) as plan:
                eng = N.Engine(plan.contig_lens, lib=emu_lib)
                try:
                    eng.push_bam_gpu(plan)
                    eng.finalize()
                    outcomes.add("ok")            # (a flipped bit outside the deflate payload and the checked fields, e.g. in the gzip header's MTIME)
                except (OSError, N.UnsupportedByGpuIngest, KeyError, IndexError, RuntimeError, N.KindelNativeError) as e:
                    outcomes.add(type(e).__name__)
                finally:
                    eng.close()
        except (OSError, N.UnsupportedByGpuIngest, N.KindelNativeError) as e:
            outcomes.add(type(e).__name__)
    assert outcomes - {"ok"}, outcomes


@pytest.mark.gpu
def test_gpu_ingest_on_the_gpu(hip_lib, tmp_path):
    """The same comparison on the MI355X: a few hundred BGZF blocks through k_gpu_inflate + k_bam_* vs the host decoder feeding the
    same engine (short reads with Phred-like qualities, native writer; long reads, python writer with small blocks)."""
    import torch
    tb = synth.make("C3", scale=0.02, device="cuda:0")
    batch = synth.to_numpy(tb)
    p = str(tmp_path / "c3.bam")
    os.environ["KD_WRITE_BAM_QUAL"] = "phred"
    try:
        N.write_bam(p, batch, lib=hip_lib)
    finally:
        os.environ.pop("KD_WRITE_BAM_QUAL", None)
    info = both_ways(hip_lib, p)
    assert info["records"] == len(batch["contig"]) and info["blocks"] > 100
    lb = synth.to_numpy(synth.long_reads([60000], 6, seed=3, median_len=4000, min_len=501, max_len=9001))
    q = str(tmp_path / "long.bam")
    synth.write_bam(q, lb, block_bytes=3000)
    both_ways(hip_lib, q)
    # a read of 66 000 CIGAR operations: the placeholder in the record, the CIGAR in a CG:B,I tag (kd_bam_real_cigar on the device)
    n_ops = 66000
    cig = np.empty(n_ops, np.uint32)
    cig[0::2] = (1 << 4) | 0
    cig[1::2] = (1 << 4) | 2
    sl = n_ops // 2
    one = dict(contig=np.zeros(1, np.uint32), pos0=np.zeros(1, np.int32), flag=np.zeros(1, np.uint32), seq_off=np.zeros(1, np.uint64),
               seq_len=np.asarray([sl], np.uint32), cig_off=np.zeros(1, np.uint64), n_cig=np.asarray([n_ops], np.uint32),
               seq4=np.full((sl + 1) // 2 + 8, 0x11, np.uint8), cigar=np.concatenate([cig, np.zeros(2, np.uint32)]),
               contig_lens=np.asarray([200000], np.uint32), contig_names=np.asarray(["c"]))
    r = str(tmp_path / "cg.bam")
    synth.write_bam(r, one)
    assert both_ways(hip_lib, r)["kept"] == 1


@pytest.mark.gpu
def test_gpu_ingest_api_consensus_on_the_gpu(hip_lib, tmp_path, monkeypatch):
    from kindel_amd import kindel
    batch = synth.to_numpy(synth.short_reads([30000, 5000], 60, seed=11))
    p = str(tmp_path / "x.bam")
    N.write_bam(p, batch, lib=hip_lib)
    host = kindel.bam_to_consensus(p)
    monkeypatch.setenv("KINDEL_INGEST", "gpu")
    assert kindel.pileup_file(p).ingest["path"] == "gpu"
    gpu = kindel.bam_to_consensus(p)
    assert [str(s.sequence) for s in gpu.consensuses] == [str(s.sequence) for s in host.consensuses]
    assert gpu.refs_reports == host.refs_reports and gpu.refs_changes == host.refs_changes


@pytest.mark.gpu
def test_reference_bams_on_the_gpu(hip_lib):
    """The reference's own htslib-written BAM files (aux tags, real read names, bwa / minimap2 / segemehl record layouts, the 6.1 Mbp
    bact.tiny) through k_gpu_inflate + k_bgzf_crc + k_bam_* ON the MI355X vs the host decoder feeding the same engine: every
    table of every contig, insertion dicts, first-appearance order, record counts.  (Staged by oracle/make_ref.py.)"""
    from oracle import make_ref
    paths = make_ref.fixtures("*.bam")
    assert len(paths) >= 15, "oracle/_ref/fixtures is missing: run __graft_entry__.build() where /root/reference exists"
    for p in paths:
        info = both_ways(hip_lib, p)
        assert info["records"] > 0, p


@pytest.mark.gpu
@pytest.mark.parametrize("key,tag", __import__("tests.refcheck", fromlist=["x"]).FASTA_CASES)
def test_reference_fasta_through_the_device_side_ingest_on_the_gpu(hip_lib, tmp_path, monkeypatch, key, tag):
    """The reference's 21 golden FASTA files (default and -r) from the reference's OWN input files with KINDEL_INGEST=gpu on the
    MI355X (/root/reference/tests/test_kindel.py:114-278).  BAM inputs must really take the device-side path; the three SAM
    inputs are text and go to the host decoder."""
    from kindel_amd import kindel as K
    from tests import refcheck as RC
    path = RC.reference_input(key)
    assert path, "oracle/_ref/fixtures is missing: run __graft_entry__.build() where /root/reference exists"
    monkeypatch.setenv("KINDEL_INGEST", "gpu")
    took = K.pileup_file(path).ingest.get("path")
    assert took == ("gpu" if path.endswith(".bam") else took) and (path.endswith(".bam") or took != "gpu")
    RC.check_reference_fasta(K, tmp_path, key, tag, path=path)


@pytest.mark.gpu
def test_full_size_c3_through_the_device_side_ingest(hip_lib, tmp_path):
    """BASELINE.json's headline config as a FILE: 16.7 M reads, 2 GB of BGZF with Phred-like qualities -> device-side ingest -> pileup
    -> consensus, against the oracle's consensus of the same reads and the host decoder's record count."""
    import torch
    from kindel_amd import kindel as K
    from oracle import oracle as ko
    tb = synth.make("C3", device="cuda:0")
    batch = synth.to_numpy(tb)
    del tb
    torch.cuda.empty_cache()
    p = str(tmp_path / "c3_full.bam")
    os.environ["KD_WRITE_BAM_QUAL"] = "phred"
    try:
        N.write_bam(p, batch, lib=hip_lib)
    finally:
        os.environ.pop("KD_WRITE_BAM_QUAL", None)
    pl = K.pileup_file(p, ingest="gpu")
    assert pl.ingest["path"] == "gpu" and pl.ingest["kept"] == len(batch["contig"])
    done = K._device_consensus_all(pl, {c: None for c in pl.order}, False, 1, False)
    oa = ko.parse_records(batch, 0)
    assert done[pl.order[0]][0] == oa.consensus_sequence()[0]
    t = np.asarray(pl.tables(pl.order[0]))
    assert np.array_equal(t[0:5, :oa.L].T, oa.weights) and np.array_equal(t[5], oa.deletions) and np.array_equal(t[18], oa.ins_totals)
    pl.engine.close()


def test_a_file_too_big_for_the_device_goes_to_the_host_decoder(emu_lib, api_on_emu, tmp_path, monkeypatch):
    # file + inflated stream + batch are resident at once on the device-side path: what does not fit takes the streamed host decoder
    from kindel_amd import kindel
    batch = synth.to_numpy(synth.short_reads([4000], 30, seed=2))
    p = str(tmp_path / "x.bam")
    synth.write_bam(p, batch)
    monkeypatch.setenv("KD_EMU_FREE_BYTES", "100000")
    with N.BgzfPlan(p, lib=emu_lib) as plan:
        eng = N.Engine(plan.contig_lens, lib=emu_lib)
        with pytest.raises(N.UnsupportedByGpuIngest, match="does not fit"):
            eng.push_bam_gpu(plan)
        eng.close()
    monkeypatch.setenv("KINDEL_INGEST", "gpu")
    pl = kindel.pileup_file(p)
    assert getattr(pl, "ingest", {}).get("path") != "gpu" and int(np.asarray(pl.tables(0))[:5].sum()) > 0


def test_plan_edge_cases(emu_lib, tmp_path):
    """kd_bgzf_plan_open: a BAM header that spans several BGZF blocks (a thousand contigs, 300-byte blocks), a truncated file, plain
    gzip, a context whose contig table is not the file's."""
    import gzip
    lens = [1000 + 7 * i for i in range(1000)]
    batch = synth.to_numpy(synth.short_reads([lens[0], lens[1]], 20, seed=3))
    batch = dict(batch, contig_lens=np.asarray(lens, np.uint32), contig_names=np.asarray(["contig_number_%04d" % i for i in range(1000)]))
    p = str(tmp_path / "many.bam")
    synth.write_bam(p, batch, block_bytes=300)
    with N.BgzfPlan(p, lib=emu_lib) as plan:
        assert plan.contig_names[999] == "contig_number_0999" and plan.contig_lens.tolist() == lens
    both_ways(emu_lib, p)
    raw = open(p, "rb").read()
    q = str(tmp_path / "cut.bam")
    open(q, "wb").write(raw[:len(raw) // 2 + 5])
    with pytest.raises(OSError):
        N.BgzfPlan(q, lib=emu_lib)
    g = str(tmp_path / "plain.bam.gz")
    with gzip.open(g, "wb") as fh:
        fh.write(b"BAM\x01" + b"\0" * 100)
    with pytest.raises(N.UnsupportedByGpuIngest):
        N.BgzfPlan(g, lib=emu_lib)
    with N.BgzfPlan(p, lib=emu_lib) as plan:
        eng = N.Engine(np.asarray([5, 6, 7], np.uint32), lib=emu_lib)
        with pytest.raises(Exception, match="@SQ table"):
            eng.push_bam_gpu(plan)
        eng.close()
