"""The Python API through the kernel emulator against what the reference holds / returned: its 21 golden FASTA files,
features() DataFrames, the derived `alignment` arrays; plus decoder / host behaviours found in review."""
import os

import numpy as np
import pytest

from kindel_amd import _native as N
from tools import synth
from tests import parity as P
from tests import refcheck as RC

GOLD = P.golden_outputs()


@pytest.mark.parametrize("key,tag", RC.FASTA_CASES)
def test_reference_fasta(api_on_emu, tmp_path, key, tag):
    from kindel_amd import kindel as K
    RC.check_reference_fasta(K, tmp_path, key, tag)


@pytest.mark.parametrize("key,tag", RC.FASTA_CASES)
def test_reference_fasta_through_the_device_side_ingest(api_on_emu, tmp_path, monkeypatch, key, tag):
    """The same 21 golden FASTA files of the reference with KINDEL_INGEST=gpu: BGZF inflate, BAM record walk and the batch arrays by
    the device-side kernels (emulated here), then the same engine."""
    from kindel_amd import kindel as K
    monkeypatch.setenv("KINDEL_INGEST", "gpu")
    assert K.pileup_file(RC.bam_of(tmp_path, key)).ingest.get("path") == "gpu"
    RC.check_reference_fasta(K, tmp_path, key, tag)


@pytest.mark.parametrize("key", RC.FEATURE_KEYS)
def test_features_dataframe(api_on_emu, tmp_path, key):
    from kindel_amd import kindel as K
    RC.check_features(K, tmp_path, key)


@pytest.mark.parametrize("key", ["bwa_mem__2.1.sub_test", "minimap2__1.1.multi", "ext__1.issue23.debug"])
def test_derived_alignment_arrays(api_on_emu, tmp_path, key):
    from kindel_amd import kindel as K
    RC.check_derived_arrays(K, tmp_path, key, GOLD)


@pytest.mark.skipif(not os.path.isdir(os.path.join(P.REF_TESTS, "data_bwa_mem")), reason="needs /root/reference")
@pytest.mark.parametrize("rel", ["data_bwa_mem/3.1.sub_test", "data_minimap2/1.1.multi", "data_ext/2.issue23.bc63"])
def test_reference_files_read_in_place(api_on_emu, rel):
    """The reference's own BAM / SAM files (not the decoded fixtures) through decoder + engine, against the FASTA next to them."""
    from kindel_amd import kindel as K
    path = os.path.join(P.REF_TESTS, rel + (".sam" if "data_ext" in rel else ".bam"))
    res = K.bam_to_consensus(path, min_overlap=7)
    want, name = {}, None
    for line in open(os.path.join(P.REF_TESTS, rel + ".fa")):
        if line.startswith(">"):
            name = line[1:].split()[0]; want[name] = ""
        else:
            want[name] += line.strip()
    got = {c.name: c.sequence for c in res.consensuses}
    for k, v in want.items():
        assert got[k].upper() == v.upper(), (rel, k)


def test_header_only_contigs_get_no_tables(api_on_emu, tmp_path):
    """Only contigs with records are laid out on the device (the reference allocates per RNAME seen, kindel.py:150-151):
    a header with two 2.4 Gbp contigs and one read on a 20 bp contig must work and cost 20 bp of tables."""
    from kindel_amd import kindel as K
    p = tmp_path / "h.sam"
    p.write_text("@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:big1\tLN:2500000000\n@SQ\tSN:big2\tLN:2400000000\n@SQ\tSN:tiny\tLN:20\n"
                 "r1\t0\ttiny\t3\t60\t8M\t*\t0\t0\tACGTACGT\tIIIIIIII\n")
    res = K.bam_to_consensus(str(p))
    assert [c.name for c in res.consensuses] == ["tiny_cns"]
    assert res.consensuses[0].sequence == "NNACGTACGTNNNNNNNNNN"
    alns = K.parse_bam(str(p))
    assert list(alns) == ["tiny"] and len(alns["tiny"].weights) == 20
    # first-appearance order survives the remapping to dense ids
    q = tmp_path / "o.sam"
    q.write_text("@SQ\tSN:a\tLN:30\n@SQ\tSN:unused\tLN:1000\n@SQ\tSN:b\tLN:30\n"
                 "r1\t0\tb\t1\t60\t4M\t*\t0\t0\tACGT\tIIII\nr2\t0\ta\t1\t60\t4M\t*\t0\t0\tTTTT\tIIII\n")
    assert [c.name for c in K.bam_to_consensus(str(q)).consensuses] == ["b_cns", "a_cns"]
    # no records at all: the reference returns empty results
    e = tmp_path / "e.sam"
    e.write_text("@SQ\tSN:a\tLN:30\n")
    assert K.bam_to_consensus(str(e)).consensuses == [] and len(K.parse_bam(str(e))) == 0


def test_unknown_rname_is_a_keyerror(api_on_emu, tmp_path):
    """refs_lens[ref_id] of kindel.py:151."""
    from kindel_amd import kindel as K
    p = tmp_path / "u.sam"
    p.write_text("@SQ\tSN:a\tLN:30\nr1\t0\tzzz\t1\t60\t4M\t*\t0\t0\tACGT\tIIII\n")
    with pytest.raises(KeyError) as ei:
        K.bam_to_consensus(str(p))
    assert ei.value.args[0] == "zzz"


def test_large_header_is_scanned_then_streamed(api_on_emu, tmp_path):
    """A header beyond STREAM_MAX_SITES (a human genome's @SQ table, reads on two small contigs in between): the file is scanned once
    for the contigs in use and then STREAMED over those alone -- several chunks, a contig map inside the decoder
    (kd_stream_set_contig_map) -- instead of being decoded whole; same tables, order and FASTA as the one-batch path."""
    from kindel_amd import kindel as K
    small = synth.to_numpy(synth.short_reads([3000, 2000], 40, seed=21, planted=False))
    batch = dict(small)
    batch["contig"] = np.where(small["contig"] == 0, 1, 3).astype(np.uint32)            # header slots 1 and 3 of five
    batch["contig_lens"] = np.asarray([2_100_000_000, 3000, 1_900_000_000, 2000, 50_000_000], np.uint32)
    names = ["chr1", "virusA", "chr2", "virusB", "chrUn"]
    p = str(tmp_path / "big_header.bam")
    synth.write_bam(p, batch, names=names, block_bytes=4000)
    pl = K.pileup_file(p, chunk_bytes=20000)                                             # many chunks: the map applies to every batch
    assert pl.names == ["virusA", "virusB"] and list(pl.lens) == [3000, 2000] and pl.ingest["batches"] > 3
    ref = K.pileup_file(p, stream=False)
    assert ref.names == pl.names and pl.order == ref.order
    for cid in pl.order:
        assert np.array_equal(pl.engine.tables(cid), ref.engine.tables(cid))
    a, b = K.bam_to_consensus(p), K.bam_to_consensus(p, min_depth=2)
    assert [c.name for c in a.consensuses] == ["virusA_cns", "virusB_cns"] and len(b.consensuses) == 2
    want = K.pileup_batch(small)
    for cid in (0, 1):
        assert np.array_equal(pl.engine.tables(cid), want.engine.tables(cid))
    # a record on an @SQ entry the map leaves out is refused by the decoder, not mis-filed
    st = N.Stream(p)
    st.set_contig_map(np.asarray([0xFFFFFFFF, 0, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF], np.uint32))
    with pytest.raises(N.KindelNativeError, match="which the map leaves out"):
        while st.next_batch() is not None:
            pass
    st.close()


def test_large_header_through_a_pipe_is_read_once(api_on_emu, tmp_path):
    """The same large header arriving through a FIFO (kindel consensus /dev/stdin): the default route must not open the input a
    second time -- the second pass would find an empty or blocked pipe -- and decodes it whole in one pass, like the reference."""
    import threading
    from kindel_amd import kindel as K
    small = synth.to_numpy(synth.short_reads([3000, 2000], 20, seed=22, planted=False))
    batch = dict(small)
    batch["contig"] = np.where(small["contig"] == 0, 1, 3).astype(np.uint32)
    batch["contig_lens"] = np.asarray([2_100_000_000, 3000, 1_900_000_000, 2000, 50_000_000], np.uint32)
    p = str(tmp_path / "big_header.bam")
    synth.write_bam(p, batch, names=["chr1", "virusA", "chr2", "virusB", "chrUn"], block_bytes=4000)
    want = K.bam_to_consensus(p)
    fifo = str(tmp_path / "in.fifo")
    os.mkfifo(fifo)

    def feed():
        with open(fifo, "wb") as out, open(p, "rb") as src:       # ONE writer, closed at the end: a second open would block
            out.write(src.read())
    th = threading.Thread(target=feed, daemon=True)
    th.start()
    got = K.bam_to_consensus(fifo)
    th.join(timeout=30)
    assert not th.is_alive()
    assert [(c.name, c.sequence) for c in got.consensuses] == [(c.name, c.sequence) for c in want.consensuses]


def test_sam_insertion_outside_the_bam_alphabet_is_refused(api_on_emu, tmp_path):
    """4-bit base codes cannot hold e.g. 'U'.  In M / clip context that is a KeyError like in the reference; inside an
    insertion the reference would keep the text verbatim (kindel.py:55-58): refused loudly instead of emitting '='."""
    from kindel_amd import kindel as K
    p = tmp_path / "i.sam"
    p.write_text("@SQ\tSN:a\tLN:30\nr1\t0\ta\t1\t60\t5M2I5M\t*\t0\t0\tACGTAUUCGTAC\tIIIIIIIIIIII\n"
                 "r2\t0\ta\t1\t60\t5M2I5M\t*\t0\t0\tACGTAUUCGTAC\tIIIIIIIIIIII\n")
    with pytest.raises(OSError, match="outside the BAM base alphabet"):
        K.bam_to_consensus(str(p))
    q = tmp_path / "m.sam"
    q.write_text("@SQ\tSN:a\tLN:30\nr1\t0\ta\t1\t60\t12M\t*\t0\t0\tACGTAUUCGTAC\tIIIIIIIIIIII\n")
    with pytest.raises(KeyError):
        K.bam_to_consensus(str(q))
    # IUPAC letters of the BAM alphabet inside an insertion stay legal (the reference does not check insertions)
    r = tmp_path / "r.sam"
    r.write_text("@SQ\tSN:a\tLN:30\nr1\t0\ta\t1\t60\t5M2I5M\t*\t0\t0\tACGTARYCGTAC\tIIIIIIIIIIII\n"
                 "r2\t0\ta\t1\t60\t5M2I5M\t*\t0\t0\tACGTARYCGTAC\tIIIIIIIIIIII\n")
    assert "ry" in K.bam_to_consensus(str(r)).consensuses[0].sequence


def test_long_cigar_in_cg_tag(emu_lib, tmp_path):
    """BAM stores a CIGAR of more than 65535 operations as <l_seq>S<ref_len>N plus a CG:B,I tag (SAMv1 4.2.2): the decoder
    must hand the real CIGAR to the engine."""
    n_ops = 70001
    ops = np.empty(n_ops, np.uint32)
    ops[0::2] = (1 << 4) | 0          # 1M
    ops[1::2] = (1 << 4) | 2          # 1D
    n_m = (n_ops + 1) // 2
    L = n_ops + 10
    rng = np.random.default_rng(3)
    codes = rng.choice(np.array([1, 2, 4, 8], np.uint8), n_m + (n_m & 1))
    seq4 = (codes[0::2] << 4 | codes[1::2]).astype(np.uint8)
    if n_m & 1:
        seq4[-1] &= 0xf0
    batch = dict(contig=np.zeros(2, np.uint32), pos0=np.asarray([3, 5], np.int32), flag=np.zeros(2, np.uint32),
                 seq_off=np.asarray([0, len(seq4)], np.uint64), seq_len=np.asarray([n_m, 4], np.uint32),
                 cig_off=np.asarray([0, n_ops], np.uint64), n_cig=np.asarray([n_ops, 1], np.uint32),
                 seq4=np.concatenate([seq4, np.asarray([0x12, 0x48], np.uint8), np.zeros(16, np.uint8)]),
                 cigar=np.concatenate([ops, np.asarray([(4 << 4) | 0, 0, 0], np.uint32)]),
                 contig_names=np.asarray(["c"]), contig_lens=np.asarray([L], np.uint32))
    path = str(tmp_path / "cg.bam")
    synth.write_bam(path, batch, sort_order="unknown")
    dec = N.decode_file(path, lib=emu_lib)
    assert dec["n_cig"].tolist() == [n_ops, 1]
    assert np.array_equal(dec["cigar"][:n_ops], ops) and int(dec["cigar"][n_ops]) == (4 << 4)
    P.assert_matches_oracle(P.Run(emu_lib, dict(dec, contig_names=batch["contig_names"]), window=256))
