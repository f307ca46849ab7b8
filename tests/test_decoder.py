"""Native SAM/BAM decoder (kd_decode.cpp) against the independent pure-Python reader (oracle/samio_py.py)."""
import os

import numpy as np
import pytest

from kindel_amd import _native as N
from tools import synth
from oracle import quirk_cases, samio_py
from tests import parity as P

KEYS = ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig")


def same_batch(a, b):
    for k in KEYS:
        assert np.array_equal(np.asarray(a[k]).astype(np.int64), np.asarray(b[k]).astype(np.int64)), k
    assert a["seq4"][: int(a["seq_off"][-1]) if len(a["seq_off"]) else 0].tobytes() == \
        b["seq4"][: int(b["seq_off"][-1]) if len(b["seq_off"]) else 0].tobytes()
    nw = int(np.asarray(a["n_cig"]).sum())
    assert np.array_equal(a["cigar"][:nw], b["cigar"][:nw])
    assert [str(x) for x in a["contig_names"]] == [str(x) for x in b["contig_names"]]
    assert np.array_equal(a["contig_lens"], b["contig_lens"])


@pytest.mark.parametrize("name", ["plain_M", "lowercase_bases", "star_rname_dropped", "ERR_cigar_star_mapped",
                                  "many_ops_long_cigar", "unmapped_and_short_reads_skipped", "P_eq_X_ops"])
def test_sam_text(tmp_path, name):
    p = tmp_path / "x.sam"
    p.write_text(quirk_cases.CASES[name])
    same_batch(N.decode_file(p), samio_py.load_batch(str(p)))


def test_bam_roundtrip_synthetic(tmp_path):
    batch = synth.to_numpy(synth.short_reads([3000, 1500], 20, seed=7))
    p = tmp_path / "x.bam"
    synth.write_bam(str(p), batch)
    for threads in (1, 4):
        got = N.decode_file(p, threads=threads)
        same_batch(got, batch)
        assert got["n_records"] == len(batch["contig"])
    same_batch(samio_py.load_batch(str(p)), batch)


def test_bam_long_reads_roundtrip(tmp_path):
    batch = synth.to_numpy(synth.long_reads([40000], 5, seed=9))
    p = tmp_path / "l.bam"
    synth.write_bam(str(p), batch)
    same_batch(N.decode_file(p), batch)


@pytest.mark.parametrize("range_bytes", ["64", "700", "5000"])
def test_bam_parallel_record_scan(tmp_path, monkeypatch, range_bytes):
    """Pass 1 of the BAM parser walks the block_size chain in parallel ranges with speculative starts; forcing tiny
    ranges (smaller than a record / a few records) exercises the speculation, the empty ranges and the hand-off check.
    A truncated file must report the same error as the sequential walk."""
    batch = synth.to_numpy(synth.short_reads([3000, 1500], 40, seed=11))
    lr = synth.to_numpy(synth.long_reads([40000], 5, seed=9))
    p, q = tmp_path / "x.bam", tmp_path / "l.bam"
    synth.write_bam(str(p), batch)
    synth.write_bam(str(q), lr)
    monkeypatch.setenv("KD_DECODE_RANGE_BYTES", range_bytes)
    for threads in (0, 3):
        same_batch(N.decode_file(p, threads=threads), batch)
        same_batch(N.decode_file(q, threads=threads), lr)


def test_sam_parallel_line_ranges(tmp_path, monkeypatch):
    """The SAM text parser splits the alignment lines into line-aligned ranges parsed in parallel: same batch as the
    pure-Python reader whatever the range size; errors name the first failing line; a header line after the first
    alignment is rejected."""
    ref = os.path.join(P.REF_TESTS, "data_ext", "1.issue23.debug.sam")
    texts = {"many": "@SQ\tSN:c\tLN:500\n" + "".join(
        "r%d\t0\tc\t%d\t60\t%dM%dI%dM\t*\t0\t0\t%s\t*\n" % (i, 1 + i % 300, 5 + i % 7, 1 + i % 3, 4, "ACGTN"[i % 5] * (10 + i % 7 + i % 3))
        for i in range(400))}
    if os.path.exists(ref):
        texts["issue23"] = open(ref).read()
    for name, text in texts.items():
        p = tmp_path / (name + ".sam")
        p.write_text(text)
        want = samio_py.load_batch(str(p))
        for rb in ("64", "1000", "100000000"):
            monkeypatch.setenv("KD_DECODE_RANGE_BYTES", rb)
            for threads in (0, 3):
                same_batch(N.decode_file(p, threads=threads), want)
    monkeypatch.setenv("KD_DECODE_RANGE_BYTES", "64")
    bad = tmp_path / "bad.sam"
    lines = texts["many"].split("\n")
    lines[200] = "broken\tline"
    lines[300] = "r\t0\tnope\t1\t60\t5M\t*\t0\t0\tACGTA\t*"
    bad.write_text("\n".join(lines))
    with pytest.raises(OSError, match="fewer than 10 fields"):
        N.decode_file(bad)
    late = tmp_path / "late.sam"
    late.write_text(texts["many"] + "@SQ\tSN:d\tLN:5\n")
    with pytest.raises(OSError, match="header line after"):
        N.decode_file(late)


def test_unreadable_file(tmp_path):
    with pytest.raises(OSError):
        N.decode_file(tmp_path / "missing.bam")
    bad = tmp_path / "bad.bam"
    bad.write_bytes(b"\x1f\x8b\x08\x04garbage")
    with pytest.raises(OSError):
        N.decode_file(bad)


@pytest.mark.skipif(not os.path.isdir(P.REF_TESTS), reason="reference fixtures only exist in the build container")
@pytest.mark.parametrize("rel", ["data_bwa_mem/1.1.sub_test.bam", "data_minimap2/1.1.multi.bam",
                                 "data_minimap2/hxb2-gp120-mutated.bam", "data_minimap2_bact/bact.tiny.bam",
                                 "data_ext/1.issue23.debug.sam", "data_segemehl/6.1.sub_test.bam"])
def test_reference_fixture_files(rel):
    path = os.path.join(P.REF_TESTS, rel)
    key = rel.replace("data_", "").replace("/", "__").rsplit(".", 1)[0]
    got = N.decode_file(path)
    same_batch(got, P.load_fixture(key))


_NIBS = "=ACMGRSVTWYHKDBN"


def _write_sam(path, batch):
    """batch -> SAM text (what the fixtures were decoded from, minus names / qualities)"""
    names = [str(x) for x in batch["contig_names"]]
    with open(path, "w") as fh:
        fh.write("@HD\tVN:1.6\tSO:unknown\n")
        for n, l in zip(names, batch["contig_lens"]):
            fh.write("@SQ\tSN:%s\tLN:%d\n" % (n, int(l)))
        for i in range(len(batch["contig"])):
            so, sl, co, nc = int(batch["seq_off"][i]), int(batch["seq_len"][i]), int(batch["cig_off"][i]), int(batch["n_cig"][i])
            by = batch["seq4"][so: so + (sl + 1) // 2]
            seq = "".join(_NIBS[b >> 4] + _NIBS[b & 15] for b in by.tolist())[:sl] or "*"
            cig = "".join("%d%s" % (int(w) >> 4, "MIDNSHP=X"[int(w) & 15]) for w in batch["cigar"][co: co + nc].tolist()) or "*"
            fh.write("r%d\t%d\t%s\t%d\t60\t%s\t*\t0\t0\t%s\t*\n" % (i, int(batch["flag"][i]), names[int(batch["contig"][i])],
                                                                     int(batch["pos0"][i]) + 1, cig, seq))


def _concat_stream(st):
    parts = []
    while True:
        b = st.next_batch()
        if b is None:
            break
        parts.append(b)
    keys = ("contig", "pos0", "flag", "seq_len", "n_cig")
    out = {k: np.concatenate([p[k] for p in parts]) if parts else np.zeros(0) for k in keys}
    # per-read payloads (offsets are chunk-local)
    seqs, cigs = [], []
    for p in parts:
        for so, sl, co, nc in zip(p["seq_off"].tolist(), p["seq_len"].tolist(), p["cig_off"].tolist(), p["n_cig"].tolist()):
            seqs.append(p["seq4"][so: so + (sl + 1) // 2].tobytes())
            cigs.append(p["cigar"][co: co + nc].tobytes())
    return out, seqs, cigs, len(parts)


@pytest.mark.parametrize("key,kind,chunk", [("bwa_mem__1.1.sub_test", "bam", 3000), ("minimap2__1.1.multi", "bam", 700),
                                            ("minimap2__hxb2-gp120-mutated", "bam", 20000), ("segemehl__3.1.sub_test", "sam", 2500),
                                            ("ext__2.issue23.bc63", "bam", 1 << 20)])
def test_stream_chunks_equal_whole_file(emu_lib, tmp_path, key, kind, chunk):
    """kd_stream_*: batches of ~chunk uncompressed bytes (records cut by a chunk boundary are carried over) == one decode."""
    batch = P.load_fixture(key)
    path = str(tmp_path / ("s." + kind))
    if kind == "bam":
        synth.write_bam(path, batch, sort_order="unknown")
    else:
        _write_sam(path, batch)
    whole = N.decode_file(path, lib=emu_lib)
    st = N.Stream(path, chunk_bytes=chunk, lib=emu_lib)
    assert list(st.contig_names) == [str(x) for x in whole["contig_names"]] and np.array_equal(st.contig_lens, whole["contig_lens"])
    got, seqs, cigs, n_parts = _concat_stream(st)
    assert st.n_records() == whole["n_records"]
    st.close()
    assert n_parts > (1 if chunk < 100000 else 0)
    for k in ("contig", "pos0", "flag", "seq_len", "n_cig"):
        assert np.array_equal(got[k], whole[k]), k
    for i, (so, sl, co, nc) in enumerate(zip(whole["seq_off"].tolist(), whole["seq_len"].tolist(), whole["cig_off"].tolist(),
                                             whole["n_cig"].tolist())):
        assert seqs[i] == whole["seq4"][so: so + (sl + 1) // 2].tobytes(), i
        assert cigs[i] == whole["cigar"][co: co + nc].tobytes(), i


@pytest.mark.parametrize("block_bytes,range_bytes,chunk", [(61, 64, 4000), (200, 64, 1500), (333, 700, 20000), (1000, 3000, 1 << 20),
                                                           (4096, 5000, 50000), (97, 300, 997)])
def test_stream_workers_inflate_and_walk_their_own_ranges(emu_lib, tmp_path, monkeypatch, block_bytes, range_bytes, chunk):
    """kd_stream: every worker inflates a run of whole BGZF blocks and walks the records in it; records it cannot judge at
    the end of its run (header behind what exists yet) are left to the hand-off pass.  Tiny blocks and tiny ranges put a
    range boundary inside almost every record; the result must equal the two-phase path and the whole-file decode."""
    batch = P.load_fixture("minimap2__1.1.multi")
    path = str(tmp_path / "t.bam")
    synth.write_bam(path, batch, sort_order="unknown", block_bytes=block_bytes)
    monkeypatch.setenv("KD_DECODE_RANGE_BYTES", str(range_bytes))
    whole = N.decode_file(path, lib=emu_lib)
    same_batch(whole, batch)
    outs = []
    for unfused in (False, True):
        if unfused:
            monkeypatch.setenv("KD_DECODE_UNFUSED", "1")
        st = N.Stream(path, chunk_bytes=chunk, threads=5, lib=emu_lib)
        got, seqs, cigs, n_parts = _concat_stream(st)
        assert st.n_records() == whole["n_records"]
        st.close()
        outs.append((got, seqs, cigs))
    for k in ("contig", "pos0", "flag", "seq_len", "n_cig"):
        assert np.array_equal(outs[0][0][k], whole[k]), k
        assert np.array_equal(outs[1][0][k], whole[k]), k
    assert outs[0][1] == outs[1][1] and outs[0][2] == outs[1][2]
    for i, (so, sl) in enumerate(zip(whole["seq_off"].tolist(), whole["seq_len"].tolist())):
        assert outs[0][1][i] == whole["seq4"][so: so + (sl + 1) // 2].tobytes(), i


@pytest.mark.parametrize("key,chunk", [("bwa_mem__2.1.sub_test", 5000), ("minimap2__1.1.multi", 900), ("ext__1.issue23.debug", 30000)])
def test_streamed_pileup_equals_whole_file_pileup(api_on_emu, tmp_path, key, chunk):
    """kd_push_stream (decode thread + pushing thread, many small batches whose boundaries cut windows) == one batch."""
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
    assert [c.sequence for c in ra.consensuses] == [g["consensus"] for g in P.golden_outputs()[key]["contigs"]]


@pytest.mark.parametrize("key,kind", [("bwa_mem__1.1.sub_test", "bam"), ("minimap2__1.1.multi", "bam"), ("segemehl__3.1.sub_test", "sam"),
                                      ("minimap2__hxb2-gp120-mutated", "bam")])
def test_whole_file_decode_is_the_streams_chunks_appended(emu_lib, tmp_path, monkeypatch, key, kind):
    """kd_decode_open reads the file through the stream (64 MiB chunks appended to one another); with tiny chunks the result
    must equal the one-chunk decode (KD_DECODE_ONE_CHUNK) and the fixture."""
    batch = P.load_fixture(key)
    path = str(tmp_path / ("w." + kind))
    if kind == "bam":
        synth.write_bam(path, batch, sort_order="unknown", block_bytes=900)
    else:
        _write_sam(path, batch)
    monkeypatch.setenv("KD_DECODE_ONE_CHUNK", "1")
    one = N.decode_file(path, lib=emu_lib)
    monkeypatch.delenv("KD_DECODE_ONE_CHUNK")
    for chunk in (1500, 40000):
        monkeypatch.setenv("KD_DECODE_CHUNK_BYTES", str(chunk))
        got = N.decode_file(path, threads=3, lib=emu_lib)
        same_batch(got, one)
        same_batch(got, batch)
        assert got["n_records"] == one["n_records"] and list(got["contig_names"]) == list(one["contig_names"])


@pytest.mark.parametrize("cut", [0.999, 0.9, 0.5, 0.1])
def test_truncated_bgzf_file_is_an_error_not_a_crash(emu_lib, tmp_path, cut):
    """The stream maps the file and scans its blocks as it goes: a file cut inside a block (or inside the header) must end in
    OSError from kd_stream_open / kd_stream_next, as it does from the whole-file decoder."""
    batch = P.load_fixture("bwa_mem__1.1.sub_test")
    full = str(tmp_path / "f.bam")
    synth.write_bam(full, batch, sort_order="unknown", block_bytes=700)
    data = open(full, "rb").read()
    path = str(tmp_path / "t.bam")
    with open(path, "wb") as fh:
        fh.write(data[: int(len(data) * cut)])
    with pytest.raises(OSError):
        N.decode_file(path, lib=emu_lib)
    with pytest.raises(OSError):
        st = N.Stream(path, chunk_bytes=5000, lib=emu_lib)
        while st.next_batch() is not None:
            pass


def _run_in_child(code):
    """The failure mode under test is a SIGSEGV of the process: run the decoder in a child and look at how it ended."""
    import subprocess
    import sys
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=P.ROOT)


@pytest.mark.parametrize("where", ["alone", "after_a_valid_block"])
def test_gzip_header_whose_extra_field_points_past_the_file_end(emu_lib, tmp_path, where):
    """A gzip member header with FEXTRA and xlen = 0xffff in the last bytes of a (memory-mapped) file: the subfield walk must
    not leave the mapping.  OSError, not a crash -- from the whole-file decoder and from the stream."""
    evil = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 0xff, 0xff]) + b"BC\x02\x00\x10\x00"      # 18 bytes, xlen 65535
    head = b""
    if where == "after_a_valid_block":
        full = str(tmp_path / "f.bam")
        synth.write_bam(full, P.load_fixture("minimap2__1.1.multi"), sort_order="unknown", block_bytes=700)
        data = open(full, "rb").read()
        head = data[:-28]                      # everything but the BGZF end-of-file marker
    path = str(tmp_path / "evil.bam")
    with open(path, "wb") as fh:
        fh.write(head + evil)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from kindel_amd import _native as N\n"
            "lib = N.Library(%r)\n"
            "for how in ('file', 'stream'):\n"
            "    try:\n"
            "        if how == 'file': N.decode_file(%r, lib=lib)\n"
            "        else:\n"
            "            st = N.Stream(%r, chunk_bytes=4096, lib=lib)\n"
            "            while st.next_batch() is not None: pass\n"
            "        print(how, 'no error')\n"
            "    except OSError as e: print(how, 'OSError')\n") % (P.ROOT, emu_lib.path, path, path)
    r = _run_in_child(code)
    assert r.returncode == 0, "decoder crashed (rc %d): %s" % (r.returncode, r.stderr[-400:])
    assert r.stdout.split() == ["file", "OSError", "stream", "OSError"], r.stdout


def test_native_bam_writer_roundtrip(emu_lib, tmp_path):
    """kd_write_bam (parallel record layout + parallel deflate) -> decoder == the batch; also readable by the pure-Python reader;
    a read with more than 65535 CIGAR operations goes through the CG tag."""
    batch = synth.to_numpy(synth.short_reads([3000, 1500], 30, seed=17))
    p = tmp_path / "n.bam"
    for threads in (1, 5):
        N.write_bam(str(p), batch, threads=threads, lib=emu_lib)
        same_batch(N.decode_file(p, lib=emu_lib), batch)
    same_batch(samio_py.load_batch(str(p)), batch)
    lr = synth.to_numpy(synth.long_reads([40000], 5, seed=9))
    N.write_bam(str(p), lr, lib=emu_lib)
    same_batch(N.decode_file(p, lib=emu_lib), lr)
    empty = dict(batch)
    for k in KEYS:
        empty[k] = batch[k][:0]
    N.write_bam(str(p), empty, lib=emu_lib)
    got = N.decode_file(p, lib=emu_lib)
    assert len(got["contig"]) == 0 and list(got["contig_lens"]) == [3000, 1500]


@pytest.mark.parametrize("key,block_bytes,parts", [("bwa_mem__1.1.sub_test", 700, 5), ("minimap2__1.1.multi", 300, 8),
                                                    ("ext__1.issue23.debug", 4000, 3), ("bwa_mem__3.1.sub_test", 65280, 4)])
def test_spans_of_a_bam_file_tile_it_exactly(emu_lib, tmp_path, key, block_bytes, parts, monkeypatch):
    """kd_decode_open_span: cutting the file's BGZF blocks into `parts` byte shares, the spans' records concatenated are the
    file's records in order -- every record exactly once, whichever block a cut falls into (records straddle blocks) -- and the
    end offset of each span is the start offset of the next (what the ranks verify with one small all-gather)."""
    monkeypatch.setenv("KD_DECODE_RANGE_BYTES", "2000")      # several worker ranges per chunk even on these small files
    batch = P.load_fixture(key)
    path = str(tmp_path / "s.bam")
    synth.write_bam(path, batch, sort_order="unknown", block_bytes=block_bytes)
    whole = N.decode_file(path, lib=emu_lib)
    off = N.bgzf_index(path, lib=emu_lib)
    size = os.path.getsize(path)
    cuts = [0] + [int(np.searchsorted(off, size * p // parts)) for p in range(1, parts)] + [len(off)]
    got = {k: [] for k in ("contig", "pos0", "flag", "seq_len", "n_cig")}
    seqs, cigs, spans, n_rec = [], [], [], 0
    for p in range(parts):
        sp = N.decode_span(path, cuts[p], cuts[p + 1], threads=3, lib=emu_lib)
        spans.append(sp["span"])
        n_rec += sp["span"][2]
        for k in got:
            got[k].append(np.asarray(sp[k]).copy())
        for so, sl in zip(sp["seq_off"].tolist(), sp["seq_len"].tolist()):
            seqs.append(sp["seq4"][so: so + (sl + 1) // 2].tobytes())
        for co, nc in zip(sp["cig_off"].tolist(), sp["n_cig"].tolist()):
            cigs.append(sp["cigar"][co: co + nc].tobytes())
        assert list(sp["contig_names"]) == list(whole["contig_names"])
    for a, b in zip(spans, spans[1:]):
        assert a[1] == b[0]
    assert spans[-1][3] == 1 and n_rec == whole["n_records"]
    for k in got:
        assert np.array_equal(np.concatenate(got[k]), whole[k]), k
    for i, (so, sl) in enumerate(zip(whole["seq_off"].tolist(), whole["seq_len"].tolist())):
        assert seqs[i] == whole["seq4"][so: so + (sl + 1) // 2].tobytes(), i
    for i, (co, nc) in enumerate(zip(whole["cig_off"].tolist(), whole["n_cig"].tolist())):
        assert cigs[i] == whole["cigar"][co: co + nc].tobytes(), i
    # single-block spans and an empty one
    one = N.decode_span(path, 3, 4, lib=emu_lib)
    assert one["span"][0] <= one["span"][1]
    assert N.decode_span(path, len(off), len(off) + 5, lib=emu_lib)["span"][2] == 0


def test_span_decoder_refuses_what_is_not_bgzf(emu_lib, tmp_path):
    p = tmp_path / "x.sam"
    p.write_text("@SQ\tSN:c\tLN:10\nr\t0\tc\t1\t60\t4M\t*\t0\t0\tACGT\t*\n")
    with pytest.raises(OSError):
        N.bgzf_index(str(p), lib=emu_lib)
    with pytest.raises(OSError):
        N.decode_span(str(p), 0, 1, lib=emu_lib)
