"""Checks shared by the emulated (CPU) and the `-m gpu` suites: the Python API against goldens the REFERENCE holds or
produced (tests/golden/, written by oracle/make_golden.py)."""
import json
import os

import numpy as np

from tools import synth
from oracle import oracle as ko
from tests import parity as P

#: the FASTA files /root/reference/tests/test_kindel.py compares `kindel consensus` / `kindel consensus -r` with
#: (:114-124, :127-140, :143-158, :161-178, :181-238, :241-278; the -r golden of 3.issue23.bc75 is the reference's own
#: disabled test, :281-299)
REF_FASTA = json.load(open(os.path.join(P.GOLD, "reference_fasta.json")))
FASTA_CASES = [(k, tag) for k in sorted(REF_FASTA) for tag in sorted(REF_FASTA[k])
               if not (k == "ext__3.issue23.bc75" and tag == "realign")]
FEATURE_KEYS = ["bwa_mem__1.1.sub_test", "ext__3.issue23.bc75", "ext__1.issue23.debug"]


def reference_input(key):
    """The reference's own file behind fixture key "<dir>__<stem>" (data_<dir>/<stem>.bam or .sam), from /root/reference or its
    staged copy (oracle/make_ref.py); None when neither is on this box."""
    from oracle import make_ref
    d, stem = key.split("__", 1)
    return make_ref.fixture("data_%s/%s.bam" % (d, stem)) or make_ref.fixture("data_%s/%s.sam" % (d, stem))


def bam_of(tmp_path, key):
    p = str(tmp_path / (key + ".bam"))
    if not os.path.exists(p):
        synth.write_bam(p, P.load_fixture(key), sort_order="unknown")
    return p


def check_reference_fasta(K, tmp_path, key, tag, path=None):
    """What the reference's CLI tests assert: same record names, sequences equal ignoring case (cli.py defaults:
    min_overlap 7)."""
    res = K.bam_to_consensus(path or bam_of(tmp_path, key), realign=(tag == "realign"), min_overlap=7)
    got = {c.name: c.sequence for c in res.consensuses}
    want = REF_FASTA[key][tag]
    assert set(want) <= set(got), (key, tag, sorted(want), sorted(got))
    for name, seq in want.items():
        assert got[name].upper() == seq.upper(), (key, tag, name)


def check_features(K, tmp_path, key):
    """features() (kindel.py:633-664) column by column against the DataFrame the reference returned."""
    df = K.features(bam_of(tmp_path, key))
    g = np.load(os.path.join(P.GOLD, "features_%s.npz" % key), allow_pickle=True)
    assert list(df.columns) == [str(c) for c in g["columns"]]
    for c in df.columns:
        a, b = df[c].to_numpy(), g[c]
        if a.dtype.kind == "f":
            assert np.allclose(a, b, rtol=0, atol=1e-12, equal_nan=True), (key, c)
        elif a.dtype == object:
            assert (a.astype(str) == b.astype(str)).all(), (key, c)
        else:
            assert np.array_equal(a, b), (key, c)


def check_derived_arrays(K, tmp_path, key, gold):
    """The derived fields of `alignment` (kindel.py:83-96): consensus_depth / clip_depth against the reference's digests,
    clip_start_depth / clip_end_depth / aligned depth against the oracle."""
    alns = K.parse_bam(bam_of(tmp_path, key))
    batch = P.load_fixture(key)
    names = [str(x) for x in batch["contig_names"]]
    recs = {c["name"]: c for c in gold[key]["contigs"]}
    assert list(alns) == [c["name"] for c in gold[key]["contigs"]]
    for ref_id, aln in alns.items():
        g = recs[ref_id]
        assert P.sha(np.asarray(aln.consensus_depth, np.uint32)) == g["sha"]["consensus_depth"], (key, ref_id)
        assert P.sha(np.asarray(aln.clip_depth, np.uint32)) == g["sha"]["clip_depth"], (key, ref_id)
        d = ko.parse_records(batch, names.index(ref_id)).derived()
        assert np.array_equal(np.asarray(aln.clip_start_depth, np.uint32), d["clip_start_depth"])
        assert np.array_equal(np.asarray(aln.clip_end_depth, np.uint32), d["clip_end_depth"])
        assert np.array_equal(np.asarray(aln.consensus_depth, np.uint32), d["consensus_depth"])
        assert np.array_equal(np.asarray(aln.clip_depth, np.uint32), d["clip_depth"])
