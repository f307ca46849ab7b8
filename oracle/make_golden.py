"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/ from the unmodified reference.

Run in the build container (needs /root/reference):   python -m oracle.make_golden

Writes
  tests/golden/fixtures/<name>.npz   decoded SoA read batches of the reference's own test
                                     inputs (tests/data_*), so the GPU box -- which has no
                                     /root/reference -- can replay them through the C-ABI
  tests/golden/reference_outputs.json  what the reference returned for each of them:
                                     sha256 of every integer table, table sums, consensus
                                     (default and --realign), changes, report, CDR regions
  tests/golden/quirks.json           reference outputs for oracle/quirk_cases.py
  tests/golden/weights_*.npz         reference weights() DataFrames for two fixtures
and, while doing so, asserts that oracle/kindel_oracle.c reproduces every integer table,
insertion dict and consensus string exactly (this is what pins the oracle).
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

from oracle import oracle as ko
from oracle import quirk_cases, samio_py
from oracle.refrun import REF_ROOT, load_reference

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
CH = "ATGCN"

FIXTURES = [
    "data_bwa_mem/%d.1.sub_test.bam" % i for i in range(1, 7)] + [
    "data_segemehl/%d.1.sub_test.bam" % i for i in range(1, 7)] + [
    "data_minimap2/1.1.multi.bam", "data_minimap2/hxb2-gp120-mutated.bam",
    "data_minimap2_bact/bact.tiny.bam",
    "data_ext/1.issue23.debug.sam", "data_ext/2.issue23.bc63.sam", "data_ext/3.issue23.bc75.sam",
]
BIG_L = 100000  # above this only digests are stored and the slow reference paths are skipped


def fixture_key(rel):
    return rel.replace("data_", "").replace("/", "__").rsplit(".", 1)[0]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sha_s(s):
    return hashlib.sha256(s.encode()).hexdigest()


def wtab(list_of_dicts):
    return np.asarray([[d[c] for c in CH] for d in list_of_dicts], np.uint32).reshape(-1, 5)


def ins_list(insertions):
    """reference insertions (list of defaultdict) -> [(site, string, count)] in dict order"""
    return [(p, s, c) for p, d in enumerate(insertions) for s, c in d.items()]


def ins_digest(items):
    canon = sorted((int(p), s, int(c)) for p, s, c in items)
    return hashlib.sha256(json.dumps(canon).encode()).hexdigest()


def changes_str(changes):
    return "".join("." if c is None else c for c in changes)


def aln_record(aln):
    W, S, E = wtab(aln.weights), wtab(aln.clip_start_weights), wtab(aln.clip_end_weights)
    cs, ce, de = (np.asarray(x, np.uint32) for x in (aln.clip_starts, aln.clip_ends, aln.deletions))
    il = ins_list(aln.insertions)
    rec = dict(
        L=len(aln.weights),
        sha=dict(weights=sha(W), clip_start_weights=sha(S), clip_end_weights=sha(E),
                 clip_starts=sha(cs), clip_ends=sha(ce), deletions=sha(de), insertions=ins_digest(il),
                 consensus_depth=sha(np.asarray(aln.consensus_depth, np.uint32)),
                 clip_depth=sha(np.asarray(aln.clip_depth, np.uint32))),
        sums=dict(weights=int(W.sum()), deletions=int(de.sum()), ins_events=int(sum(c for _, _, c in il)),
                  ins_keys=len(il), ins_sites=len({p for p, _, _ in il}),
                  clip_starts=int(cs.sum()), clip_ends=int(ce.sum()),
                  clip_start_weights=int(S.sum()), clip_end_weights=int(E.sum())),
    )
    return rec, (W, S, E, cs, ce, de, il)


def check_oracle(tag, oa, tabs):
    W, S, E, cs, ce, de, il = tabs
    assert np.array_equal(oa.weights, W), tag + " weights"
    assert np.array_equal(oa.clip_start_weights, S), tag + " csw"
    assert np.array_equal(oa.clip_end_weights, E), tag + " cew"
    assert np.array_equal(oa.clip_starts, cs), tag + " clip_starts"
    assert np.array_equal(oa.clip_ends, ce), tag + " clip_ends"
    assert np.array_equal(oa.deletions, de), tag + " deletions"
    assert oa.insertions == [(p, s.upper(), c) for p, s, c in il], tag + " insertions"


def do_fixtures(K):
    out = {}
    os.makedirs(os.path.join(GOLD, "fixtures"), exist_ok=True)
    for rel in FIXTURES:
        path = os.path.join(REF_ROOT, "tests", rel)
        key = fixture_key(rel)
        print("fixture", key, flush=True)
        batch = samio_py.load_batch(path)
        np.savez_compressed(os.path.join(GOLD, "fixtures", key + ".npz"), **batch)
        alns = K.parse_bam(path)
        big = max(len(a.weights) for a in alns.values()) > BIG_L
        names = list(batch["contig_names"])
        contigs = []
        res = K.bam_to_consensus(path)
        res_r = None if big else K.bam_to_consensus(path, realign=True, min_overlap=7)
        for i, (ref_id, aln) in enumerate(alns.items()):
            rec, tabs = aln_record(aln)
            rec["name"] = ref_id
            oa = ko.parse_records(batch, names.index(ref_id))
            check_oracle(key + ":" + ref_id, oa, tabs)
            cns = res.consensuses[i].sequence
            ocns, och = oa.consensus_sequence()
            assert ocns == cns and och == res.refs_changes[ref_id], key + " oracle consensus"
            ad = [sum(w[c] for c in "ACGT") for w in aln.weights]
            assert oa.depth_minmax() == (min(ad), max(ad))
            rec["depth_minmax"] = [min(ad), max(ad)]
            rec["consensus_len"] = len(cns)
            rec["consensus_sha"] = sha_s(cns)
            rec["changes_sha"] = sha_s(changes_str(res.refs_changes[ref_id]))
            if not big:
                rec["consensus"] = cns
                rec["changes"] = changes_str(res.refs_changes[ref_id])
                rec["report"] = res.refs_reports[ref_id].replace(str(path), "{bam_path}")
                rcns = res_r.consensuses[i].sequence
                rec["realign_consensus"] = rcns
                rec["realign_changes"] = changes_str(res_r.refs_changes[ref_id])
                rec["realign_report"] = res_r.refs_reports[ref_id].replace(str(path), "{bam_path}")
                cdrps = K.cdrp_consensuses(aln.weights, aln.deletions, aln.clip_start_weights,
                                           aln.clip_end_weights, aln.clip_start_depth,
                                           aln.clip_end_depth, 0.1, 50)
                rec["cdrps_0.1_50"] = [[[r.start, r.end, r.seq, r.direction] for r in pair] for pair in cdrps]
                cdrps10 = K.cdrp_consensuses(aln.weights, aln.deletions, aln.clip_start_weights,
                                             aln.clip_end_weights, aln.clip_start_depth,
                                             aln.clip_end_depth, 0.1, 10)
                rec["cdrps_0.1_10"] = [[[r.start, r.end, r.seq, r.direction] for r in pair] for pair in cdrps10]
                for md in (0, 5, 20):  # min_depth sweep, through the reference's own function
                    s, ch = K.consensus_sequence(aln.weights, aln.insertions, aln.deletions, None,
                                                 md == 5, md, md == 20)
                    os_, och_ = oa.consensus_sequence(None, md == 5, md, md == 20)
                    assert (os_, och_) == (s, ch), key + " oracle consensus min_depth %d" % md
                    rec["consensus_sha_min_depth_%d" % md] = sha_s(s)
            contigs.append(rec)
        out[key] = dict(source="tests/" + rel, contigs=contigs,
                        n_records=int(len(batch["contig"])))
    return out


def do_quirks(K):
    out = {}
    for name, sam in quirk_cases.CASES.items():
        with tempfile.NamedTemporaryFile("w", suffix=".sam", delete=False) as fh:
            fh.write(sam)
            path = fh.name
        _, refs, recs = samio_py.read_alignment_file(path)
        batch = samio_py.records_to_batch(refs, recs)
        names = [n for n, _ in refs]
        entry = dict(sam=sam)
        try:
            alns = K.parse_bam(path)
        except Exception as e:  # the reference raised: record the exception type
            entry["raises"] = type(e).__name__
            try:
                for cid in ko.contig_order(batch):
                    ko.parse_records(batch, cid)
                raise AssertionError(name + ": oracle did not raise")
            except (KeyError, IndexError, RuntimeError) as oe:
                assert type(oe).__name__ == entry["raises"], (name, oe, entry["raises"])
            out[name] = entry
            os.unlink(path)
            continue
        assert not name.startswith("ERR_"), name + " expected to raise"
        contigs = []
        for ref_id, aln in alns.items():
            rec, tabs = aln_record(aln)
            W, S, E, cs, ce, de, il = tabs
            rec.update(name=ref_id, weights=W.tolist(), clip_start_weights=S.tolist(),
                       clip_end_weights=E.tolist(), clip_starts=cs.tolist(), clip_ends=ce.tolist(),
                       deletions=de.tolist(), insertions=[[p, s, c] for p, s, c in il],
                       consensus_depth=[int(x) for x in aln.consensus_depth],
                       clip_start_depth=[int(x) for x in aln.clip_start_depth],
                       clip_end_depth=[int(x) for x in aln.clip_end_depth])
            oa = ko.parse_records(batch, names.index(ref_id))
            check_oracle(name + ":" + ref_id, oa, tabs)
            d = oa.derived()
            assert d["consensus_depth"].tolist() == rec["consensus_depth"], name
            assert d["clip_start_depth"].tolist() == rec["clip_start_depth"], name
            assert d["clip_end_depth"].tolist() == rec["clip_end_depth"], name
            rec["runs"] = []
            for md, trim, upper in quirk_cases.OPTION_SETS:
                s, ch = K.consensus_sequence(aln.weights, aln.insertions, aln.deletions, None,
                                             trim, md, upper)
                assert oa.consensus_sequence(None, trim, md, upper) == (s, ch), (name, md, trim, upper)
                rec["runs"].append(dict(min_depth=md, trim_ends=trim, uppercase=upper,
                                        consensus=s, changes=changes_str(ch)))
            contigs.append(rec)
        res = K.bam_to_consensus(path)
        entry["contigs"] = contigs
        entry["names"] = [c.name for c in res.consensuses]
        entry["reports"] = [r.replace(path, "{bam_path}") for r in res.refs_reports.values()]
        out[name] = entry
        os.unlink(path)
    return out


def do_long_quirks(K):
    """oracle/quirk_cases.py: long_cases() (reads with > 16 CIGAR words) through the unmodified reference: digests of every
    table, the insertion dicts, consensus and change codes -- and the oracle asserted equal on all of them."""
    out = {}
    for name, (sam, exc) in quirk_cases.long_cases().items():
        with tempfile.NamedTemporaryFile("w", suffix=".sam", delete=False) as fh:
            fh.write(sam)
            path = fh.name
        _, refs, recs = samio_py.read_alignment_file(path)
        batch = samio_py.records_to_batch(refs, recs)
        names = [n for n, _ in refs]
        entry = {}
        try:
            alns = K.parse_bam(path)
        except Exception as e:
            entry["raises"] = type(e).__name__
            assert exc is not None and type(e) is exc, (name, e)
            try:
                for cid in ko.contig_order(batch):
                    ko.parse_records(batch, cid)
                raise AssertionError(name + ": oracle did not raise")
            except (KeyError, IndexError, RuntimeError) as oe:
                assert type(oe).__name__ == entry["raises"], (name, oe, entry["raises"])
            out[name] = entry
            os.unlink(path)
            continue
        assert exc is None, name + " expected to raise"
        contigs = []
        for ref_id, aln in alns.items():
            rec, tabs = aln_record(aln)
            oa = ko.parse_records(batch, names.index(ref_id))
            check_oracle(name + ":" + ref_id, oa, tabs)
            s, ch = K.consensus_sequence(aln.weights, aln.insertions, aln.deletions, None, False, 1, False)
            assert oa.consensus_sequence(None, False, 1, False) == (s, ch), name
            rec.update(name=ref_id, consensus=s, changes=changes_str(ch))
            contigs.append(rec)
        entry["contigs"] = contigs
        out[name] = entry
        os.unlink(path)
    return out


def do_patch_cases(K):
    """consensus_sequence with cdr_patches (kindel.py:393-401) on a tiny table, incl. overlaps."""
    sam = quirk_cases.CASES["clip_both_ends_pair"]
    with tempfile.NamedTemporaryFile("w", suffix=".sam", delete=False) as fh:
        fh.write(sam)
        path = fh.name
    aln = list(K.parse_bam(path).values())[0]
    _, refs, recs = samio_py.read_alignment_file(path)
    oa = ko.parse_records(samio_py.records_to_batch(refs, recs), 0)
    os.unlink(path)
    R = K.Region
    sets = {
        "single": [R(12, 15, "ACGTT", None)],
        "span1": [R(12, 13, "GG", None)],
        "none_seq_elsewhere": [R(5, 9, None, None), R(12, 15, "AC", None)],
        "overlapping_second_ignored": [R(10, 14, "TTTT", None), R(12, 16, "GGGG", None)],
        "empty_seq_not_applied": [R(12, 15, "", None)],
        "at_zero_and_end": [R(0, 3, "CCC", None), R(27, 30, "AAA", None)],
    }
    out = {"sam": sam, "sets": {}}
    for k, patches in sets.items():
        s, ch = K.consensus_sequence(aln.weights, aln.insertions, aln.deletions, patches, False, 1, False)
        assert oa.consensus_sequence(patches, False, 1, False) == (s, ch), k
        out["sets"][k] = dict(patches=[[p.start, p.end, p.seq] for p in patches],
                              consensus=s, changes=changes_str(ch))
    return out


def do_weights(K):
    for rel in ("data_bwa_mem/1.1.sub_test.bam", "data_minimap2/1.1.multi.bam"):
        path = os.path.join(REF_ROOT, "tests", rel)
        for relative in (False, True):
            df = K.weights(path, relative=relative)
            cols = {c: df[c].to_numpy() for c in df.columns}
            cols["chrom"] = cols["chrom"].astype(str)
            np.savez_compressed(os.path.join(GOLD, "weights_%s_%s.npz" % (
                fixture_key(rel), "rel" if relative else "abs")), columns=np.asarray(list(df.columns)), **cols)


def do_features(K):
    """features() (kindel.py:633-664) of the reference on single-contig fixtures (it crashes on multi-contig input)."""
    for rel in ("data_bwa_mem/1.1.sub_test.bam", "data_ext/3.issue23.bc75.sam", "data_ext/1.issue23.debug.sam"):
        path = os.path.join(REF_ROOT, "tests", rel)
        df = K.features(path)
        cols = {c: df[c].to_numpy() for c in df.columns}
        cols["chrom"] = cols["chrom"].astype(str)
        np.savez_compressed(os.path.join(GOLD, "features_%s.npz" % fixture_key(rel)), columns=np.asarray(list(df.columns)), **cols)


def do_fasta():
    """The FASTA files the reference's own tests compare against (tests/test_kindel.py:114-124,143-158,181-238 and the
    -r variants): record name -> sequence, per file."""
    out = {}
    for rel in FIXTURES:
        for suffix, tag in ((".fa", "default"), (".realign.fa", "realign")):
            path = os.path.join(REF_ROOT, "tests", rel.rsplit(".", 1)[0] + suffix)
            if not os.path.exists(path):
                continue
            recs, name = {}, None
            for line in open(path):
                line = line.rstrip("\n")
                if line.startswith(">"):
                    name = line[1:].split()[0]
                    recs[name] = ""
                elif name is not None:
                    recs[name] += line
            out.setdefault(fixture_key(rel), {})[tag] = recs
    with open(os.path.join(GOLD, "reference_fasta.json"), "w") as fh:
        json.dump(out, fh, indent=0, sort_keys=True)
    print("reference FASTA goldens:", sum(len(v) for v in out.values()), "files")


def main():
    only = set(sys.argv[1:])   # e.g. `python -m oracle.make_golden quirks features fasta` regenerates just those
    K = load_reference()
    os.makedirs(GOLD, exist_ok=True)
    if not only or "quirks" in only:
        quirks = do_quirks(K)
        quirks["__patches__"] = do_patch_cases(K)
        with open(os.path.join(GOLD, "quirks.json"), "w") as fh:
            json.dump(quirks, fh, indent=0, sort_keys=True)
        print("quirk cases:", len(quirks), "raising:", sum("raises" in v for v in quirks.values()))
    if not only or "fixtures" in only:
        fx = do_fixtures(K)
        with open(os.path.join(GOLD, "reference_outputs.json"), "w") as fh:
            json.dump(fx, fh, indent=0, sort_keys=True)
    if not only or "long" in only:
        lq = do_long_quirks(K)
        with open(os.path.join(GOLD, "long_quirks.json"), "w") as fh:
            json.dump(lq, fh, indent=0, sort_keys=True)
        print("long-read cases:", len(lq), "raising:", sum("raises" in v for v in lq.values()))
    if not only or "weights" in only:
        do_weights(K)
    if not only or "features" in only:
        do_features(K)
    if not only or "fasta" in only:
        do_fasta()
    print("done")


if __name__ == "__main__":
    sys.exit(main())
