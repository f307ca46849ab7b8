"""TEST / MEASUREMENT INFRASTRUCTURE ONLY -- time the UNMODIFIED reference (kindel.kindel.parse_records +
consensus_sequence, /root/reference/kindel/kindel.py:21-128, :384-430) on a sample of an SoA read batch.

Used by bench.py's `cpu_baseline` leg when /root/reference is present (the build container) and by
`python -m oracle.refbaseline` to write profiles/reference_python_baseline.json, which bench.py quotes -- with its
provenance -- where the reference tree does not exist (the GPU box).  The reference is single-threaded: cores = 1.
"""
import json
import os
import sys
import time

import numpy as np

from oracle import oracle as ko
from oracle.refrun import load_reference, reference_available

_NIB = np.frombuffer(b"=ACMGRSVTWYHKDBN", np.uint8)
_OPS = "MIDNSHP=X"


class _Rec:
    __slots__ = ("pos", "mapped", "seq", "cigars")

    def __init__(self, pos, mapped, seq, cigars):
        self.pos, self.mapped, self.seq, self.cigars = pos, mapped, seq, cigars


def sample_prefix(batch, contig_id, max_events):
    """Reads of `contig_id` that lie wholly inside its first Ls sites, Ls chosen so that about max_events aligned bases
    are kept: the same workload at full depth over a prefix of the contig.  -> (sub-batch, Ls)"""
    c = np.asarray(batch["contig"])
    pos = np.asarray(batch["pos0"]).astype(np.int64)
    sl = np.asarray(batch["seq_len"]).astype(np.int64)
    L = int(batch["contig_lens"][contig_id])
    mine = c == contig_id
    total = int(sl[mine].sum())
    Ls = L if total <= max_events else max(2000, int(L * max_events / max(total, 1)))
    # end of the read's footprint <= start + query length + deletions; deletions are bounded by the CIGAR, use a margin
    keep = mine & (pos >= 0) & (pos + sl + 64 <= Ls)
    sub = dict(batch)
    for k in ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig"):
        sub[k] = np.ascontiguousarray(np.asarray(batch[k])[keep])
    lens = np.asarray(batch["contig_lens"], np.uint32).copy()
    lens[contig_id] = Ls
    sub["contig_lens"] = lens
    return sub, Ls


def to_records(batch):
    """SoA batch -> the simplesam-like records the reference's parse_records consumes (kindel.py:42-47)."""
    seq4 = np.asarray(batch["seq4"])
    asc = np.empty(2 * len(seq4), np.uint8)
    asc[0::2] = _NIB[seq4 >> 4]
    asc[1::2] = _NIB[seq4 & 15]
    text = asc.tobytes().decode("ascii")
    cg = np.asarray(batch["cigar"])
    recs = []
    for p, fl, so, sl, co, nc in zip(batch["pos0"].tolist(), batch["flag"].tolist(), batch["seq_off"].tolist(),
                                     batch["seq_len"].tolist(), batch["cig_off"].tolist(), batch["n_cig"].tolist()):
        ops = cg[co:co + nc].tolist()
        recs.append(_Rec(p + 1, not (fl & 4), text[2 * so: 2 * so + sl] if sl else "*",
                         tuple((w >> 4, _OPS[w & 15] if (w & 15) < 9 else "?") for w in ops)))
    return recs


def time_reference(batch, contig_id=0, max_events=4.0e7):
    """-> dict(value events/s, events, seconds, sites, ...) for the reference's two loops on a prefix sample; also
    checks that the oracle (C restatement) returns the same consensus on that sample."""
    K = load_reference()
    sub, Ls = sample_prefix(batch, contig_id, max_events)
    recs = to_records(sub)
    t0 = time.perf_counter()
    aln = K.parse_records("ref", Ls, recs)
    t1 = time.perf_counter()
    seq, changes = K.consensus_sequence(aln.weights, aln.insertions, aln.deletions, None, False, 1, False)
    t2 = time.perf_counter()
    oa = ko.parse_records(sub, contig_id)
    oseq, och = oa.consensus_sequence()
    events = int(oa.n_events_aligned)
    return dict(value=events / (t2 - t0), unit="events/s", cores=1, host_cores=os.cpu_count(), kind="reference",
                events=events, reads=len(recs), sites=Ls, seconds=round(t2 - t0, 2),
                parse_records_seconds=round(t1 - t0, 2), consensus_sequence_seconds=round(t2 - t1, 2),
                same_consensus_as_oracle=bool(seq == oseq and changes == och),
                sample="reads wholly inside the first %d sites of contig %d at full depth: %d reads, %d aligned-base events; "
                       "unmodified kindel.kindel.parse_records + consensus_sequence on 1 core" % (Ls, contig_id, len(recs), events))


def main():
    """python -m oracle.refbaseline [config] -> profiles/reference_python_baseline.json (needs /root/reference)"""
    if not reference_available():
        sys.exit("needs /root/reference")
    from tools import synth
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
    c = dict(synth.CONFIGS[cfg])
    # the sample is a prefix of the contig at full depth: generate just that much of the workload
    c["contig_lens"] = [200_000] if cfg == "C3" else list(c["contig_lens"])
    batch = synth.to_numpy(synth.make(c))
    out = time_reference(batch, 0)
    out["workload"] = "%s error model and depth (synthetic, seed as in tools/synth.py), contig shortened to %d sites for generation" % (
        cfg, int(c["contig_lens"][0]))
    out["where"] = "build container (%d host cores), %s" % (os.cpu_count(), time.strftime("%Y-%m-%d"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "reference_python_baseline.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
