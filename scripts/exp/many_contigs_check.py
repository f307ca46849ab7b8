#!/usr/bin/env python
"""More contigs than any 16-bit field or 65 536-wide launch dimension could hold (a fragmented assembly, a metagenome): N contigs of a
few hundred sites each through kd_step and the per-contig read-outs, a sample of contigs (the first, the last, the ones around 65 535 /
65 536 and a random set) against the oracle -- tables, insertion dicts, consensus, change codes, depth range.
    python scripts/exp/many_contigs_check.py [n_contigs] [emu]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kindel_amd import _native as N      # noqa: E402
from oracle import oracle as ko          # noqa: E402
from tools import synth                  # noqa: E402
import __graft_entry__ as g              # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 70001
    emu = len(sys.argv) > 2 and sys.argv[2] == "emu"
    lib = N.Library(g.build_emu()) if emu else None
    rng = np.random.default_rng(5)
    lens = rng.integers(160, 400, n).astype(np.uint32)
    t0 = time.time()
    tb = synth.short_reads(lens, 4, read_len=100, seed=9, device="cpu" if emu else "cuda:0")
    host = synth.to_numpy(tb)
    print("contigs %d, reads %d, generated in %.1f s" % (n, len(host["contig"]), time.time() - t0), flush=True)
    eng = N.Engine(lens, device=0, lib=lib) if emu else N.Engine(lens, device=0)
    out = np.zeros(int(lens.sum()) * 2 + 4096, np.uint8)
    t0 = time.time()
    if emu:
        eng.push(host)
        off = eng.finish(out)
    else:
        off = eng.step_device(synth.device_ptrs(tb), len(host["contig"]), tb["seq4_bytes"], tb["cigar_words"], out)
    print("step %.2f s, consensus bytes %d" % (time.time() - t0, int(off[-1])), flush=True)
    pick = sorted(set([0, 1, n - 1, n - 2] + [c for c in (65534, 65535, 65536, 65537, 32767, 32768) if c < n] + [int(x) for x in rng.integers(0, n, 300)]))
    bad = 0
    for cid in pick:
        oa = ko.parse_records(host, cid)
        L = oa.L
        t = eng.tables(cid)
        ok = (np.array_equal(t[0:5, :L].T, oa.weights) and np.array_equal(t[5], oa.deletions) and np.array_equal(t[6:11, :L].T, oa.clip_start_weights) and
              np.array_equal(t[11:16, :L].T, oa.clip_end_weights) and np.array_equal(t[16], oa.clip_starts) and np.array_equal(t[17], oa.clip_ends) and
              np.array_equal(t[18], oa.ins_totals))
        site, count, strings = eng.insertions(cid)
        ok = ok and sorted((int(p), s, int(c)) for p, c, s in zip(site, count, strings)) == sorted(oa.insertions)
        oseq, och = oa.consensus_sequence(min_depth=1)
        ok = ok and out[int(off[cid]): int(off[cid + 1])].tobytes().decode() == oseq
        seq, ch, mm, _ = eng.consensus_fetch(cid)
        ok = ok and seq.decode() == oseq and [None if c == 0 else chr(c) for c in ch] == och and mm == oa.depth_minmax()
        if not ok:
            bad += 1
            print("MISMATCH contig", cid, flush=True)
    eng.close()
    print("checked %d of %d contigs: %s" % (len(pick), n, "all equal to the oracle" if not bad else "%d MISMATCHES" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
