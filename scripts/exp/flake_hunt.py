#!/usr/bin/env python
"""Hunt for an intermittent mismatch: tests/test_gpu_parity.py::test_mostly_clipped_reads_and_deep_sites failed ONCE in the plain suite
(577 passed fenced, in the same gpurun call).  The test's batch, every mode, N times; every difference is named (table channel, site,
engine value, oracle value; insertion dicts; consensus).   python scripts/exp/flake_hunt.py [loops]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kindel_amd import _native as N      # noqa: E402
from oracle import oracle as ko          # noqa: E402
from tests import parity as P            # noqa: E402
from tools import synth                  # noqa: E402

loops = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dirty = len(sys.argv) > 2 and sys.argv[2] == "dirty"
if len(sys.argv) > 2 and sys.argv[2] == "aged":      # the process first runs the suite's tests in front of the one that failed (a long-lived process: thousands of contexts)
    import pytest
    rc = pytest.main(["-m", "gpu", "-q", "-x", "-p", "no:faulthandler", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_gpu_parity.py"),
                      "-k", "quirk_case or reference_fixture or window_tunings or unsorted_input or known_answers or fetch_all or multi_tile"])
    print("aged by pytest.main: rc", rc, flush=True)      # device memory dirtied before every run (hipMalloc hands recycled pages out as they are)
lib = N.default_library()
batch = synth.to_numpy(synth.short_reads([2500, 1200], 2500, seed=13, clip_p=0.6, indel_p=0.3))
oas = {cid: ko.parse_records(batch, cid) for cid in ko.contig_order(batch)}
want = {}
for cid, oa in oas.items():
    L = oa.L
    t = np.zeros((N.KD_NCH, L + 1), np.uint32)
    t[0:5, :L] = oa.weights.T; t[5] = oa.deletions; t[6:11, :L] = oa.clip_start_weights.T; t[11:16, :L] = oa.clip_end_weights.T
    t[16] = oa.clip_starts; t[17] = oa.clip_ends; t[18] = oa.ins_totals
    want[cid] = (t, sorted(oa.insertions), oa.consensus_sequence(min_depth=1), oa.depth_minmax())
bad = 0
t0 = time.time()
for it in range(loops):
    for mode in (N.KD_MODE_AUTO, N.KD_MODE_COOP, N.KD_MODE_STRIP, N.KD_MODE_GLOBAL):
        if dirty:
            import torch
            pat = [0xA5, 0xFF, 0x01, 0x80, 0x7F][(it + mode) % 5]
            junk = [torch.full(((8 << 20) * (1 + (k + it) % 5),), pat, dtype=torch.uint8, device="cuda") for k in range(12)]
            if it % 3 == 0:
                junk.append(torch.randint(0, 256, (64 << 20,), dtype=torch.uint8, device="cuda"))
            torch.cuda.synchronize()
            del junk
            torch.cuda.empty_cache()
        run = P.Run(lib, batch, mode=mode)
        for cid in run.order:
            t, ins, (oseq, och), mm = want[cid]
            got = run.tables[cid]
            if not np.array_equal(got, t):
                bad += 1
                ch, site = np.nonzero(got != t)
                print("iter %d mode %d contig %d: %d table cells differ; first: %s" % (it, mode, cid, len(ch),
                      [(int(c), int(s), int(got[c, s]), int(t[c, s])) for c, s in list(zip(ch, site))[:12]]), flush=True)
                print("   channels hit: %s  site range %d..%d" % (sorted(set(int(c) for c in ch)), int(site.min()), int(site.max())), flush=True)
            if sorted(run.ins[cid]) != ins:
                bad += 1
                a, b = set(run.ins[cid]), set(ins)
                print("iter %d mode %d contig %d: insertion dicts differ: engine-only %s oracle-only %s" % (it, mode, cid, sorted(a - b)[:6], sorted(b - a)[:6]), flush=True)
            seq, chg, gmm, _ = run.cns[cid]
            if seq.decode() != oseq or [None if c == 0 else chr(c) for c in chg] != och or gmm != mm:
                bad += 1
                print("iter %d mode %d contig %d: consensus / changes / depth range differ (%s vs %s)" % (it, mode, cid, gmm, mm), flush=True)
print("flake hunt: %d iterations x 4 modes in %.0f s: %d differences" % (loops, time.time() - t0, bad))
