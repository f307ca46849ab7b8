#!/usr/bin/env python
"""CAMPAIGN (round 6): batches of valid long reads with ONE read rebuilt without any validity constraint (tests/test_fuzz.py:
_one_wild_long_read) -- 17 to 1 500 CIGAR ops, i.e. up to two dozen of k_prep_long's 64-op tiles, and reads with one op of 2^23 bases or
more -- through the engine and the oracle: the same tables, or the same exception BY TYPE.  `--gpu`: the HIP library (default: the emulator).

    python scripts/exp/wild_long_campaign.py [--gpu] [--scale 1.0]
"""
import argparse, os, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as g  # noqa: E402
from kindel_amd import _native as N  # noqa: E402
from tests import test_fuzz as T  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gpu", action="store_true")
ap.add_argument("--scale", type=float, default=1.0)
a = ap.parse_args()
lib = N.Library(g.build_hip()) if a.gpu else N.Library(g.build_emu())
n = lambda k: max(1, int(k * a.scale))
total = {}
for lo, cnt, ops, lens in ((20000, 1200, (65, 400), (6000, 2500)), (30000, 600, (17, 64), (6000, 2500)), (40000, 300, (400, 1500), (30000, 14000))):
    out = T._wild_long_campaign(lib, range(lo, lo + n(cnt)), ops, [N.KD_MODE_AUTO], contig_lens=lens)
    print("ops", ops, out, flush=True)
    for k, v in out.items():
        total[k] = total.get(k, 0) + v
for k, (op, length) in enumerate([(0, 1 << 23), (2, 1 << 23), (1, 1 << 23), (4, (1 << 23) + 5), (3, 1 << 24), (0, (1 << 28) - 1), (2, (1 << 23) - 1), (7, 1 << 25), (8, 1 << 23), (4, 1 << 27)]):
    out = T._wild_long_campaign(lib, range(50000 + 100 * k, 50000 + 100 * k + n(60)), (65, 200), [N.KD_MODE_AUTO], big=(op, length))
    print("one op %d of %d bases" % (op, length), out, flush=True)
    for kk, v in out.items():
        total[kk] = total.get(kk, 0) + v
print("total", total, "-- no disagreement with the oracle (a disagreement raises)")
