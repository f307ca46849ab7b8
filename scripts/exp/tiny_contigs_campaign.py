#!/usr/bin/env python
"""CAMPAIGN (round 6): hundreds to thousands of TINY contigs (1 - 150 sites: every contig a few 64-site segments of G-space, contig
boundaries inside every consensus tile, most reads hanging over an end or exactly filling their contig) against the oracle --
tables, insertion dicts, consensus, change codes, depth ranges, or an exception where the oracle raises.
    python scripts/exp/tiny_contigs_campaign.py SECONDS [SEED] [emu]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kindel_amd import _native as N      # noqa: E402
from tests import fuzz                   # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 900000
if len(sys.argv) > 3 and sys.argv[3] == "emu":
    import __graft_entry__ as g
    lib = N.Library(g.build_emu())
else:
    lib = N.default_library()
t0 = time.time()
n = {"ok": 0, "raise": 0}
bad = 0
while time.time() - t0 < budget:
    seed += 1
    rng = np.random.default_rng(seed)
    n_contigs = int([40, 300, 1200, 3000][seed % 4])
    lens = tuple(int(x) for x in rng.integers(1, [150, 40, 90, 64][seed % 4] + 1, n_contigs))
    b = fuzz.random_batch(rng, int(n_contigs * [6, 3, 2, 1][seed % 4]), contig_lens=lens, wild=[0.0, 0.0, 0.002, 0.0][seed % 4], sort=bool(seed & 1))
    try:
        r = fuzz.check_engine(lib, b, [N.KD_MODE_AUTO, N.KD_MODE_GLOBAL, N.KD_MODE_COOP, N.KD_MODE_STRIP][(seed // 4) % 4],
                              window=[0, 64, 256, 448][(seed // 16) % 4], slice_reads=[0, 16][(seed // 64) % 2])
        n[r] += 1
    except AssertionError as e:
        bad += 1
        print("FAIL seed", seed, str(e)[:200], flush=True)
print("tiny contigs campaign: %d batches in %.0f s %s, %d failures (last seed %d)" % (sum(n.values()), time.time() - t0, n, bad, seed))
sys.exit(1 if bad else 0)
