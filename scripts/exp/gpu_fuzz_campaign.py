"""LOCAL GPU campaign (round 5): the engine against the oracle on seeds the suites do not hold -- mixed shapes at depth (what the
emulator cannot show: 20 wavefronts per CU racing on k_window's lists), long CIGARs of several tiles, wild reads.
  python scripts/exp/gpu_fuzz_campaign.py SECONDS"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from kindel_amd import _native as N
from tests import fuzz
lib = N.default_library()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
t0 = time.time(); bad = 0; n = {"mixed": 0, "long": 0, "wild": 0, "valid": 0}
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100000      # (round 6: another range under KD_GUARD=1)
while time.time() - t0 < budget:
    seed += 1
    rng = np.random.default_rng(seed)
    kind = ["mixed", "long", "wild", "valid", "mixed"][seed % 5]
    try:
        if kind == "mixed":
            L, nr = [(700, 6000), (4000, 30000), (60000, 20000), (9000, 3000), (300000, 40000)][seed % 5 if seed % 5 < 5 else 0]
            b = fuzz.mixed_batch(rng, L, nr, piled=bool(seed & 1))
            r = fuzz.check_engine(lib, b, N.KD_MODE_AUTO, window=[0, 0, 64, 256, 448, 1024, 2048][seed % 7], slice_reads=[0, 0, 64, 300, 2000][seed % 5])
        elif kind == "long":
            b = fuzz.random_batch(rng, 150, contig_lens=(30000, 14000), wild=0.0, sort=bool(seed & 1), long_ops=(17, 2600))
            r = fuzz.check_engine(lib, b, [N.KD_MODE_AUTO, N.KD_MODE_GLOBAL][seed % 2], window=[64, 256, 640][seed % 3], slice_reads=[0, 16][seed % 2])
        elif kind == "wild":
            b = fuzz.random_batch(rng, 40, wild=0.25, sort=bool(seed & 1))
            r = fuzz.check_engine(lib, b, [N.KD_MODE_AUTO, N.KD_MODE_GLOBAL, N.KD_MODE_COOP, N.KD_MODE_STRIP][seed % 4], window=64, slice_reads=[0, 16][seed % 2])
        else:
            b = fuzz.random_batch(rng, 3000, contig_lens=(5000, 3000, 800), wild=0.0, sort=bool(seed & 1))
            r = fuzz.check_engine(lib, b, [N.KD_MODE_AUTO, N.KD_MODE_COOP, N.KD_MODE_STRIP][seed % 3], window=[64, 256, 640][seed % 3], slice_reads=[0, 16][seed % 2])
        n[kind] += 1
        if r not in ("ok", "raise"):
            bad += 1; print("BAD", kind, seed, r, flush=True)
    except Exception as e:
        bad += 1; print("FAIL", kind, seed, type(e).__name__, str(e)[:200], flush=True)
print("gpu fuzz campaign:", n, "batches in %.0f s, failures: %d" % (time.time() - t0, bad), flush=True)
