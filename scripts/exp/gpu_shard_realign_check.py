"""EXPERIMENT RECORD (round 5): tests/shard_fuzz.py's --realign campaign with the HIP library instead of the emulator -- two ranks
sharing the one GPU over gloo, files built to have clip-dominant regions, against the single-process run on the same GPU.
`python scripts/exp/gpu_shard_realign_check.py N SEED0` (gpurun; not part of any test run)."""
import os, sys, logging, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
logging.disable(logging.WARNING)
if __name__ == "__main__":
    from kindel_amd import _native as N
    from tests import shard_fuzz
    t0 = time.time()
    n, seed0 = int(sys.argv[1]), int(sys.argv[2])
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    files, diffs = shard_fuzz.run_campaign(n, seed0, world, N.default_library().path, kw=dict(realign=True, min_overlap=7), structured=True)
    for d in diffs:
        print("DIFF", d)
    print("gpu shard realign: %d files, %d ranks on one GPU, diffs %d, %.0f s" % (len(files), world, len(diffs), time.time() - t0), flush=True)
