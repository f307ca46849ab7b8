#!/usr/bin/env python
"""Tables that do not fit the GPU are a clean MemoryError (KD_E_NOMEM from hipMalloc), and the process goes on: a context over 4.28 G sites
(325 GB = 303 GiB of tables) pushed a handful of reads, then a small run against the oracle on the same device.
    timeout 300 python scripts/exp/gpu_nomem_check.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kindel_amd import _native as N      # noqa: E402
from tests import parity as P            # noqa: E402
from tools import synth                  # noqa: E402


def main():
    small = synth.to_numpy(synth.short_reads([4000], 10, seed=1))
    big = dict(small)
    big["contig_lens"] = np.asarray([1_070_000_000] * 4, np.uint32)
    eng = N.Engine(big["contig_lens"], device=0)
    try:
        eng.push(big)
        print("UNEXPECTED: 303 GiB of tables were allocated")
        return 1
    except MemoryError as e:
        print("MemoryError as expected:", str(e)[:160])
    finally:
        eng.close()
    P.assert_matches_oracle(P.Run(N.default_library(), small), what="after the failed allocation")
    print("a small run after it: equal to the oracle")
    return 0


if __name__ == "__main__":
    sys.exit(main())
