"""EXPERIMENT RECORD / TEST TOOL (round 5), run through scripts/exp/asan_emu.sh: `bam_to_consensus` (default and realign) of every
mutated file of a directory (scripts/exp/bam_mutations.py) on the sanitizer build of the emulator -- whatever the decoder lets through
must leave the engine with a result or one of the reference's exception types, never with a memory error.  1 500 files x 2:
744 results, 1630 OSError, 536 KeyError, 84 IndexError, 6 RuntimeError, no report."""
import sys, os, glob, logging, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
logging.disable(logging.WARNING)
from kindel_amd import _native as N
N._default = N.Library(os.environ.get("EMULIB", '/tmp/libkindel_emu_asan.so'))
from kindel_amd import kindel as K
out = collections.Counter()
for p in sorted(glob.glob(sys.argv[1] + "/m*.bam")):
    for kw in (dict(), dict(realign=True)):
        try:
            K.bam_to_consensus(p, **kw); out["ok"] += 1
        except Exception as e:
            out[type(e).__name__] += 1
print(dict(out))
