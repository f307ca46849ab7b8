"""EXPERIMENT RECORD (round 5), run through scripts/exp/asan_emu.sh: single-process --realign over tests/shard_fuzz structured files SEED_A..SEED_B on the
emulator library $EMULIB, one digest per file (runs under different garbage fills must print the same digests)."""
import sys, os, tempfile, logging, random, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
logging.disable(logging.WARNING)
from kindel_amd import _native as N
N._default = N.Library(os.environ.get("EMULIB", '/tmp/libkindel_emu_asan.so'))
from kindel_amd import kindel as K
from tests import shard_fuzz
a, b = int(sys.argv[1]), int(sys.argv[2])
tmp = tempfile.mkdtemp()
files = shard_fuzz.make_files(b - a, a, tmp, structured=True)
for i, p in enumerate(files):
    try:
        r = K.bam_to_consensus(p, realign=True, min_overlap=7)
        h = hashlib.sha256(repr(([(c.name, c.sequence) for c in r.consensuses], {k: list(v) for k, v in r.refs_changes.items()}, sorted(r.refs_reports.items()))).replace(tmp, "").encode()).hexdigest()[:12]
    except Exception as e:
        h = "raise " + type(e).__name__
    print("file", a + i, h, flush=True)
print("done")
