#!/bin/bash
# Round 3: the device-side ingest on the GPU: its tests, then end to end (BAM path -> FASTA) against the host decoder, C3 at 0.1 and 0.5 x depth
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=$PWD/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_inflate_proto.py -m gpu -q -x -p no:cacheprovider > $O/pytest_ingest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_ingest.log
for spec in "0.1 phred" "0.1 absent" "0.5 phred"; do set -- $spec
  timeout 600 python scripts/e2e_bench.py --scale $1 --qual $2 --repeat 2 --out $O/r3y_e2e_$1_$2.json > /dev/null 2> $O/r3y_e2e_$1_$2.err || tail -3 $O/r3y_e2e_$1_$2.err
  python - <<PY
import json
try:
    d=json.load(open("$O/r3y_e2e_$1_$2.json"))
    print("$1 $2: bam %.0f MB  whole %.3f s  streamed %.3f s (decode %.3f)  gpu-ingest %.3f s %s" % (d["bam_bytes"]/1e6, d["whole_file"]["total_s"], d["streamed"]["total_s"], d["streamed"].get("ingest_decode_s",0), d["gpu_ingest"]["total_s"], d["gpu_ingest"]))
except Exception as e: print("$1 $2 failed", e)
PY
done
