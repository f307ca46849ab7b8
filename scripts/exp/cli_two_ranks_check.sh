#!/bin/bash
# the CLI with two ranks sharing the one GPU over gloo (host tensors in the collective, the engine's row in HBM) against the
# single-process run: the short form of tests/test_gpu_parity.py::test_cli_two_ranks_on_one_gpu
cd "${GRAFT_REPO_ROOT:-/root/repo}"; B=oracle/_ref/fixtures/data_bwa_mem/3.1.sub_test.bam
export KINDEL_DIST_BACKEND=gloo PYTHONPATH=$PWD
python -m kindel_amd consensus $B > /tmp/one.fa 2> /tmp/one.err &
python -m kindel_amd consensus --gpus 2 $B > /tmp/two.fa 2> /tmp/two.err; rc2=$?
wait; echo "two-rank rc=$rc2 single $(sha256sum < /tmp/one.fa | cut -c1-12) two $(sha256sum < /tmp/two.fa | cut -c1-12) bytes $(wc -c < /tmp/two.fa)"; tail -3 /tmp/two.err | cut -c1-300
