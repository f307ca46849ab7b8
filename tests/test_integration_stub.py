"""INTEGRATION.md's stub, executed: the UNMODIFIED reference module (oracle/refrun.py) with its parse_bam / consensus_sequence
replaced by tests/integration_stub.py's ctypes versions over the C-ABI library runs ITS OWN bam_to_consensus() and reproduces
the golden FASTA the reference produced on its own (tests/golden) -- through the emulator library here, through
libkindel_hip.so on the GPU."""
import importlib

import pytest

from tools import synth
from oracle import refrun
from tests import parity as P
from tests.integration_stub import Stub

GOLD = P.golden_outputs()
pytestmark = pytest.mark.skipif(not refrun.reference_available(), reason="needs the reference (source tree or oracle/_ref bytecode)")


def _run(lib_path, tmp_path, key):
    K = refrun.load_reference()
    saved = (K.parse_bam, K.consensus_sequence)
    try:
        Stub(lib_path).patch(K)
        path = str(tmp_path / (key + ".bam"))
        synth.write_bam(path, P.load_fixture(key), sort_order="unknown")
        res = K.bam_to_consensus(path)        # the reference's own orchestration, report text and record naming
        gold = GOLD[key]["contigs"]
        assert [(c.name, c.sequence) for c in res.consensuses] == [(g["name"] + "_cns", g["consensus"]) for g in gold]
        for g in gold:
            assert res.refs_reports[g["name"]] == g["report"].replace("{bam_path}", path)
            assert "".join("." if c is None else c for c in res.refs_changes[g["name"]]) == g["changes"]
    finally:
        K.parse_bam, K.consensus_sequence = saved


@pytest.mark.parametrize("key", ["bwa_mem__1.1.sub_test", "minimap2__1.1.multi", "ext__2.issue23.bc63"])
def test_reference_module_through_the_stub_on_the_emulator(emu_lib, tmp_path, key):
    _run(emu_lib.path, tmp_path, key)


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["bwa_mem__1.1.sub_test", "minimap2__1.1.multi", "segemehl__3.1.sub_test"])
def test_reference_module_through_the_stub_on_the_gpu(hip_lib, tmp_path, key):
    _run(hip_lib.path, tmp_path, key)
