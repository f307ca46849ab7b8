"""The unmodified reference against kindel_amd on random SAM files, end to end (tests/reference_fuzz.py): default and --realign."""
import pytest

from oracle import refrun
from tests import reference_fuzz as RF

pytestmark = pytest.mark.skipif(not refrun.reference_available(), reason="the reference (or its bytecode, oracle/make_ref.py) is not on this box")


def test_random_files_through_reference_and_emulated_engine(api_on_emu):
    from kindel_amd import kindel as K
    R = refrun.load_reference()
    diffs = [d for d in (RF.check_seed(R, K, seed) for seed in range(400, 430)) if d]
    assert not diffs, diffs


def test_wild_files_same_result_or_same_exception(api_on_emu):
    """6 000 such files ran clean as a local campaign (round 5); these seeds stay."""
    from kindel_amd import kindel as K
    R = refrun.load_reference()
    diffs = [d for d in (RF.check_wild_seed(R, K, seed) for seed in range(700, 760)) if d]
    assert not diffs, diffs


def test_weights_features_parse_bam_against_the_reference(api_on_emu):
    """weights() incl. its float columns (no value assertions in the reference's own tests: pinned here by running it side by side),
    features(), the alignment namedtuple of parse_bam() -- on random files."""
    import warnings
    from kindel_amd import kindel as K
    R = refrun.load_reference()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        diffs = [d for d in (RF.check_api_seed(R, K, seed) for seed in range(600, 612)) if d]
    assert not diffs, diffs


@pytest.mark.gpu
def test_random_files_through_reference_and_the_gpu(hip_lib):
    from kindel_amd import kindel as K
    R = refrun.load_reference()
    diffs = [d for d in (RF.check_seed(R, K, seed) for seed in range(500, 540)) if d]
    assert not diffs, diffs
