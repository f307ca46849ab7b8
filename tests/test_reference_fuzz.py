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


def test_files_with_clip_dominant_regions_through_realign(api_on_emu):
    """reference_fuzz.structured_sam: novel segments in the sample, reads soft-clipped at the junctions -- files on which --realign
    patches something (the random files above rarely do).  Default and two realign settings, reference vs kindel_amd."""
    import logging
    from kindel_amd import kindel as K
    R = refrun.load_reference()
    logging.disable(logging.WARNING)
    try:
        res = [RF.check_structured_seed(R, K, seed) for seed in range(2000, 2012)]
    finally:
        logging.disable(logging.NOTSET)
    assert not [d for d, _ in res if d], [d for d, _ in res if d]
    assert sum(bool(did) for _, did in res) >= 5      # realign changed the consensus of these files


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


@pytest.mark.gpu
def test_files_with_clip_dominant_regions_through_realign_on_the_gpu(hip_lib):
    """The structured files (clip-dominant regions that realign patches) through the reference and the HIP path."""
    import logging
    from kindel_amd import kindel as K
    R = refrun.load_reference()
    logging.disable(logging.WARNING)
    try:
        res = [RF.check_structured_seed(R, K, seed) for seed in range(2100, 2110)]
    finally:
        logging.disable(logging.NOTSET)
    assert not [d for d, _ in res if d], [d for d, _ in res if d]
    assert sum(bool(did) for _, did in res) >= 4
