"""Seeded CIGAR fuzzing: random reads with arbitrary op sequences -- valid and invalid -- engine vs oracle.
CPU: kernel logic under the emulator.  GPU (`-m gpu`): the HIP library."""
import numpy as np
import pytest

from kindel_amd import _native as N
from tests import fuzz

MODES = [N.KD_MODE_GLOBAL, N.KD_MODE_AUTO, N.KD_MODE_COOP, N.KD_MODE_STRIP]   # AUTO = k_window (lane per read), COOP = k_window_coop


def _campaign(lib, seeds, n_reads, wild):
    outcomes = {"ok": 0, "raise": 0}
    for seed in seeds:
        rng = np.random.default_rng(seed)
        batch = fuzz.random_batch(rng, n_reads, wild=wild, sort=bool(seed & 1))
        for mode in MODES:
            outcomes[fuzz.check_engine(lib, batch, mode, window=64, slice_reads=[0, 16][seed % 2])] += 1
    return outcomes


def test_fuzz_valid_reads_emulated(emu_lib):
    out = _campaign(emu_lib, range(100, 112), n_reads=60, wild=0.0)
    assert out["ok"] >= 8      # mostly valid inputs: the comparison is on full tables


def test_fuzz_wild_reads_emulated(emu_lib):
    out = _campaign(emu_lib, range(200, 212), n_reads=12, wild=0.3)
    assert out["raise"] >= 2 and out["ok"] >= 2


@pytest.mark.parametrize("n_reads", [64, 65, 2049, 8192, 8193, 17000])
def test_fuzz_batch_sizes_around_the_prep_block_emulated(emu_lib, n_reads):
    """k_prep classifies 8192 reads per workgroup, 2048 per wavefront, and hands out record slots / insertion slots with
    wavefront-wide votes and prefix sums in which lanes past the end of the batch take part: sizes around those boundaries."""
    rng = np.random.default_rng(7000 + n_reads)
    batch = fuzz.random_batch(rng, n_reads, contig_lens=(5000, 3000, 800), wild=0.0 if n_reads % 3 else 0.02, sort=bool(n_reads & 1))
    for mode in (N.KD_MODE_GLOBAL, N.KD_MODE_AUTO):
        assert fuzz.check_engine(emu_lib, batch, mode, window=[64, 256, 640][n_reads % 3], slice_reads=[0, 16][n_reads % 2]) in ("ok", "raise")


def _long_campaign(lib, seeds, n_reads, long_ops=(17, 700), contig_lens=(6000, 2500)):
    """Reads with hundreds of ops (clips at both ends, indels, N/H/P) on contigs that hold them: the long-read
    path (k_prep_long, k_long_expand's rows, k_window's row pass), sorted and unsorted, two window sizes."""
    n_ok = 0
    for seed in seeds:
        rng = np.random.default_rng(seed)
        batch = fuzz.random_batch(rng, n_reads, contig_lens=contig_lens, wild=0.0, sort=bool(seed & 1), long_ops=long_ops)
        for mode in MODES:
            n_ok += fuzz.check_engine(lib, batch, mode, window=[64, 256][seed % 2], slice_reads=[0, 16][(seed >> 1) % 2]) == "ok"
    return n_ok


def test_fuzz_long_cigars_emulated(emu_lib):
    assert _long_campaign(emu_lib, range(300, 304), n_reads=40) >= 6


def test_fuzz_long_cigars_of_several_segments_emulated(emu_lib):
    """Very long CIGARs (480 - 2 600 words: dozens of k_long_expand's 64-op tiles per read) with insertion runs, deletions and the
    trailing clip anywhere in them.  (Written for round 5's expansion by 512-word segments, which was measured slower and dropped --
    profiles/r05_long_expand_segments_nogo.json; kept as long-CIGAR coverage: kd_long.h has no segments.)"""
    assert _long_campaign(emu_lib, range(310, 316), n_reads=30, long_ops=(480, 2600), contig_lens=(30000, 14000)) >= 8


def _mixed_campaign(lib, seeds, sizes):
    for seed in seeds:
        rng = np.random.default_rng(seed)
        L, n_reads = sizes[seed % len(sizes)]
        batch = fuzz.mixed_batch(rng, L, n_reads, piled=bool(seed & 1))
        window, slice_reads = [0, 0, 64, 256, 448, 1024, 2048][seed % 7], [0, 0, 64, 300, 2000][seed % 5]
        assert fuzz.check_engine(lib, batch, N.KD_MODE_AUTO, window=window, slice_reads=slice_reads) == "ok", (seed, L, n_reads, window, slice_reads)


def test_fuzz_mixed_shapes_at_depth_emulated(emu_lib):
    """What k_window's tile lists carry and what they do not (long plain reads, long leading clips, long deletions), at depths where its
    queue regimes and the deep-tile walk switch; a few thousand such batches ran clean as a local campaign, these seeds stay."""
    _mixed_campaign(emu_lib, range(900, 935), [(700, 1500), (1500, 4000), (4000, 9000), (9000, 300), (20000, 4000)])


def test_fuzz_with_the_launch_geometry_of_the_gpu_emulated(emu_lib, monkeypatch):
    """The emulator sizes its grids for 2 CUs by default; an MI355X launches persistent kernels with 256 x k workgroups, most of them
    idle on small inputs (k_window: 1280 workgroups for a 600-site contig) -- KD_EMU_CUS=256 gives the emulator that geometry
    (round 5, after a fault on hardware that no emulated run showed): the same campaigns, fewer seeds."""
    monkeypatch.setenv("KD_EMU_CUS", "256")
    out = _campaign(emu_lib, range(100, 104), n_reads=60, wild=0.0)
    assert out["ok"] >= 3
    out = _campaign(emu_lib, range(200, 204), n_reads=12, wild=0.3)
    assert out["ok"] + out["raise"] == 16
    _mixed_campaign(emu_lib, range(900, 907), [(700, 1500), (1500, 4000), (4000, 9000), (9000, 300), (20000, 4000)])


@pytest.mark.gpu
def test_fuzz_mixed_shapes_at_depth_gpu(hip_lib):
    _mixed_campaign(hip_lib, range(900, 914), [(700, 6000), (4000, 30000), (60000, 20000), (300000, 40000)])


def test_oracle_fuzz_is_deterministic():
    a = fuzz.random_batch(np.random.default_rng(5), 30)
    b = fuzz.random_batch(np.random.default_rng(5), 30)
    assert all(np.array_equal(a[k], b[k]) for k in a)


@pytest.mark.gpu
def test_fuzz_valid_reads_gpu(hip_lib):
    out = _campaign(hip_lib, range(1000, 1150), n_reads=300, wild=0.0)
    assert out["ok"] >= 100


@pytest.mark.gpu
def test_fuzz_wild_reads_gpu(hip_lib):
    out = _campaign(hip_lib, range(2000, 2400), n_reads=10, wild=0.25)
    assert out["raise"] >= 50 and out["ok"] >= 50


@pytest.mark.gpu
def test_fuzz_large_valid_batch_gpu(hip_lib):
    rng = np.random.default_rng(77)
    batch = fuzz.random_batch(rng, 20000, contig_lens=(5000, 3000, 800), wild=0.0, sort=True)
    for mode in MODES:
        assert fuzz.check_engine(hip_lib, batch, mode, window=256) in ("ok", "raise")


@pytest.mark.gpu
def test_fuzz_long_cigars_gpu(hip_lib):
    assert _long_campaign(hip_lib, range(3000, 3040), n_reads=200) >= 60
