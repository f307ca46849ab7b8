"""Seeded CIGAR fuzzing: random reads with arbitrary op sequences -- valid and invalid -- engine vs oracle.
CPU: kernel logic under the emulator.  GPU (`-m gpu`): the HIP library."""
import numpy as np
import pytest

from kindel_amd import _native as N
from tests import fuzz
from tests import parity as P

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


def _one_wild_long_read(rng, contig_lens, n_valid, ops_range, big=None):
    """A valid long-read batch with ONE read rebuilt without any validity constraint (ops_range CIGAR ops of any kind and length 0 - 11,
    POS anywhere from in front of the contig to behind its end, a query a few bases short or long) -- so that the exception the
    reference raises, if any, is THAT read's and can be compared by type.  big = (op, length): one op of the wild read gets that length
    (k_prep_long routes a read with an op of 2^23 bases or more to the exact walk)."""
    batch = fuzz.random_batch(rng, n_valid, contig_lens=contig_lens, wild=0.0, sort=False, long_ops=ops_range)
    i = int(rng.integers(0, n_valid))
    n_ops = int(rng.integers(*ops_range))
    ops = [(int(rng.integers(0, 12)), int(rng.integers(0, 9))) for _ in range(n_ops)]
    if big is not None:
        ops[int(rng.integers(0, n_ops))] = (big[1], big[0])
    qlen = sum(ln for ln, op in ops if op in (0, 1, 4, 7, 8) and ln < (1 << 20))
    sl = max(0, qlen + int(rng.integers(-3, 4)))
    L = int(batch["contig_lens"][int(batch["contig"][i])])
    nib = rng.choice([1, 2, 4, 8, 15, 3], sl, p=[0.2495, 0.2495, 0.2495, 0.2495, 0.0015, 0.0005]).astype(np.uint8)
    nn = np.concatenate([nib, np.zeros(len(nib) & 1, np.uint8)])
    packed = ((nn[0::2] << 4) | nn[1::2]).astype(np.uint8)
    batch["seq_off"][i] = len(batch["seq4"])
    batch["seq4"] = np.concatenate([batch["seq4"], packed, np.zeros(32, np.uint8)])
    batch["cig_off"][i] = len(batch["cigar"])
    batch["cigar"] = np.concatenate([batch["cigar"], np.asarray([(ln << 4) | op for ln, op in ops] + [0, 0], np.uint32)])
    batch["n_cig"][i] = n_ops
    batch["seq_len"][i] = sl
    batch["pos0"][i] = int(rng.integers(-3, L + 4)) if rng.random() < 0.5 else int(rng.integers(max(0, L - 40 * n_ops // 9), L + 4))
    batch["flag"][i] = 0
    return batch


def _wild_long_campaign(lib, seeds, ops_range, modes, big=None, contig_lens=(6000, 2500)):
    """-> {"ok": n, exception name: n}.  The engine must end exactly as the oracle does: the same tables, or the same exception type."""
    out = {}
    for seed in seeds:
        rng = np.random.default_rng(seed)
        batch = _one_wild_long_read(rng, contig_lens, 24, ops_range, big=big)
        kind, val = fuzz.oracle_outcome(batch)
        for mode in modes:
            if kind == "raise":
                with pytest.raises(val):
                    P.Run(lib, batch, mode=mode, window=[64, 256][seed % 2])
            else:
                P.assert_matches_oracle(P.Run(lib, batch, mode=mode, window=[64, 256][seed % 2]))
        key = val.__name__ if kind == "raise" else "ok"
        out[key] = out.get(key, 0) + 1
    return out


def test_fuzz_one_wild_read_of_several_tiles_emulated(emu_lib):
    """k_prep_long decides op-parallel, tile of 64 ops by tile, whether a long read is regular -- since round 6 branch-free on 32-bit
    tile-relative coordinates against limits clamped per tile (kd_long.h).  The valid long-read campaigns never show it an invalid read
    of more than one tile: here one read per batch is anything at all (65 - 400 ops: runs off either end of its contig in any tile,
    clips in the middle, a query too short or too long, bases outside the dict) and the engine must raise the reference's exception
    for it, by type, or produce the reference's tables."""
    out = _wild_long_campaign(emu_lib, range(7100, 7160), (65, 400), [N.KD_MODE_AUTO, N.KD_MODE_GLOBAL])
    assert out.get("ok", 0) >= 3 and out.get("IndexError", 0) >= 5 and out.get("KeyError", 0) >= 2, out
    out = _wild_long_campaign(emu_lib, range(7200, 7230), (17, 64), [N.KD_MODE_AUTO])
    assert sum(out.values()) == 30, out


@pytest.mark.parametrize("op,length", [(0, 1 << 23), (2, 1 << 23), (1, 1 << 23), (4, (1 << 23) + 5), (3, 1 << 24), (0, (1 << 28) - 1), (2, (1 << 23) - 1)])
def test_fuzz_one_wild_read_with_a_huge_op_emulated(emu_lib, op, length):
    """An op of 2^23 bases or more (M, I, D, S, N; and one just below the line) inside a long CIGAR: the read goes to the exact walk,
    the scans' 32-bit tile sums stay in range, and the outcome is the reference's."""
    out = _wild_long_campaign(emu_lib, range(7300 + op * 10, 7304 + op * 10), (65, 200), [N.KD_MODE_AUTO], big=(op, length))
    assert sum(out.values()) == 4, out


@pytest.mark.gpu
def test_fuzz_one_wild_read_of_several_tiles_gpu(hip_lib):
    out = _wild_long_campaign(hip_lib, range(7100, 7400), (65, 400), [N.KD_MODE_AUTO])
    assert out.get("ok", 0) >= 10 and out.get("IndexError", 0) >= 25 and out.get("KeyError", 0) >= 10, out
    for k, (op, length) in enumerate([(0, 1 << 23), (2, 1 << 23), (1, 1 << 23), (4, (1 << 23) + 5), (3, 1 << 24), (2, (1 << 23) - 1)]):
        out = _wild_long_campaign(hip_lib, range(7500 + 10 * k, 7506 + 10 * k), (65, 200), [N.KD_MODE_AUTO], big=(op, length))
        assert sum(out.values()) == 6, out


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
