"""kd_host_inflate (kindel_amd/csrc/kd_inflate.h): the BGZF reader's raw-DEFLATE block decoder against zlib.

The decoder's contract: the stream must decode to exactly the announced size, malformed / truncated input is refused, and
nothing is written outside the output buffer (the binding puts a canary behind it).  zlib is the checker here.
"""
import os
import random
import zlib

import pytest

from kindel_amd import _native as N


def raw_deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=9):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strategy)
    return c.compress(data) + c.flush()


def _payloads():
    rng = random.Random(7)
    bam_like = b"".join(
        bytes([rng.randrange(256) for _ in range(8)]) + b"r%07d\0" % i + bytes(rng.choice(b"\x11\x12\x14\x18\x21\x22\x24\x28\x41\x42\x44\x48\x81\x82\x84\x88")
                                                                                   for _ in range(75)) + b"I" * 150
        for i in range(400))
    return {
        "empty": b"",
        "one_byte": b"a",
        "run": b"a" * 70000,                                   # distance 1, maximum-length matches
        "short_period": b"abcde" * 9000,                       # distances < 8
        "random": bytes(rng.getrandbits(8) for _ in range(65536)),   # stored blocks at any level
        "acgt": bytes(rng.choice(b"ACGT") for _ in range(150000)),
        "acgt_runs": b"".join(bytes([rng.choice(b"ACGT")]) * rng.randint(1, 40) for _ in range(12000)),
        "bam_like": bam_like,
        "all_bytes": bytes(range(256)) * 200,
        "far_matches": os.urandom(300) + b"x" * 32000 + os.urandom(300) * 2,
    }


PAYLOADS = _payloads()


@pytest.mark.parametrize("name", sorted(PAYLOADS))
@pytest.mark.parametrize("level", [0, 1, 6, 9])
@pytest.mark.parametrize("strategy", [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE],
                         ids=["default", "fixed", "huffman_only", "rle"])
def test_round_trip_against_zlib(emu_lib, name, level, strategy):
    data = PAYLOADS[name]
    z = raw_deflate(data, level, strategy)
    assert N.host_inflate(z, len(data), lib=emu_lib) == data
    for wrong in {len(data) + 1, max(len(data) - 1, 0)} - {len(data)}:
        with pytest.raises(ValueError):
            N.host_inflate(z, wrong, lib=emu_lib)
    if len(data) > 0 and len(z) > 4:
        with pytest.raises(ValueError):
            N.host_inflate(z[:len(z) // 2], len(data), lib=emu_lib)


def test_many_small_blocks_and_window_sizes(emu_lib):
    rng = random.Random(3)
    data = PAYLOADS["bam_like"]
    for mem in (1, 4, 9):
        c = zlib.compressobj(6, zlib.DEFLATED, -15, mem)
        z = b""
        at = 0
        while at < len(data):       # Z_FULL_FLUSH: many deflate blocks (incl. empty stored ones) in one stream
            n = rng.randint(1, 3000)
            z += c.compress(data[at:at + n]) + c.flush(zlib.Z_FULL_FLUSH)
            at += n
        z += c.flush()
        assert N.host_inflate(z, len(data), lib=emu_lib) == data


def test_corrupted_streams_never_crash_or_overrun(emu_lib):
    rng = random.Random(11)
    data = PAYLOADS["bam_like"]
    z = raw_deflate(data, 6)
    refused = 0
    for _ in range(1500):
        zz = bytearray(z)
        for _ in range(rng.randint(1, 4)):
            zz[rng.randrange(len(zz))] ^= 1 << rng.randrange(8)
        try:
            out = N.host_inflate(bytes(zz), len(data), lib=emu_lib)   # the binding asserts the canary behind the output
            assert len(out) == len(data)
        except ValueError:
            refused += 1
    assert refused > 500          # (a flipped bit can still give a well-formed stream of the same length: no CRC at this level)


def test_bgzf_reader_uses_it(emu_lib, tmp_path):
    from kindel_amd import synth
    batch = synth.to_numpy(synth.short_reads([3000], 40, seed=2))
    p = str(tmp_path / "x.bam")
    N.write_bam(p, batch, lib=emu_lib)
    from tests.test_decoder import same_batch
    same_batch(N.decode_file(p, lib=emu_lib), batch)
