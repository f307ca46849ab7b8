"""scripts/gpu_inflate_proto.h -- the GPU-side raw-DEFLATE prototype (one wavefront per BGZF block) -- executed
on the CPU through the test emulator and checked against zlib: round trips over payloads / levels / strategies, multi-block streams,
stored blocks, corrupted input (refused or decoded to the announced size, never a write outside the output: canary).  The same
kernel source is what scripts/gpu_inflate_proto.py times on the MI355X (profiles/r03_gpu_inflate_prototype.json)."""
import ctypes as C
import os
import random
import subprocess
import zlib

import numpy as np
import pytest

from tests.test_inflate import PAYLOADS, raw_deflate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "emu", "gpu_inflate_emu.cpp")
LIB = os.path.join(ROOT, "tests", "emu", "libgpu_inflate_emu.so")
DEPS = [SRC, os.path.join(ROOT, "tests", "emu", "hip_emu.h"), os.path.join(ROOT, "kindel_amd", "csrc", "kd_gpu_inflate.h"),
        os.path.join(ROOT, "kindel_amd", "csrc", "kd_gpu_inflate2.h")]


class GiBlock(C.Structure):
    _fields_ = [("in_off", C.c_uint64), ("out_off", C.c_uint64), ("in_len", C.c_uint32), ("out_len", C.c_uint32)]


class _Proto:
    """One of the two GPU inflaters behind the same call: the one-pass kernel (a wavefront per block, rounds 3 - 5) or round 6's two-pass
    pair (kd_gpu_inflate2.h: a lane per block records the matches, a wavefront per block resolves them)."""

    def __init__(self, dll, two_pass):
        self.dll, self.two_pass = dll, two_pass
        self.gi_inflate_blocks = dll.gi_inflate_blocks2 if two_pass else dll.gi_inflate_blocks
        self.gi_crc_blocks = dll.gi_crc_blocks


@pytest.fixture(scope="module", params=["one_pass", "two_pass"])
def proto(request):
    import __graft_entry__ as g
    dll = C.CDLL(g.build_inflate_emu())
    for f in (dll.gi_inflate_blocks, dll.gi_inflate_blocks2):
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        f.restype = C.c_int
    return _Proto(dll, request.param == "two_pass")


def inflate_blocks(dll, streams, sizes):
    """streams: raw DEFLATE byte strings, sizes: announced output sizes -> (list of outputs, status array)"""
    comp = b"".join(streams) + b"\0" * 64
    blocks = (GiBlock * len(streams))()
    at = out_at = 0
    for k, (z, n) in enumerate(zip(streams, sizes)):
        blocks[k] = GiBlock(at, out_at, len(z), n)
        at += len(z)
        out_at += n
    cbuf = np.frombuffer(comp, np.uint8).copy()
    out = np.full(out_at + 64, 0xA5, np.uint8)           # canary behind the last output
    status = np.full(len(streams), 99, np.uint32)
    rc = dll.gi_inflate_blocks(cbuf.ctypes.data, C.addressof(blocks), len(streams), out.ctypes.data, status.ctypes.data, 1, None)
    assert rc == 0
    assert (out[out_at:] == 0xA5).all(), "wrote past the end of the output"
    outs, o = [], 0
    for n in sizes:
        outs.append(out[o:o + n].tobytes())
        o += n
    return outs, status


@pytest.mark.parametrize("level", [0, 1, 6, 9])
@pytest.mark.parametrize("strategy", [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE],
                         ids=["default", "fixed", "huffman_only", "rle"])
def test_round_trips_against_zlib(proto, level, strategy):
    names = sorted(PAYLOADS)
    data = [PAYLOADS[n][:65280] for n in names]          # a BGZF block inflates to at most 0xff00 bytes
    streams = [raw_deflate(d, level, strategy) for d in data]
    outs, status = inflate_blocks(proto, streams, [len(d) for d in data])
    for n, d, o, st in zip(names, data, outs, status):
        assert st == 0, (n, int(st))
        assert o == d, n


def test_multi_block_streams_and_far_matches(proto):
    rng = random.Random(3)
    data = PAYLOADS["bam_like"][:65000]
    streams, sizes = [], []
    for mem in (1, 4, 9):
        c = zlib.compressobj(6, zlib.DEFLATED, -15, mem)
        z, at = b"", 0
        while at < len(data):       # Z_FULL_FLUSH: many deflate blocks (incl. empty stored ones) in one stream
            n = rng.randint(1, 3000)
            z += c.compress(data[at:at + n]) + c.flush(zlib.Z_FULL_FLUSH)
            at += n
        streams.append(z + c.flush()); sizes.append(len(data))
    far = os.urandom(700) + b"x" * 20000 + bytes(rng.getrandbits(8) for _ in range(9000)) + b"y" * 3000
    far = far + far[:700] + far[20000:29000] + far[:5]     # matches 30 k+ back: read from the flushed output, not the ring
    streams.append(raw_deflate(far, 9)); sizes.append(len(far))
    outs, status = inflate_blocks(proto, streams, sizes)
    assert status.tolist() == [0, 0, 0, 0]
    assert outs[0] == data and outs[1] == data and outs[2] == data and outs[3] == far


def test_wrong_sizes_and_corrupted_streams_are_refused_or_harmless(proto):
    rng = random.Random(11)
    data = PAYLOADS["bam_like"][:30000]
    z = raw_deflate(data, 6)
    outs, status = inflate_blocks(proto, [z, z, z[:len(z) // 2]], [len(data) + 1, len(data) - 1, len(data)])
    assert all(int(s) != 0 for s in status)
    streams = []
    for _ in range(300):
        zz = bytearray(z)
        for _ in range(rng.randint(1, 4)):
            zz[rng.randrange(len(zz))] ^= 1 << rng.randrange(8)
        streams.append(bytes(zz))
    outs, status = inflate_blocks(proto, streams, [len(data)] * len(streams))      # (asserts the canary)
    refused = int((status != 0).sum())
    assert refused > 100
    for zz, o, st in zip(streams, outs, status):    # what it accepts, zlib accepts with the same bytes (no CRC at this level)
        if st == 0:
            assert zlib.decompress(zz, -15)[:len(data)] == o


def test_lane_parallel_crc_model_equals_zlib():
    """The decomposition k_bgzf_crc uses (kd_gpu_inflate.h), in Python: lane l owns dword column l of the rows of 256 bytes,
    acc = Z_256(acc) ^ raw(dword) per row, every column shifted to the block's end, the columns xor-ed, the initial state n bytes on."""
    POLY = 0xEDB88320
    T = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ POLY if c & 1 else c >> 1
        T.append(c)

    def multmodp(a, b):
        m, p = 1 << 31, 0
        while True:
            if a & m:
                p ^= b
                if (a & (m - 1)) == 0:
                    return p
            m >>= 1
            b = (b >> 1) ^ POLY if b & 1 else b >> 1
    x2n = [1 << 30]
    for _ in range(24):
        x2n.append(multmodp(x2n[-1], x2n[-1]))

    def xpow8(k):
        n, p, j = 8 * k, 1 << 31, 0
        while n:
            if n & 1:
                p = multmodp(x2n[j], p)
            n >>= 1
            j += 1
        return p

    def raw(bs, s=0):
        for b in bs:
            s = T[(s ^ b) & 0xff] ^ (s >> 8)
        return s
    c256 = xpow8(256)
    for n in (0, 1, 3, 4, 5, 255, 256, 257, 300, 511, 512, 1000, 4097, 65279, 65280):
        M = os.urandom(n)
        rows, total = n // 256, 0
        for l in range(64):
            acc, end = 0, 0
            for j in range(rows):
                acc = multmodp(c256, acc) ^ raw(M[256 * j + 4 * l:256 * j + 4 * l + 4]) if acc else raw(M[256 * j + 4 * l:256 * j + 4 * l + 4])
            if rows:
                end = 256 * (rows - 1) + 4 * l + 4
            s0 = 256 * rows + 4 * l
            if s0 < n:
                d = M[s0:min(s0 + 4, n)]
                acc = (multmodp(xpow8(s0 + len(d) - end), acc) if end and acc else 0) ^ raw(d)
                end = s0 + len(d)
            if end:
                total ^= multmodp(xpow8(n - end), acc) if acc else 0
        total ^= multmodp(xpow8(n), 0xffffffff)
        assert total ^ 0xffffffff == zlib.crc32(M), n


def test_block_crc_kernel_against_zlib(proto):
    """k_bgzf_crc on the emulator: blocks of every size class (empty, shorter than a row, partial last rows, a full BGZF block) laid
    out as in a BGZF file (payload, CRC-32, ISIZE); a flipped bit in one inflated block is counted, and only that one."""
    import struct
    rng = random.Random(9)
    datas = [b"", b"a", b"abc", b"abcd", os.urandom(255), os.urandom(256), os.urandom(257), os.urandom(1000), PAYLOADS["bam_like"][:65280],
             PAYLOADS["acgt"][:4097], b"\xff" * 65279] + [os.urandom(rng.randint(1, 3000)) for _ in range(20)]
    comp, blocks, outs = b"", (GiBlock * len(datas))(), b""
    for k, d in enumerate(datas):
        z = raw_deflate(d, 6)
        blocks[k] = GiBlock(len(comp), len(outs), len(z), len(d))
        comp += z + struct.pack("<II", zlib.crc32(d), len(d))
        outs += d
    proto.gi_crc_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
    cbuf = np.frombuffer(comp + b"\0" * 16, np.uint8).copy()
    for grid in (1, 3, 64):           # a wavefront takes blocks in turn
        obuf = np.frombuffer(outs + b"\0" * 16, np.uint8).copy()
        bad = np.zeros(1, np.uint32)
        assert proto.gi_crc_blocks(cbuf.ctypes.data, C.addressof(blocks), len(datas), obuf.ctypes.data, bad.ctypes.data, grid) == 0
        assert int(bad[0]) == 0, grid
        obuf[int(blocks[8].out_off) + 4321] ^= 0x10
        assert proto.gi_crc_blocks(cbuf.ctypes.data, C.addressof(blocks), len(datas), obuf.ctypes.data, bad.ctypes.data, grid) == 0
        assert int(bad[0]) == 1, grid


def test_a_second_stored_block_is_bounded_by_what_is_left_of_the_input(proto):
    """stored(300 bytes) followed by a stored block that CLAIMS 200 bytes while only 10 are left of the BGZF block's input: the length
    check must be made against the remaining input, not against the whole block's (the window restarts behind a stored block) --
    else the kernel reads on into the next block's compressed bytes and reports success (round 3's review)."""
    import struct
    first = bytes(range(256)) + bytes(44)
    good = b"\x00" + struct.pack("<HH", 300, 300 ^ 0xffff) + first
    lying = good + b"\x01" + struct.pack("<HH", 200, 200 ^ 0xffff) + b"Z" * 10
    follower = raw_deflate(b"the next block's bytes " * 40, 6, zlib.Z_DEFAULT_STRATEGY)
    outs, status = inflate_blocks(proto, [lying, follower], [500, 23 * 40])
    assert status[0] == 7, int(status[0])            # GI_E_INPUT
    assert status[1] == 0 and outs[1] == b"the next block's bytes " * 40
    # the honest version of the same shape: two stored blocks, the second one complete
    honest = good + b"\x01" + struct.pack("<HH", 10, 10 ^ 0xffff) + b"Z" * 10
    outs, status = inflate_blocks(proto, [honest, follower], [310, 23 * 40])
    assert status[0] == 0 and outs[0] == first + b"Z" * 10 and status[1] == 0
    # ... and a dynamic block behind a stored one whose code words run past the block's end is still caught by the closing check
    cut = good + follower[:len(follower) // 2]
    outs, status = inflate_blocks(proto, [cut, follower], [300 + 23 * 40, 23 * 40])
    assert status[0] != 0 and status[1] == 0
