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
DEPS = [SRC, os.path.join(ROOT, "tests", "emu", "hip_emu.h"), os.path.join(ROOT, "kindel_amd", "csrc", "kd_gpu_inflate.h")]


class GiBlock(C.Structure):
    _fields_ = [("in_off", C.c_uint64), ("out_off", C.c_uint64), ("in_len", C.c_uint32), ("out_len", C.c_uint32)]


@pytest.fixture(scope="module")
def proto():
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in DEPS):
        subprocess.check_call(["g++", "-O1", "-std=c++20", "-shared", "-fPIC", "-pthread", "-Wno-unknown-pragmas", SRC, "-o", LIB])
    dll = C.CDLL(LIB)
    dll.gi_inflate_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    dll.gi_inflate_blocks.restype = C.c_int
    return dll


def inflate_blocks(dll, streams, sizes):
    """streams: raw DEFLATE byte strings, sizes: announced output sizes -> (list of outputs, status array)"""
    comp = b"".join(streams) + b"\0" * 16
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
