#!/usr/bin/env python3
"""CAMPAIGN (round 6): the two GPU inflaters (one-pass k_gpu_inflate, two-pass k_inflate_tokens + k_inflate_resolve) on the kernel
emulator against zlib over random payloads x levels x strategies x window sizes, and over bit-flipped / truncated / spliced streams:
  * a stream zlib inflates to the announced size must come back with status 0 and the same bytes;
  * a stream zlib refuses (or that inflates to another size) must be refused (status != 0);
  * nothing is ever written behind the announced output (canary), whatever the stream.
    python scripts/exp/inflate_fuzz.py [rounds] [seed]          (KD_INFLATE_EMU_LIB: another build, e.g. the ASan one)
    python scripts/exp/inflate_fuzz.py [rounds] [seed] --gpu    the same streams through the kernels on the MI355X (scripts/gpu_inflate_proto.hip)
Prints one line per round and a summary; exit code 1 on any disagreement."""
import ctypes as C
import os
import random
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from tests.test_gpu_inflate_proto import GiBlock, _Proto, inflate_blocks
from tests.test_inflate import PAYLOADS


def payload(rng):
    kind = rng.randrange(7)
    n = rng.choice([0, 1, 2, 3, 7, 64, 257, 258, 259, 1000, 4096, 30000, 65280]) if rng.random() < 0.3 else rng.randrange(1, 65281)
    if kind == 0:
        return bytes(rng.getrandbits(8) for _ in range(min(n, 20000)))
    if kind == 1:
        return bytes([rng.randrange(4)]) * n
    if kind == 2:      # quality-like: few symbols, short runs
        return bytes(rng.choice(b"#',-5:<FI") for _ in range(min(n, 30000)))
    if kind == 3:      # repeats at every distance up to the window
        unit = bytes(rng.getrandbits(8) for _ in range(rng.choice([1, 2, 3, 4, 5, 8, 31, 32, 33, 255, 258, 1024, 32767, 32768])))
        return (unit * (n // max(1, len(unit)) + 1))[:n]
    if kind == 4:
        src = PAYLOADS[rng.choice(sorted(PAYLOADS))]
        a = rng.randrange(max(1, len(src) - 1))
        return src[a:a + n]
    if kind == 5:      # far matches: a block of noise repeated 20 - 32 k later
        noise = bytes(rng.getrandbits(8) for _ in range(rng.randrange(3, 600)))
        gap = bytes([rng.randrange(256)]) * rng.randrange(20000, 32700)
        return (noise + gap + noise + gap[:rng.randrange(1, 3000)] + noise)[:65280]
    return bytes(rng.choice(b"ACGTN") for _ in range(min(n, 40000)))


def deflate(rng, d):
    level = rng.choice([0, 1, 2, 4, 6, 9])
    strat = rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED])
    c = zlib.compressobj(level, zlib.DEFLATED, -rng.choice([9, 12, 15]), rng.choice([1, 8, 9]), strat)
    if rng.random() < 0.3:      # many deflate blocks in one stream
        z, at = b"", 0
        while at < len(d):
            k = rng.randrange(1, 5000)
            z += c.compress(d[at:at + k]) + c.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_NO_FLUSH]))
            at += k
        return z + c.flush()
    return c.compress(d) + c.flush()


def mutate(rng, z):
    zz = bytearray(z)
    how = rng.randrange(5)
    if how == 0 and zz:
        for _ in range(rng.randrange(1, 5)):
            zz[rng.randrange(len(zz))] ^= 1 << rng.randrange(8)
    elif how == 1 and len(zz) > 2:
        zz = zz[:rng.randrange(1, len(zz))]
    elif how == 2 and len(zz) > 8:
        a = rng.randrange(len(zz) - 4)
        zz[a:a + rng.randrange(1, 4)] = bytes(rng.getrandbits(8) for _ in range(rng.randrange(1, 6)))
    elif how == 3:
        zz = bytearray(rng.getrandbits(8) for _ in range(rng.randrange(1, 400)))
    else:
        zz += bytes(rng.getrandbits(8) for _ in range(rng.randrange(1, 9)))      # garbage behind the end-of-block
    return bytes(zz)


def zlib_says(z, n):
    """the bytes zlib gives for exactly this stream when it ends cleanly with n bytes out, else None"""
    try:
        d = zlib.decompressobj(-15)
        o = d.decompress(z, n + 1)
        if not d.eof or len(o) != n:
            return None
        return o
    except zlib.error:
        return None


class _GpuProto:
    """the same call on the GPU: device tensors, the prototype driver's entry points (all pointers device pointers)"""

    def __init__(self, dll, two_pass):
        import torch
        self.torch, self.dll, self.two_pass = torch, dll, two_pass

    def gi_inflate_blocks(self, comp_ptr, blocks_ptr, n, out_ptr, status_ptr, repeat, ms):
        torch = self.torch
        blocks = (GiBlock * n).from_address(blocks_ptr)
        n_comp = max(int(b.in_off) + int(b.in_len) for b in blocks) + 64
        n_out = sum(int(b.out_len) for b in blocks) + 64
        host_comp = np.ctypeslib.as_array((C.c_uint8 * n_comp).from_address(comp_ptr))
        host_out = np.ctypeslib.as_array((C.c_uint8 * n_out).from_address(out_ptr))
        host_status = np.ctypeslib.as_array((C.c_uint32 * n).from_address(status_ptr))
        d_comp = torch.from_numpy(host_comp.copy()).cuda()
        d_blocks = torch.from_numpy(np.frombuffer(bytes(blocks), np.uint8).copy()).cuda()
        d_out = torch.from_numpy(host_out.copy()).cuda()
        d_status = torch.from_numpy(host_status.copy().view(np.int32)).cuda()
        torch.cuda.synchronize()
        if self.two_pass:
            rc = self.dll.gi_inflate_blocks2(d_comp.data_ptr(), d_blocks.data_ptr(), n, d_out.data_ptr(), d_status.data_ptr(), 1, None, n_out - 64)
        else:
            rc = self.dll.gi_inflate_blocks(d_comp.data_ptr(), d_blocks.data_ptr(), n, d_out.data_ptr(), d_status.data_ptr(), 1, None)
        if rc:
            print("driver rc %d (%s, %d blocks, %d bytes in, %d out)" % (rc, "two_pass" if self.two_pass else "one_pass", n, n_comp, n_out), flush=True)
        torch.cuda.synchronize()
        host_out[:] = d_out.cpu().numpy()
        host_status[:] = d_status.cpu().numpy().view(np.uint32)
        return rc


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rounds = int(args[0]) if len(args) > 0 else 20
    seed = int(args[1]) if len(args) > 1 else 1
    if "--gpu" in sys.argv:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
        import torch      # (before the library: it must bind to the HIP runtime torch brings, not load a second one)
        torch.cuda.init()
        import gpu_inflate_proto as gp
        dll = C.CDLL(gp.build())
        dll.gi_inflate_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        dll.gi_inflate_blocks2.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
        dll.gi_inflate_blocks.restype = dll.gi_inflate_blocks2.restype = C.c_int
        protos = {"one_pass": _GpuProto(dll, False), "two_pass": _GpuProto(dll, True)}
    else:
        import __graft_entry__ as g
        dll = C.CDLL(os.environ.get("KD_INFLATE_EMU_LIB") or g.build_inflate_emu())
        for f in (dll.gi_inflate_blocks, dll.gi_inflate_blocks2):
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
            f.restype = C.c_int
        protos = {"one_pass": _Proto(dll, False), "two_pass": _Proto(dll, True)}
    rng = random.Random(seed)
    bad = 0
    tot = {"valid": 0, "mutated": 0, "mutated_still_valid": 0}
    for r in range(rounds):
        streams, sizes = [], []
        for _ in range(40):
            d = payload(rng)
            z = deflate(rng, d)
            streams.append(z); sizes.append(len(d)); tot["valid"] += 1
            for _ in range(4):
                streams.append(mutate(rng, z)); sizes.append(len(d) if rng.random() < 0.9 else max(0, len(d) + rng.randrange(-3, 4))); tot["mutated"] += 1
        want = [zlib_says(z, n) for z, n in zip(streams, sizes)]
        tot["mutated_still_valid"] += sum(1 for k, w in enumerate(want) if w is not None and k % 5)
        for name, p in protos.items():
            outs, status = inflate_blocks(p, streams, sizes)
            for k, (w, o, st) in enumerate(zip(want, outs, status)):
                if w is not None and (st != 0 or o != w):
                    bad += 1; print("MISMATCH %s round %d stream %d: zlib accepts %d bytes, kernel status %d, equal %s" % (name, r, k, len(w), int(st), o == w), flush=True)
                elif w is None and st == 0:
                    # zlib's strictness differs in ONE documented way: bytes behind the final block are not the kernel's business (a BGZF block's
                    # length comes from its header), so a stream that is valid up to its end-of-block and has garbage behind it is accepted
                    try:
                        d2 = zlib.decompressobj(-15); o2 = d2.decompress(streams[k], sizes[k] + 1)
                        ok = d2.eof and o2 == o and len(o2) == sizes[k]
                    except zlib.error:
                        ok = False
                    if not ok:
                        bad += 1; print("MISMATCH %s round %d stream %d: zlib refuses, kernel accepts" % (name, r, k), flush=True)
        print("round %d: %d streams, mismatches so far %d" % (r, len(streams), bad), flush=True)
    print("SUMMARY seed %d rounds %d: %s, mismatches %d" % (seed, rounds, tot, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
