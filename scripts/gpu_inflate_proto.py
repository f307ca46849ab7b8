#!/usr/bin/env python
"""GPU-side BGZF inflate PROTOTYPE, measured (VERDICT r2 "next" 9): scripts/gpu_inflate_proto.h -- one wavefront per
BGZF block -- against the host decoder on the same file, on the GPU box.  Not a product path: go / no-go evidence for DESIGN.md.

  python scripts/gpu_inflate_proto.py [--config C3 --scale 0.1] --out gpurun_out/r03_gpu_inflate_prototype.json

For each quality mode (absent: 0xff bytes, compresses 15 x; phred: sequencer-like, 2.2 x) it writes the config's reads as a BAM with the
native writer, puts the FILE in HBM, inflates every BGZF block with k_gpu_inflate (best of 5 launches, hipEvents), checks every block's
bytes against zlib on the host, and times the host decoder (kd_decode_*: inflate + record walk, all host threads) on the same file.
"""
import argparse
import ctypes as C
import json
import os
import struct
import subprocess
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "exp", "libgpu_inflate_proto.so")
SRC = os.path.join(ROOT, "scripts", "gpu_inflate_proto.hip")


def build():
    deps = [SRC, os.path.join(ROOT, "scripts", "gpu_inflate_proto.h"), os.path.join(ROOT, "kindel_amd", "csrc", "kd_gpu_inflate.h"),
            os.path.join(ROOT, "kindel_amd", "csrc", "kd_gpu_inflate2.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", SRC, "-o", LIB])
    return LIB


def bgzf_blocks(raw):
    """-> arrays (payload offset, payload bytes, inflated bytes) of the file's BGZF blocks"""
    off, ilen, olen = [], [], []
    o, n = 0, len(raw)
    while o + 18 <= n:
        assert raw[o] == 0x1f and raw[o + 1] == 0x8b and raw[o + 3] & 4, "not BGZF"
        xlen = struct.unpack_from("<H", raw, o + 10)[0]
        x, bsize = o + 12, None
        while x < o + 12 + xlen:
            si1, si2, sl = raw[x], raw[x + 1], struct.unpack_from("<H", raw, x + 2)[0]
            if si1 == 66 and si2 == 67:
                bsize = struct.unpack_from("<H", raw, x + 4)[0] + 1
            x += 4 + sl
        assert bsize
        off.append(o + 12 + xlen); ilen.append(bsize - (12 + xlen) - 8); olen.append(struct.unpack_from("<I", raw, o + bsize - 4)[0])
        o += bsize
    return np.asarray(off, np.uint64), np.asarray(ilen, np.uint32), np.asarray(olen, np.uint32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--scale", type=float, default=0.1)
    ap.add_argument("--out", default="")
    ap.add_argument("--lib", default="", help="a prebuilt variant of exp/libgpu_inflate_proto.so")
    ap.add_argument("--clocks", action="store_true", help="the library was built with -DGI_CLOCKS: print where the wavefronts' clocks went")
    ap.add_argument("--two-pass", action="store_true", help="round 6: time kd_gpu_inflate2.h (a lane per block records the matches, a wavefront per block resolves them)")
    ap.add_argument("--no-verify", action="store_true", help="skip the zlib comparison and the host decoder (variants: timing only)")
    a = ap.parse_args()
    import torch
    from kindel_amd import _native as N
    from tools import synth
    dll = C.CDLL(a.lib or build())
    dll.gi_inflate_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float)]
    dll.gi_inflate_blocks.restype = C.c_int
    dll.gi_inflate_blocks2.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_uint64]
    dll.gi_inflate_blocks2.restype = C.c_int
    batch = synth.to_numpy(synth.make(a.config, scale=a.scale, device="cuda:0"))
    res = {"_what": "PROTOTYPE: raw DEFLATE of BGZF blocks on the GPU, one wavefront per block (scripts/gpu_inflate_proto.h), vs the host decoder "
                    "on the same file in the same run; every block's bytes compared with zlib",
           "config": a.config, "scale": a.scale, "reads": int(len(batch["contig"])), "host_threads": N.host_threads(), "runs": {}}
    for qual in ("absent", "phred"):
        path = os.path.join(tempfile.gettempdir(), "kd_gi_%s_%g_%s.bam" % (a.config, a.scale, qual))
        if qual == "phred":
            os.environ["KD_WRITE_BAM_QUAL"] = "phred"
        N.write_bam(path, batch)
        os.environ.pop("KD_WRITE_BAM_QUAL", None)
        raw = open(path, "rb").read()
        off, ilen, olen = bgzf_blocks(raw)
        keep = olen > 0
        off, ilen, olen = off[keep], ilen[keep], olen[keep]
        nb = len(off)
        ooff = np.concatenate([[0], np.cumsum(olen.astype(np.uint64))]).astype(np.uint64)
        total = int(ooff[-1])
        blocks = np.zeros(nb, dtype=[("in_off", "<u8"), ("out_off", "<u8"), ("in_len", "<u4"), ("out_len", "<u4")])
        blocks["in_off"], blocks["out_off"], blocks["in_len"], blocks["out_len"] = off, ooff[:-1], ilen, olen
        d_comp = torch.from_numpy(np.frombuffer(raw + b"\0" * 64, np.uint8).copy()).cuda()
        d_blocks = torch.from_numpy(blocks.view(np.uint8).copy()).cuda()
        d_out = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
        d_status = torch.zeros(nb + 64, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ms2 = (C.c_float * 2)(0, 0)
        if a.two_pass:
            rc = dll.gi_inflate_blocks2(d_comp.data_ptr(), d_blocks.data_ptr(), nb, d_out.data_ptr(), d_status.data_ptr(), 5, ms2, total)
        else:
            rc = dll.gi_inflate_blocks(d_comp.data_ptr(), d_blocks.data_ptr(), nb, d_out.data_ptr(), d_status.data_ptr(), 1 if a.clocks else 5, ms2)
        assert rc == 0, rc
        ms = C.c_float(ms2[0])
        status = d_status.cpu().numpy()
        if a.clocks:
            dbg = status[(nb + 1) & ~1:][:22].view(np.uint64)
            names = ["build", "symbols", "near", "far", "flush", "other", "n_lit", "n_near", "n_far", "n_flush", "n_slow"]
            tot = float(sum(int(x) for x in dbg[:6])) or 1.0
            print(qual, "clocks:", " ".join("%s=%.1f%%" % (n, 100.0 * int(v) / tot) for n, v in zip(names[:6], dbg[:6])),
                  "| per block:", " ".join("%s=%.0f" % (n, int(v) / nb) for n, v in zip(names[6:], dbg[6:11])),
                  "| clocks per symbol %.0f" % (int(dbg[1]) / max(1, int(dbg[6]) + int(dbg[7]) + int(dbg[8]))), flush=True)
        status = status[:nb]
        out = d_out.cpu().numpy()
        if a.no_verify:
            print(qual, "gpu_ms %.3f (pass 1 %.3f) GBps_out %.1f blocks_ok %d / %d" % (ms.value, ms2[1], total / ms.value * 1e-6, int((status == 0).sum()), nb), flush=True)
            os.remove(path)
            continue
        t0 = time.perf_counter()
        same = True
        for k in range(nb):      # zlib is the checker (one core: this is the slow part of the script)
            want = zlib.decompress(raw[int(off[k]):int(off[k]) + int(ilen[k])], -15)
            if out[int(ooff[k]):int(ooff[k + 1])].tobytes() != want:
                same = False
                break
        t_zlib = time.perf_counter() - t0
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            d = N.decode_file(path)
            best = min(best, time.perf_counter() - t0)
        n_rec = int(d["contig"].size)
        del d
        res["runs"][qual] = {
            "file_bytes": len(raw), "inflated_bytes": total, "compression": round(total / len(raw), 2), "blocks": nb,
            "kernel": "two-pass (k_inflate_tokens + k_inflate_resolve)" if a.two_pass else "one-pass (k_gpu_inflate)",
            **({"pass1_ms": round(float(ms2[1]), 3), "pass2_ms": round(float(ms2[0] - ms2[1]), 3)} if a.two_pass else {}),
            "gpu_ms": round(float(ms.value), 3), "gpu_GBps_out": round(total / ms.value * 1e-6, 1), "gpu_GBps_in": round(len(raw) / ms.value * 1e-6, 1),
            "blocks_ok": int((status == 0).sum()), "bytes_equal_zlib": bool(same and (status == 0).all()),
            "h2d_of_the_file_at_50GBps_ms": round(len(raw) / 50e9 * 1e3, 2),
            "host_decode_s": round(best, 4), "host_decode_GBps_out": round(total / best * 1e-9, 2), "host_records": n_rec,
            "zlib_one_core_GBps": round(total / t_zlib * 1e-9, 3),
            "speedup_vs_host_decode": round(best * 1e3 / ms.value, 1)}
        print(qual, json.dumps(res["runs"][qual]), flush=True)
        os.remove(path)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
