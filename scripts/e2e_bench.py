#!/usr/bin/env python
"""End-to-end leg of SURVEY 8d (ii): BAM path -> FASTA bytes, on the GPU box's own host cores.

Not the headline number (bench.py's `value` is the device-resident rate): this shows where a real run spends its time.
Two ingest paths are timed on the same file:
  whole   kd_decode_open (whole file) -> kd_push_batch (one batch) -> finalize -> consensus
  stream  kd_stream_open -> kd_push_stream: a decoder thread produces batch k+1 while the pushing thread copies batch k to
          the device and launches its kernels (what kindel_amd.kindel.bam_to_consensus does) -> finalize -> consensus
Usage: python scripts/e2e_bench.py [--config C3] [--scale 1.0] [--threads 0] [--chunk-mb 64] [--out profiles/x.json]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kindel_amd import _native as N, kindel as K  # noqa: E402
from tools import synth  # noqa: E402


def _quota():
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(p), 2)
    except Exception:
        return None


def fasta_of(pl):
    done = K._device_consensus_all(pl, {c: None for c in pl.order}, False, 1, False)
    return "".join(">%s_cns\n%s\n" % (pl.names[c], done[c][0]) for c in pl.order).encode()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--chunk-mb", type=int, default=64)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--out", default="")
    ap.add_argument("--qual", default="absent", choices=["absent", "phred"], help="base qualities of the written BAM: absent (0xff, compresses "
                    "~15x) or a Phred-like spread (compresses 3-4x like sequencer output: what the inflater really has to do)")
    ap.add_argument("--no-gpu-ingest", action="store_true", help="skip the device-side ingest leg (kd_push_bam_gpu)")
    ap.add_argument("--check", action="store_true", help="compare the FASTA with the oracle's consensus of the same batch (full-size bit-exactness)")
    ap.add_argument("--sweep", default="", help="streamed ingest only, for each THREADSxCHUNK_MB of a comma-separated list (e.g. 16x64,24x128)")
    a = ap.parse_args()

    t0 = time.time()
    tb = synth.make(a.config, scale=a.scale, device="cuda:0")
    _, ev, _ = synth.counts(tb)
    batch = synth.to_numpy(tb)
    del tb
    path = os.path.join(tempfile.gettempdir(), "kd_e2e_%s_%g.bam" % (a.config, a.scale))
    t1 = time.time()
    if a.qual == "phred":
        os.environ["KD_WRITE_BAM_QUAL"] = "phred"
    N.write_bam(path, batch, threads=a.threads)
    os.environ.pop("KD_WRITE_BAM_QUAL", None)
    want_fasta = None
    if a.check:     # the oracle's consensus of the batch that was written (TEST INFRASTRUCTURE, outside every timed region)
        from oracle import oracle as ko
        names = ["ctg%d" % i for i in range(len(batch["contig_lens"]))]
        want_fasta = "".join(">%s_cns\n%s\n" % (names[c], ko.parse_records(batch, c).consensus_sequence()[0]) for c in ko.contig_order(batch)).encode()
    t_write = time.time() - t1
    t_make = time.time() - t0
    size = os.path.getsize(path)
    n_reads = int(len(batch["contig"]))
    del batch

    def whole():
        out = {}
        t = time.perf_counter()
        b = N.decode_file(path, threads=a.threads)
        out["decode_s"] = time.perf_counter() - t
        t = time.perf_counter()
        pl = K.pileup_batch(b, bam_path=path)
        out["pileup_s"] = time.perf_counter() - t
        t = time.perf_counter()
        fa = fasta_of(pl)
        out["consensus_s"] = time.perf_counter() - t
        out["total_s"] = out["decode_s"] + out["pileup_s"] + out["consensus_s"]
        pl.engine.close()
        return out, fa

    def stream():
        out = {}
        t = time.perf_counter()
        pl = K.pileup_file(path, threads=a.threads, chunk_bytes=a.chunk_mb << 20, stream=True)
        out["ingest_s"] = time.perf_counter() - t
        out.update({"ingest_" + k: v for k, v in pl.ingest.items()})
        t = time.perf_counter()
        fa = fasta_of(pl)
        out["consensus_s"] = time.perf_counter() - t
        out["total_s"] = out["ingest_s"] + out["consensus_s"]
        pl.engine.close()
        return out, fa

    def gpu_ingest():      # the device-side ingest (kd_push_bam_gpu): BGZF inflate + record walk + batch arrays on the GPU
        out = {}
        t = time.perf_counter()
        pl = K.pileup_file(path, ingest="gpu")
        out["ingest_s"] = time.perf_counter() - t
        assert pl.ingest.get("path") == "gpu", "the device-side ingest handed the file to the host decoder"
        out.update({"ingest_" + k: v for k, v in pl.ingest.items() if k != "path"})
        t = time.perf_counter()
        fa = fasta_of(pl)
        out["consensus_s"] = time.perf_counter() - t
        out["total_s"] = out["ingest_s"] + out["consensus_s"]
        pl.engine.close()
        return out, fa

    if a.sweep:
        for spec in a.sweep.split(","):
            th, mb = (int(x) for x in spec.split("x"))
            a.threads, a.chunk_mb = th, mb
            rs = [stream()[0] for _ in range(a.repeat)]
            b = min(rs, key=lambda r: r["total_s"])
            print("sweep threads %d chunk %d MB: stream %.3f s (decode %.3f, push %.3f) %.3g events/s   [all: %s]" % (
                th, mb, b["total_s"], b["ingest_decode_s"], b["ingest_push_s"], ev / b["total_s"], " ".join("%.3f" % r["total_s"] for r in rs)))
        os.unlink(path)
        return
    runs_w = [whole() for _ in range(a.repeat)]
    runs_s = [stream() for _ in range(a.repeat)]
    bw = min(runs_w, key=lambda r: r[0]["total_s"])
    bs = min(runs_s, key=lambda r: r[0]["total_s"])
    assert bw[1] == bs[1], "streamed and whole-file FASTA differ"
    bg = None
    if not a.no_gpu_ingest:
        runs_g = [gpu_ingest() for _ in range(a.repeat)]
        bg = min(runs_g, key=lambda r: r[0]["total_s"])
        assert bg[1] == bs[1], "device-side ingest and host decoder give different FASTA"
    if want_fasta is not None:
        assert bs[1] == want_fasta, "end-to-end FASTA differs from the oracle's"
    import hashlib
    line = {
        "what": "end-to-end BAM path -> FASTA bytes (SURVEY 8d ii)", "config": a.config, "scale": a.scale, "reads": n_reads,
        "aligned_events": ev, "bam_bytes": size, "host_cores_visible": os.cpu_count(), "host_cpu_quota": _quota(), "decode_threads": a.threads or N.host_threads(),
        "best_of": a.repeat, "chunk_mb": a.chunk_mb, "fasta_bytes": len(bw[1]), "same_fasta": True, "qualities": a.qual,
        "bit_exact_vs_oracle": (True if want_fasta is not None else None),
        "consensus_sha256": hashlib.sha256(b"\n".join(l for l in bw[1].split(b"\n") if l and not l.startswith(b">"))).hexdigest(),
        "whole_file": {k: round(v, 4) for k, v in bw[0].items()}, "whole_file_events_per_s": ev / bw[0]["total_s"],
        "streamed": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in bs[0].items()},
        "streamed_events_per_s": ev / bs[0]["total_s"],
        "synth_plus_write_s": round(t_make, 1), "native_bam_write_s": round(t_write, 2),
    }
    if bg is not None:
        line["gpu_ingest"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in bg[0].items()}
        line["gpu_ingest_events_per_s"] = ev / bg[0]["total_s"]
        line["gpu_ingest_same_fasta"] = True
    s = json.dumps(line)
    print(s)
    if a.out:
        with open(a.out, "w") as fh:
            fh.write(s + "\n")
    os.unlink(path)


if __name__ == "__main__":
    main()
