#!/usr/bin/env python
"""End-to-end leg of SURVEY §8d: BAM path -> FASTA bytes through the public Python API, phase by phase.

Not the headline number (bench.py's `value` is the device-resident rate): this shows where a real run
spends its time -- host BGZF/BAM decode, PCIe, kernels, host report/FASTA -- on the GPU box's own cores.
Usage: python scripts/e2e_bench.py [--config C3] [--scale 0.1] [--threads 0] [--out profiles/x.json]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kindel_amd import _native as N, kindel as K, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--scale", type=float, default=0.1)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--out", default="")
    a = ap.parse_args()

    t0 = time.time()
    tb = synth.make(a.config, scale=a.scale)
    _, ev, _ = synth.counts(tb)                         # flagged/short reads are not in synthetic batches
    batch = synth.to_numpy(tb)
    path = os.path.join(tempfile.gettempdir(), "kd_e2e_%s_%g.bam" % (a.config, a.scale))
    synth.write_bam(path, batch)
    t_make = time.time() - t0
    size = os.path.getsize(path)

    def phases():
        out = {}
        t = time.perf_counter()
        b = N.decode_file(path, threads=a.threads)
        out["decode_s"] = time.perf_counter() - t
        t = time.perf_counter()
        pl = K.pileup_batch(b, bam_path=path)           # kd_create + kd_push_batch (H2D) + kd_finalize
        out["pileup_s"] = time.perf_counter() - t
        t = time.perf_counter()
        recs = []
        for cid in pl.order:
            seq, ch, mm = K._device_consensus(pl, cid, None, False, 1, False)
            recs.append(">%s_cns\n%s\n" % (pl.names[cid], seq))
        fasta = "".join(recs).encode()
        out["consensus_s"] = time.perf_counter() - t
        out["fasta_bytes"] = len(fasta)
        return out

    runs = [phases() for _ in range(a.repeat)]
    best = min(runs, key=lambda r: r["decode_s"] + r["pileup_s"] + r["consensus_s"])
    t = time.perf_counter()
    res = K.bam_to_consensus(path)                      # the public call, reports included
    api_s = time.perf_counter() - t
    total = best["decode_s"] + best["pileup_s"] + best["consensus_s"]
    line = {
        "what": "end-to-end BAM path -> FASTA bytes (SURVEY 8d ii)", "config": a.config, "scale": a.scale,
        "reads": int(len(batch["contig"])), "aligned_events": ev, "bam_bytes": size,
        "host_cores": os.cpu_count(), "decode_threads": a.threads or os.cpu_count(),
        "best_of": a.repeat, **{k: round(v, 4) if isinstance(v, float) else v for k, v in best.items()},
        "total_s": round(total, 4), "events_per_s": ev / total, "reads_per_s_decode": len(batch["contig"]) / best["decode_s"],
        "bam_to_consensus_s": round(api_s, 4), "bam_to_consensus_events_per_s": ev / api_s,
        "n_consensus_records": len(res.consensuses), "synth_plus_write_s": round(t_make, 1),
    }
    s = json.dumps(line)
    print(s)
    if a.out:
        with open(a.out, "w") as fh:
            fh.write(s + "\n")
    os.unlink(path)


if __name__ == "__main__":
    main()
