#!/usr/bin/env python
"""EXPERIMENT (round 6): K whole steps (kd_step: reset, record loop, insertion reduction, consensus, FASTA into pinned memory) over one
resident batch -- by ONE context, one step after the other (what bench.py times), and by D contexts on D host threads, each with its
own stream, tables and output buffer, taking the steps in turn: step k+1's k_prep / table fill run while step k's k_window and its
consensus tail (the FASTA leaves over the host link) are still under way.  Every step is complete and independent; every output is
compared with the serial run's.

    python scripts/exp/pipeline_ab.py [--config C3] [--steps 40] [--warmup 8] [--depths 1,2,3]
"""
import argparse, hashlib, json, os, sys, threading, time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402
from kindel_amd import _native as N  # noqa: E402
from tools import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--depths", default="1,2,3")
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    if os.environ.get("KD_BENCH_LIB"):
        N._default = N.Library(os.environ["KD_BENCH_LIB"])
    cfg = dict(synth.CONFIGS[args.config])
    batch = synth.make(cfg, device="cuda:0")
    torch.cuda.synchronize()
    contig_lens = cfg["contig_lens"]
    ptrs = synth.device_ptrs(batch)
    n_reads = int(batch["contig"].numel())
    cg = batch["cigar"][: batch["cigar_words"]].long()
    ln, op = cg >> 4, cg & 15
    aligned = int(ln[(op == 0) | (op == 7) | (op == 8)].sum())
    cap = sum(int(l) + int(l) // 8 for l in contig_lens) + 4096
    n_max = max(int(d) for d in args.depths.split(","))
    engs = [N.Engine(np.asarray(contig_lens, np.uint32), device=0) for _ in range(n_max)]
    outs = [torch.empty(cap, dtype=torch.uint8, pin_memory=True) for _ in range(n_max)]
    outs_np = [o.numpy() for o in outs]

    def one(j, check=True):
        off = engs[j].step_device(ptrs, n_reads, batch["seq4_bytes"], batch["cigar_words"], outs_np[j])
        return hashlib.sha256(outs_np[j][: int(off[-1])].tobytes()).hexdigest() if check else None      # (5 MB: 2 ms of host time -- never inside the timed loop)

    want = one(0)
    off0 = engs[0].step_device(ptrs, n_reads, batch["seq4_bytes"], batch["cigar_words"], outs_np[0])
    want_bytes = outs_np[0][: int(off0[-1])].tobytes()
    rows = []
    try:
        for rep in range(args.reps):
            for d in [int(x) for x in args.depths.split(",")]:
                shas = [None] * d
                start = threading.Barrier(d + 1)

                def worker(j, n_steps):
                    start.wait()
                    for _ in range(n_steps):
                        one(j, check=False)
                    shas[j] = True

                for j in range(d):
                    for _ in range(max(1, args.warmup // d)):
                        assert one(j) == want
                counts = [args.steps // d + (1 if j < args.steps % d else 0) for j in range(d)]
                ths = [threading.Thread(target=worker, args=(j, counts[j])) for j in range(d)]
                for t in ths:
                    t.start()
                torch.cuda.synchronize()
                start.wait()
                t0 = time.perf_counter()
                for t in ths:
                    t.join()
                for e in engs[:d]:
                    e.sync()
                dt = time.perf_counter() - t0
                assert all(shas)
                for j in range(d):      # what the last step of every context left in its buffer
                    n = len(want_bytes)
                    assert outs_np[j][:n].tobytes() == want_bytes, "a pipelined step's FASTA differs from the serial one"
                ms = dt / args.steps * 1e3
                rows.append({"depth": d, "rep": rep, "ms_per_step": ms, "events_per_s": aligned * args.steps / dt})
                print("%s depth %d rep %d  %.4f ms per step  %.4e events/s  sha %s" % (args.config, d, rep, ms, aligned * args.steps / dt, want[:10]), flush=True)
    finally:
        for e in engs:
            e.close()
    print(json.dumps({"config": args.config, "steps": args.steps, "rows": rows, "fasta_sha256": want}))


if __name__ == "__main__":
    main()
