#!/usr/bin/env python
"""Strong-scaling PROJECTION on one GPU: for N in {1, 2, 4, 8} the ONE config-sized input is cut into
N work-balanced position intervals exactly as `bench.py --scaling strong` does (shard.partition_weighted, reads routed by
their CIGAR footprints), and every rank's shard is timed ALONE on the one GPU, one after another.  On N GPUs the ranks run
side by side and the step ends with one all-gather, so the projected N-GPU step is  max_r t_r  (+ the all-gather, which is
not measured here: <= 5 MB over xGMI, latency bound).  This is a projection, labelled as such -- not a measurement on N GPUs.

    python scripts/strong_projection.py [--config C3] [--steps 10] [--out profiles/r03_strong_scaling_projection.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--out", default="")
    ap.add_argument("--only-rank", type=int, default=-1, help="time only this rank of every decomposition (sweeps)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel pass (hipEvents around every launch): for a dispatch timeline of the timed steps")
    ap.add_argument("--tunings", default="0:0", help="window:slice[,window:slice...] -- every rank's shard is timed under each, the best stands "
                    "(0:0 = the engine's own choice)")
    args = ap.parse_args()
    import torch
    from kindel_amd import _native as N
    from kindel_amd import shard
    from tools import synth

    if os.environ.get("KD_BENCH_LIB"):   # a differently built library (A/B of a variant, scripts/README.md): never the default
        N._default = N.Library(os.environ["KD_BENCH_LIB"])
    dev = "cuda:0"
    tb = synth.make(args.config, device=dev)
    lens = tb["contig_lens"]
    g_lo, g_hi = shard.footprints(lens, tb)
    aligned = synth.counts(tb)[1]
    n_all = int(tb["contig"].numel())
    pinned = torch.empty(sum(int(l) + int(l) // 8 for l in lens) + 4096, dtype=torch.uint8, pin_memory=True).numpy()
    rows = []
    for world in [int(x) for x in args.ranks.split(",")]:
        ivs = shard.partition_weighted(lens, tb["contig"], tb["pos0"], tb["seq_len"], world)
        per_rank = []
        for r in range(world):
            if args.only_rank >= 0 and r != min(args.only_rank, world - 1):
                continue
            keep = shard.reads_of_rank(lens, g_lo, g_hi, r, world, intervals=ivs)
            sub = dict(tb)
            for k in ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig"):
                sub[k] = tb[k][keep].contiguous()
            n = int(sub["contig"].numel())
            eng = N.Engine(lens, device=0)
            if world > 1:
                eng.set_shard(*ivs[r])
            ptrs = synth.device_ptrs(sub)


            # N > 1: the rank's exchange row registered with the engine, as bench.py --gpus N does (kd_set_exchange): the projected step
            # contains the row's copies and k_exchange_head -- everything of an N-GPU step but the collective itself
            ex = shard.Exchange(eng, ivs[r], dev, pad=shard.row_pad(eng, ivs[r], world, ivs)).attach() if world > 1 else None

            def step():
                eng.step_device(ptrs, n, tb["seq4_bytes"], tb["cigar_words"], pinned)
                if ex is not None:
                    ex.collect()

            ms, tuned, tried = None, None, {}
            for spec in args.tunings.split(","):
                w_, s_ = (int(x) for x in spec.split(":"))
                eng.set_tuning(w_, s_)
                for _ in range(args.warmup):
                    step()
                for _ in range(3):      # best of three timed blocks: one host hiccup must not pass for a rank's step time
                    eng.sync(); torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        step()
                    eng.sync(); torch.cuda.synchronize()
                    m = (time.perf_counter() - t0) / args.steps * 1e3
                    tried[spec] = round(min(tried.get(spec, 1e9), m), 4)
                    if ms is None or m < ms:
                        ms, tuned = m, spec
            w_, s_ = (int(x) for x in tuned.split(":"))
            eng.set_tuning(w_, s_)
            prof = {}
            if not args.no_profile:
                eng.profile_enable(1); eng.profile_reset()
                for _ in range(args.steps):
                    step()
                prof = eng.profile()
                eng.profile_enable(0)
            kern = sum(v[1] for v in prof.values()) / args.steps
            per_rank.append(dict(rank=r, interval=[int(ivs[r][0]), int(ivs[r][1])], reads=n, step_ms=round(ms, 4), kernel_ms=round(kern, 4), tuning=tuned, tried=tried,
                                 kernels={k: round(v[1] / max(v[0], 1), 4) for k, v in sorted(prof.items())},
                                 k_window_ms=round(prof.get("k_window", (0, 0.0))[1] / args.steps, 4), k_prep_ms=round(prof.get("k_prep", (0, 0.0))[1] / args.steps, 4)))
            eng.close()
            del sub, keep
            torch.cuda.empty_cache()
        worst = max(p["step_ms"] for p in per_rank)
        rows.append(dict(n_ranks=world, projected_step_ms=worst, sum_of_rank_steps_ms=round(sum(p["step_ms"] for p in per_rank), 4),
                         reads_seen_total=sum(p["reads"] for p in per_rank), per_rank=per_rank))
    t1 = rows[0]["projected_step_ms"]
    # the fixed part of a step: what does not shrink with the shard (dispatch chain, read-backs, the final copy): from the two ends
    # of the curve, t(N) = fixed + work / N
    out = dict(kind="PROJECTION from one GPU: each rank's shard of the strong-scaling decomposition timed alone; projected N-GPU step = "
                    "max over ranks (every rank writes its exchange row as bench.py --gpus N does; the one all-gather of <= 5 MB itself is not included); NOT a measurement on N GPUs",
               config=args.config, reads=n_all, aligned_events=aligned, steps=args.steps,
               rows=[dict(r, projected_speedup=round(t1 / r["projected_step_ms"], 3),
                          projected_events_per_s=aligned / (r["projected_step_ms"] * 1e-3)) for r in rows])
    if len(rows) > 1:
        nl, tl = rows[-1]["n_ranks"], rows[-1]["projected_step_ms"]
        fixed = (tl * nl - t1) / (nl - 1) if nl > 1 else 0.0
        out["fixed_ms_estimate"] = round(fixed, 4)
        out["fixed_note"] = "from t(1) and t(%d) under t(N) = fixed + work / N" % nl
    line = json.dumps(out)
    print(line)
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
