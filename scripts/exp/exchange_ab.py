#!/usr/bin/env python
"""A/B of the exchange step's HOST side on one GPU (no collective: world 1): a 1/8 shard of the strong decomposition stepped
(a) alone, (b) with round 3's row assembly in torch behind every step (eight torch operations, two pageable uploads, a blocking
.item()), (c) with the row registered with the engine (kd_set_exchange: filled on the way of kd_step).  What an N-GPU step
adds on top of (c) is the all-gather itself.

    python scripts/exp/exchange_ab.py [--config C3] [--ranks 8] [--rank 3] [--out gpurun_out/exchange_ab.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def legacy_row(torch, shard, engine, interval, device, pad):
    """Round 3's gather() up to (not including) the collective, kept here for the A/B only."""
    lo, hi = interval
    coff, mm = engine.consensus_offsets()
    cptr, cbytes = engine.consensus_device()
    chptr = engine.changes_device()
    head = np.concatenate([coff.view(np.uint8), mm.reshape(-1).view(np.uint8)])
    my_size = shard._HDR + head.size + (hi - lo) + cbytes
    payload = torch.zeros(pad, dtype=torch.uint8, device=device)
    hdr = np.asarray([my_size, 0], np.uint64).view(np.uint8)
    payload[:shard._HDR] = torch.from_numpy(hdr).to(device)
    o = shard._HDR
    payload[o: o + head.size] = torch.from_numpy(head).to(device)
    o += head.size
    payload[o: o + (hi - lo)] = shard._as_tensor(chptr + lo, hi - lo, device)
    o += hi - lo
    payload[o: o + cbytes] = shard._as_tensor(cptr, cbytes, device)
    rows = payload.view(1, pad)
    return rows, int(rows[:, :8].contiguous().view(torch.int64).max().item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    from kindel_amd import _native as N
    from kindel_amd import shard
    from tools import synth

    dev = "cuda:0"
    tb = synth.make(args.config, device=dev)
    lens = tb["contig_lens"]
    g_lo, g_hi = shard.footprints(lens, tb)
    ivs = shard.partition_weighted(lens, tb["contig"], tb["pos0"], tb["seq_len"], args.ranks)
    r = args.rank
    keep = shard.reads_of_rank(lens, g_lo, g_hi, r, args.ranks, intervals=ivs)
    sub = dict(tb)
    for k in ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig"):
        sub[k] = tb[k][keep].contiguous()
    n = int(sub["contig"].numel())
    pinned = torch.empty(sum(int(l) + int(l) // 8 for l in lens) + 4096, dtype=torch.uint8, pin_memory=True).numpy()
    eng = N.Engine(lens, device=0)
    eng.set_shard(*ivs[r])
    ptrs = synth.device_ptrs(sub)
    pad = shard.row_pad(eng, ivs[r], args.ranks, ivs)

    def timed(fn):
        for _ in range(5):
            fn()
        best = 1e9
        for _ in range(3):
            eng.sync(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            eng.sync(); torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / args.steps * 1e3)
        return round(best, 4)

    def step():
        eng.step_device(ptrs, n, tb["seq4_bytes"], tb["cigar_words"], pinned)

    def step_legacy():
        step()
        return legacy_row(torch, shard, eng, ivs[r], dev, pad)

    res = dict(config=args.config, ranks=args.ranks, rank=r, reads=n, row_bytes_agreed=pad)
    res["step_alone_ms"] = timed(step)
    res["step_plus_torch_row_r3_ms"] = timed(step_legacy)
    rows_legacy, need = step_legacy()
    ex = shard.Exchange(eng, ivs[r], dev, pad=pad).attach()

    def step_attached():
        step()
        return ex.collect()

    res["step_with_registered_row_ms"] = timed(step_attached)
    rows_new = step_attached()
    res["rows_equal"] = bool((rows_new[0, :need] == rows_legacy[0, :need]).all()) and ex.need(rows_new) == need
    res["row_bytes"] = need
    ex.detach()
    eng.close()
    print(json.dumps(res))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
