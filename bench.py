#!/usr/bin/env python
"""bench.py -- aligned-base events/s of pileup + consensus on synthetic alignments (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3] [--scale 1.0]

One "step" = one full pass of the hot path over the whole synthetic batch, inputs already
resident in HBM: zero the tables, record loop (k_prep / k_plan / k_window / k_cold_lane),
insertion multiset reduction, per-site consensus, and the rank's consensus bytes copied to pinned host
memory; at N > 1 additionally the all-gather that leaves the stitched consensus of ALL ranks in every
GPU's HBM (assembling that into one host FASTA is file output, done once outside the timed region).  N > 1: one process per GPU (torchrun), reference
positions sharded by G-space interval with no collective on the pileup path.  N > 1 reports STRONG scaling as `value` (the
north star's figure: ONE config-sized input cut into N work-balanced position intervals, total work fixed) and measures weak
scaling (N copies of the config's contig set, one per rank) right after, as `other_scaling`; `--scaling weak` swaps them.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant
kernel (per-launch durations from hipEvents on the engine's stream) and, at N = 1,
`cpu_baseline`: the UNMODIFIED reference (kindel.kindel.parse_records + consensus_sequence) timed in this run on one host
core over a bounded sample of the same workload; nested in it (`c_port`) the C oracle (oracle/kindel_oracle.c, a port of the
reference's loops) over the whole batch, which doubles as the full-size bit-exactness check.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)


def algorithmic_bytes(n_reads, query_bases, cigar_ops, sites):
    """B = Q/2 + 4*O + 28*R + 144*S + 40*S"""
    return query_bases / 2.0 + 4.0 * cigar_ops + 28.0 * n_reads + 184.0 * sites


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)      # (the round-end driver's own choice: --steps 20 --warmup 5; a step is 1.5 ms)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C3", help="C2 | C3 | C4 | C5 (SURVEY.md section 8d)")
    ap.add_argument("--synth", action="append", default=[], metavar="KEY=VALUE",
                    help="override a synthetic-generator parameter (sensitivity runs; not the BASELINE workload)")
    ap.add_argument("--scale", type=float, default=1.0, help="depth multiplier (1.0 = the config as specified)")
    ap.add_argument("--mode", default="auto", choices=["auto", "global", "window", "strip", "coop"])
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--slice", type=int, default=0)
    ap.add_argument("--sweep", default="", help="extra tunings to time on the same batch: mode:window:slice,...")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; "
                    "gloo only to exercise the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--scaling", default="", choices=["", "weak", "strong"],
                    help="N > 1: weak = N copies of the config's contig set, one per rank (per-GPU work fixed); strong = ONE "
                         "config-sized input, work-balanced contiguous position intervals (whole contigs where possible), every "
                         "rank takes its reads from the one shared batch (total work fixed)")
    ap.add_argument("--one-scaling", action="store_true", help="N > 1: do not also measure the other scaling rule")
    ap.add_argument("--shuffle", nargs="?", const="records", default="", choices=["records", "index"],
                    help="permute the read order (unsorted input: exercises the device bucket sort).  records (default): the batch a decoder "
                         "hands over for an unsorted FILE -- bases and CIGAR words lie in record order too; index: only the per-read arrays "
                         "are permuted, every read's bases / CIGAR stay where the sorted batch had them (rounds 1 - 2; a layout no file produces)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline-leg", action="store_true",
                    help="an EXTRA leg behind the measurement (never the headline, off by default so that a profiler's per-kernel averages of the default "
                         "command stay those of the timed steps): the same steps taken in turn by two contexts on two host threads")
    ap.add_argument("--cpu-sample", type=float, default=0.0, help="fraction of reads for the CPU baseline (0 = auto)")
    ap.add_argument("--rccl-at-1", action="store_true",
                    help="N = 1 only: initialise torch.distributed with backend nccl (= RCCL) at world size 1, register the exchange row and run the step's "
                         "all-gather on it every step -- the N-GPU code path (kd_set_exchange + ncclAllGather on device rows) on the one GPU a gpurun box "
                         "has; the FASTA of the line is the one assembled from the gathered row")
    ap.add_argument("--no-graph", action="store_true", help="(accepted and ignored: older command lines; the hipGraph replay of rounds 3 - 5 is gone)")
    ap.add_argument("--e2e-scale", type=float, default=0.1, help="N = 1: depth scale of the live end-to-end leg (BAM file -> FASTA; 0 = skip)")
    args = ap.parse_args()
    if not args.scaling:      # N > 1: strong scaling is the headline (BASELINE.json: one input over 1 / 2 / 4 / 8 GPUs); N = 1: the contract's word
        args.scaling = "strong" if args.gpus > 1 else "weak"

    import torch
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N` (no torchrun): spawn the N ranks ourselves, one process per GPU, and hand their
        # output / exit status through -- never fall through to one rank that reports a 1-GPU number under an N-GPU request
        sys.exit(_spawn_ranks(args, torch))
    import torch.distributed as dist
    from kindel_amd import _native as N
    from kindel_amd import shard
    from tools import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = local_rank % max(1, torch.cuda.device_count())   # == local_rank on a node with >= N GPUs
    coll = world > 1 or args.rccl_at_1      # the step ends with the exchange collective
    if args.rccl_at_1 and (world != 1 or args.backend != "nccl"):
        raise SystemExit("bench.py: --rccl-at-1 is for N = 1 with the nccl backend")
    if coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch as `python bench.py --gpus N` (spawns its own ranks) or under "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, world))
    if args.backend == "nccl" and torch.cuda.device_count() < (world if world > 1 else 1):
        raise SystemExit("bench.py: %d rank(s) over RCCL need %d visible GPU(s), found %d" % (world, world, torch.cuda.device_count()))
    torch.cuda.set_device(dev_index)
    dev = "cuda:%d" % dev_index

    if os.environ.get("KD_BENCH_LIB"):   # profiling builds (e.g. hipcc -DKD_PHASE_CLOCKS): a differently built library, never the default
        from kindel_amd import _native as _N
        _N._default = _N.Library(os.environ["KD_BENCH_LIB"])
    def run(scaling):
        """One complete measurement (generation, warm-up, K timed steps, per-kernel pass) under one scaling rule."""
        cfg = dict(synth.CONFIGS[args.config])
        for kv in args.synth:            # sensitivity runs only, e.g. --synth clip_p=0 --synth indel_p=0 --synth planted=0
            k, v = kv.split("=")
            cfg[k] = float(v) if "." in v else int(v)
        if world > 1 and cfg["kind"] != "short":
            raise SystemExit("multi-GPU bench: short-read configs only (C2, C3, C4)")
        strong = world > 1 and scaling == "strong"
        if not strong:
            cfg["contig_lens"] = list(cfg["contig_lens"]) * world  # weak scaling: one config-sized interval per rank
        cfg["depth"] = cfg["depth"] * args.scale
        contig_lens = cfg["contig_lens"]
        t0 = time.time()
        intervals = None
        if strong:
            # one shared input (the same seeded batch on every rank, standing in for one decoded file), split by work
            batch = synth.make(cfg, device=dev)
            intervals = shard.partition_weighted(contig_lens, batch["contig"], batch["pos0"], batch["seq_len"], world)
            g_lo, g_hi = shard.footprints(contig_lens, batch)      # exact reach of every read, from its CIGAR (deletions included)
            keep = shard.reads_of_rank(contig_lens, g_lo, g_hi, rank, world, intervals=intervals)
            del g_lo, g_hi
            full_cigar, full_words = batch["cigar"], batch["cigar_words"]
            full_ncig = batch["n_cig"]
            for k in ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig"):
                batch[k] = batch[k][keep].contiguous()     # routed: this rank's reads (bases / CIGAR words stay where they are)
        else:
            batch = synth.make(cfg, device=dev, shard=(rank, world) if world > 1 else None)
        torch.cuda.synchronize()
        t_gen = time.time() - t0
        n_reads = int(batch["contig"].numel())
        if args.shuffle:
            batch = synth.shuffled(batch, mode=args.shuffle)
        # events are credited to the rank that owns the read's start, so every read counts once
        if strong:      # every rank holds the whole CIGAR array: the totals of the ONE input, no reduction needed
            cg = full_cigar[: full_words].long()
            n_owned = int(full_ncig.numel())
        elif world > 1:
            own = shard.owned_mask(contig_lens, batch["contig"], batch["pos0"], rank, world)
            cg_owner = torch.repeat_interleave(own, batch["n_cig"].long())
            cg = batch["cigar"][: batch["cigar_words"]].long()[cg_owner]
            n_owned = int(own.sum())
        else:
            cg = batch["cigar"][: batch["cigar_words"]].long()
            n_owned = n_reads
        ln, op = cg >> 4, cg & 15
        aligned = int(ln[(op == 0) | (op == 7) | (op == 8)].sum())
        query = aligned + int(ln[(op == 1) | (op == 4)].sum())
        walked = query + int(ln[op == 2].sum())
        n_ops = int(cg.numel())
        tot = torch.tensor([aligned, query, walked, n_ops, n_owned], dtype=torch.int64, device=dev)
        if world > 1 and not strong:
            dist.all_reduce(tot)
        aligned_g, query_g, walked_g, ops_g, reads_g = (int(x) for x in tot.cpu())

        mode = {"auto": N.KD_MODE_AUTO, "window": N.KD_MODE_WINDOW, "global": N.KD_MODE_GLOBAL, "strip": N.KD_MODE_STRIP, "coop": N.KD_MODE_COOP}[args.mode]
        eng = N.Engine(np.asarray(contig_lens, np.uint32), device=dev_index, mode=mode)
        if args.window or args.slice:
            eng.set_tuning(args.window, args.slice)
        if coll and intervals is None:
            intervals = shard.partition(contig_lens, world)
        interval = intervals[rank] if world > 1 else (0, eng.total_sites())
        ptrs = synth.device_ptrs(batch)
        n_contigs = len(contig_lens)

        if world > 1:
            eng.set_shard(*interval)

        # one pinned host buffer for the consensus bytes of all contigs: a single D2H copy, no pageable staging
        pinned = torch.empty(sum(int(l) + int(l) // 8 for l in contig_lens) + 4096, dtype=torch.uint8, pin_memory=True)
        pinned_np = pinned.numpy()

        state = {}
        # N > 1: the exchange row is registered with the engine (kd_set_exchange): kd_step leaves this rank's row -- header, contig
        # offsets, depth ranges, change codes, consensus bytes -- in device memory on its way, the step's collective is all that follows
        exch = shard.Exchange(eng, interval, dev, intervals=intervals, force_collective=args.rccl_at_1).attach() if coll else None

        def step(classic=False):
            if classic:
                # the call sequence of rounds 1 - 3, five blocking read-backs (kept as a second, independent way to the same bytes)
                eng.reset()
                eng.push_device(ptrs, n_reads, batch["seq4_bytes"], batch["cigar_words"])
                eng.finalize()
                eng.consensus_run(1)
                off = eng.consensus_fetch_all_into(pinned_np)
            else:
                # ONE call (kd_step): reset, record loop, insertion reduction, consensus and this rank's consensus bytes into pinned
                # host memory (at N = 1 the whole FASTA), two host round trips
                off = eng.step_device(ptrs, n_reads, batch["seq4_bytes"], batch["cigar_words"], pinned_np)
            seqs = [pinned_np[int(off[c]): int(off[c + 1])] for c in range(n_contigs)]
            # ... and, at N > 1, the all-gather that leaves the stitched consensus in every GPU's HBM
            if coll:   # one fixed-size all-gather (RCCL over xGMI); the classic call sequence does not fill the row: on demand there
                state["gathered"] = exch.run() if classic else exch.collect()
            return seqs

        def barrier():
            if world > 1:
                dist.barrier()
            eng.sync()
            torch.cuda.synchronize()

        for _ in range(args.warmup):
            seqs = step()
        # timed region (eager submission): hipEvents only around the dominant kernel's launches (mode 2; two events per step,
        # for the roofline).  Events around EVERY launch cost a few microseconds each, so the per-kernel table comes from an
        # extra, untimed pass of the same steps afterwards.
        eng.profile_enable(2)
        eng.profile_reset()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            seqs = step()
        barrier()
        dt = time.perf_counter() - t0
        prof_dom = eng.profile()
        dt_eager, submission = dt, "eager (kd_step): one dispatch per kernel, two host round trips per step"
        seqs_eager = seqs
        # the classic five-call sequence once (untimed): the same bytes by the other way
        seqs_c = [bytes(memoryview(x)) for x in step(classic=True)]
        assert seqs_c == [bytes(memoryview(x)) for x in step()], "kd_step and the classic call sequence disagree"
        eng.profile_enable(1)
        eng.profile_reset()
        for _ in range(args.steps):
            step()
        barrier()
        prof = eng.profile()
        eng.profile_enable(0)
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0].item())
        info = eng.batch_info()
        stats = eng.stats()
        # EXTRA LEG, never the headline: the same K steps taken in turn by TWO contexts on two host threads (each its own stream, tables
        # and pinned output) -- independent steps pipelined: one step's consensus tail (the FASTA leaving over the host link) under the
        # next one's k_prep.  A throughput figure for a stream of batches; `value` / `ms_per_step` above stay ONE context, one step after
        # the other.  (scripts/exp/pipeline_ab.py has depth 3 too.)
        pipelined = None
        if world == 1 and args.pipeline_leg and not args.shuffle:
            try:
                import threading
                eng2 = N.Engine(np.asarray(contig_lens, np.uint32), device=dev_index, mode=mode)
                if args.window or args.slice:
                    eng2.set_tuning(args.window, args.slice)
                pinned2 = torch.empty_like(pinned, pin_memory=True).numpy()
                pair = ((eng, pinned_np), (eng2, pinned2))
                try:
                    for e_, o_ in pair:
                        for _ in range(2):
                            off2 = e_.step_device(ptrs, n_reads, batch["seq4_bytes"], batch["cigar_words"], o_)
                    go = threading.Barrier(3)

                    def worker(j, n_steps):
                        go.wait()
                        for _ in range(n_steps):
                            pair[j][0].step_device(ptrs, n_reads, batch["seq4_bytes"], batch["cigar_words"], pair[j][1])

                    ths = [threading.Thread(target=worker, args=(j, args.steps // 2 + (j < args.steps % 2))) for j in range(2)]
                    for t_ in ths:
                        t_.start()
                    barrier()
                    go.wait()
                    tp0 = time.perf_counter()
                    for t_ in ths:
                        t_.join()
                    eng.sync(); eng2.sync()
                    dtp = time.perf_counter() - tp0
                    n_out = int(off2[-1])
                    same = bytes(memoryview(pinned_np[:n_out])) == bytes(memoryview(pinned2[:n_out])) == b"".join(seqs_c)
                    pipelined = dict(contexts=2, steps=args.steps, ms_per_step=round(dtp / args.steps * 1e3, 4), events_per_s=aligned_g / (dtp / args.steps),
                                     same_fasta=bool(same), note="extra leg, not `value`: independent steps taken in turn by two contexts on two host threads")
                finally:
                    eng2.close()
            except Exception as e:      # (the extra leg must never cost the bench line)
                pipelined = dict(error=repr(e)[:200])
        if coll:   # host assembly of the stitched FASTA, outside the timed region (checksum only)
            assert exch.need(state["gathered"]) <= exch.pad, "an exchange row did not fit its agreed size"
            rows = np.ascontiguousarray(state["gathered"].cpu().numpy())
            seqs, _, _ = shard.assemble(rows, contig_lens, world, intervals=intervals)
        seqs = [bytes(memoryview(x)) for x in seqs]
        fasta_sha = hashlib.sha256(b"\n".join(seqs)).hexdigest()

        out = None
        if rank == 0:
            ms_step = dt / args.steps * 1e3
            value = aligned_g / (dt / args.steps)
            sites = int(sum(contig_lens))
            B = algorithmic_bytes(reads_g, query_g, ops_g, sites)
            rows = {k: (n, ms / max(n, 1)) for k, (n, ms) in prof.items()}
            dom = max(rows.items(), key=lambda kv: kv[1][0] * kv[1][1])[0] if rows else None
            kernel_ms_per_step = sum(n * avg for n, avg in rows.values()) / args.steps
            if dom in prof_dom:   # the dominant kernel's duration as measured INSIDE the timed region
                rows[dom] = (prof_dom[dom][0], prof_dom[dom][1] / max(prof_dom[dom][0], 1))
            roofline = None
            if dom:
                # per-launch algorithmic bytes of the whole path (SURVEY 8d figure x events of one launch; at N > 1
                # one launch sees 1/N of them) over the dominant kernel's average launch duration
                launches_per_step = rows[dom][0] / args.steps
                a = B / world / max(launches_per_step, 1) / (rows[dom][1] * 1e-3) / 1e9
                roofline = dict(bound="hbm", kernel=dom, achieved=round(a, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                                frac=round(a / HBM_PEAK_GBS, 5), traffic=_pmc_traffic(dom, args, world), traffic_source=_PMC_SOURCE,
                                avg_launch_ms=round(rows[dom][1], 4),
                                algorithmic_bytes=int(B), bytes_per_event=round(B / max(aligned_g, 1), 4),
                                step_achieved=round(B / (dt / args.steps) / 1e9, 2),
                                step_frac=round(B / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 5))
            out = dict(
                metric="aligned-base events/sec pileup+consensus", value=value, unit="events/s", n_gpus=world,
                **(dict(rccl_at_1="every step ended with ncclAllGather (RCCL, world size 1) on the engine-written exchange row; this line's FASTA is assembled from the gathered row") if args.rccl_at_1 else {}),
                steps=args.steps, warmup=args.warmup, ms_per_step=ms_step, higher_is_better=True, submission=submission,
                scaling=scaling if world > 1 else "weak", vs_baseline=None, dtype="u32", data="synthetic",
                config=dict(workload="%s: synthetic %s, %d contig(s), %d sites, depth %gx%s" % (
                    args.config, "150 bp short reads" if cfg["kind"] == "short" else "ONT-like long reads",
                    n_contigs, sites, cfg["depth"], ("" if args.scale == 1.0 else " (scaled)") +
                    (" [generator overrides: %s]" % ",".join(args.synth) if args.synth else "")),
                    reads=reads_g, aligned_events=aligned_g, walked_events=walked_g, cigar_ops=ops_g,
                    parallelism=("one input, %d work-balanced position intervals (reads routed from the shared batch)" % world if strong
                                 else "interval-sharded x%d (one config-sized interval per rank)" % world) if world > 1 else "single GPU",
                    pileup_path="window-lds" if info["windowed"] else "global-atomics",
                    window_sites=eng.tuning()[0], work_items=info["work_items"]),
                roofline=roofline,
                kernels={k: dict(launches_per_step=n / args.steps, avg_ms=round(avg, 4)) for k, (n, avg) in sorted(rows.items())},
                kernels_source="hipEvents per launch: %s inside the timed region of the eager submission, the others in an extra untimed pass of the same %d steps" % (
                    "/".join(sorted(prof_dom)) or "none", args.steps),
                kernel_ms_per_step=round(kernel_ms_per_step, 4),
                fasta_sha256=fasta_sha, consensus_len=sum(len(s) for s in seqs), gen_seconds=round(t_gen, 1),
                library_sha256=_library_sha(N),
                engine_stats=stats,
                **(dict(pipelined=pipelined) if pipelined else {}),
            )
        return dict(out=out, eng=eng, batch=batch, contig_lens=contig_lens, seqs=seqs, aligned_g=aligned_g, step=step, barrier=barrier)

    res = run(args.scaling)
    out, eng, batch, contig_lens, seqs, aligned_g, step, barrier = (res[k] for k in ("out", "eng", "batch", "contig_lens", "seqs", "aligned_g", "step", "barrier"))
    if world > 1 and not args.one_scaling:
        # the other scaling rule, measured the same way right after (its own generation, warm-up and K timed steps); reported
        # as a secondary object of the same line -- `value` / `scaling` above stay the contract's figures
        eng.close()
        del res, batch
        torch.cuda.empty_cache()
        other = run("strong" if args.scaling == "weak" else "weak")
        eng, batch, contig_lens, seqs, aligned_g, step, barrier = (other[k] for k in ("eng", "batch", "contig_lens", "seqs", "aligned_g", "step", "barrier"))
        if rank == 0:
            o = other["out"]
            out["other_scaling"] = {k: o[k] for k in ("scaling", "value", "unit", "ms_per_step", "config", "roofline", "fasta_sha256")}
    if args.sweep and world == 1:
        # tuning sweep on the resident batch: "mode:window:slice,..." -> one JSON line each on stderr
        for spec in args.sweep.split(","):
            m, w, s = (spec.split(":") + ["0", "0"])[:3]
            eng.set_mode({"auto": N.KD_MODE_AUTO, "window": N.KD_MODE_WINDOW, "global": N.KD_MODE_GLOBAL, "strip": N.KD_MODE_STRIP, "coop": N.KD_MODE_COOP}[m])
            eng.set_tuning(int(w), int(s))
            step()
            eng.profile_enable(1)
            eng.profile_reset()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            barrier()
            d = (time.perf_counter() - t0) / args.steps
            pr = eng.profile()
            eng.profile_enable(0)
            print(json.dumps(dict(sweep=spec, ms_per_step=round(d * 1e3, 4), events_per_s=aligned_g / d,
                                  items=eng.batch_info()["work_items"],
                                  kernels={k: round(ms / max(n, 1), 4) for k, (n, ms) in sorted(pr.items())})),
                  file=sys.stderr, flush=True)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(batch, contig_lens, seqs, args.cpu_sample, aligned_g)
    if rank == 0:
        if world == 1 and args.e2e_scale > 0 and not args.no_cpu_baseline:
            # SURVEY 8d (ii), LIVE: BAM path -> FASTA bytes on a depth-scaled copy of the workload (this box's host cores decode)
            eng.close()
            del batch
            torch.cuda.empty_cache()
            try:
                out["e2e"] = e2e_leg(args.config, args.e2e_scale)
            except Exception as e:      # the bench line must not die with its secondary leg
                out["e2e"] = dict(error=repr(e))
            eng = None
        # the full-size figure: the sequencer-like file (Phred qualities, 2.2 x compression), both ingests (STATIC: scripts/e2e_bench.py on an
        # MI355X box, copied into profiles/ -- the run that measured it is named in the record)
        e2e = os.path.join(ROOT, "profiles", "e2e_c3_full_phred.json")
        if args.config == "C3" and os.path.exists(e2e) and out is not None:
            try:
                d = json.load(open(e2e))
                out.setdefault("e2e", {})["full_size_static"] = dict(
                    source="profiles/e2e_c3_full_phred.json: STATIC, measured by scripts/e2e_bench.py on an MI355X box, not in this run",
                    qualities=d.get("qualities", "phred"), bam_bytes=d["bam_bytes"], decode_threads=d["decode_threads"], host_cpu_quota=d.get("host_cpu_quota"),
                    host_decode=dict(events_per_s=d["streamed_events_per_s"], seconds=d["streamed"]["total_s"], whole_file_events_per_s=d["whole_file_events_per_s"]),
                    device_side_ingest=(dict(events_per_s=d["gpu_ingest_events_per_s"], seconds=d["gpu_ingest"]["total_s"], same_fasta=d.get("gpu_ingest_same_fasta"))
                                        if "gpu_ingest" in d else None))
            except Exception:
                pass
        print(json.dumps(out))
    if eng is not None:
        eng.close()
    if coll:
        dist.destroy_process_group()
    return


def e2e_leg(config, scale):
    """SURVEY 8d (ii) measured in this run: the config at `scale` x depth written as a BAM file with Phred-like qualities (so that
    it compresses like sequencer output, not 15 x), then  file path -> host decode (streamed, this box's cores) -> copies ->
    kernels -> FASTA bytes  through the product's Python entry points, best of 3; the FASTA is compared with the oracle's
    consensus of the same reads.  Never `value`: the host decode bounds it."""
    import tempfile
    import torch
    from kindel_amd import _native as N
    from kindel_amd import kindel as K
    from tools import synth
    from oracle import oracle as ko
    tb = synth.make(config, scale=scale, device="cuda:0")
    aligned = synth.counts(tb)[1]
    host = synth.to_numpy(tb)
    del tb
    torch.cuda.empty_cache()
    path = os.path.join(tempfile.gettempdir(), "kd_bench_e2e_%s_%g.bam" % (config, scale))
    os.environ["KD_WRITE_BAM_QUAL"] = "phred"
    try:
        N.write_bam(path, host)
    finally:
        os.environ.pop("KD_WRITE_BAM_QUAL", None)
    size = os.path.getsize(path)
    raw = int(sum(36 + 2 + 4 * host["n_cig"].astype(np.int64) + (host["seq_len"].astype(np.int64) + 1) // 2 + host["seq_len"].astype(np.int64)))
    best, fasta = None, None
    for _ in range(3):
        t0 = time.perf_counter()
        pl = K.pileup_file(path, stream=True)
        t1 = time.perf_counter()
        done = K._device_consensus_all(pl, {c: None for c in pl.order}, False, 1, False)
        fasta = {pl.names[c]: done[c][0] for c in pl.order}
        t2 = time.perf_counter()
        r = dict(total_s=t2 - t0, ingest_s=t1 - t0, consensus_s=t2 - t1, decode_s=pl.ingest["decode_s"], push_s=pl.ingest["push_s"], batches=pl.ingest["batches"])
        pl.engine.close()
        if best is None or r["total_s"] < best["total_s"]:
            best = r
    # the same file through the DEVICE-side ingest (opt-in product path, kd_push_bam_gpu: BGZF inflate + BAM record walk on the GPU)
    gpu = None
    try:
        for _ in range(3):
            t0 = time.perf_counter()
            pl = K.pileup_file(path, ingest="gpu")
            t1 = time.perf_counter()
            done = K._device_consensus_all(pl, {c: None for c in pl.order}, False, 1, False)
            g_fasta = {pl.names[c]: done[c][0] for c in pl.order}
            t2 = time.perf_counter()
            r = dict(seconds=round(t2 - t0, 4), ingest_s=round(t1 - t0, 4), consensus_s=round(t2 - t1, 4), path=pl.ingest.get("path", "host"),
                     same_fasta_as_host_decode=bool(g_fasta == fasta))
            pl.engine.close()
            if gpu is None or r["seconds"] < gpu["seconds"]:
                gpu = r
        gpu["events_per_s"] = aligned / gpu["seconds"]
    except Exception as e:
        gpu = dict(error=repr(e))
    os.unlink(path)
    same = all(fasta["ctg%d" % c] == ko.parse_records(host, c).consensus_sequence()[0] for c in ko.contig_order(host))
    return dict(device_side_ingest=gpu,
                what="LIVE in this run: BAM file (Phred-like qualities) -> streamed host decode -> HIP pileup + consensus -> FASTA, best of 3",
                config=config, depth_scale=scale, reads=int(len(host["contig"])), aligned_events=aligned, events_per_s=aligned / best["total_s"],
                seconds=round(best["total_s"], 4), host_decode_s=round(best["decode_s"], 4), push_s=round(best["push_s"], 4),
                consensus_s=round(best["consensus_s"], 4), batches=best["batches"], bam_bytes=size, bam_compression=round(raw / max(size, 1), 2),
                decode_threads=N.host_threads(), host_cores=os.cpu_count(), host_cpu_quota=_cpu_quota(), bit_exact_vs_oracle=bool(same))


def _spawn_ranks(args, torch):
    """`python bench.py --gpus N` without a launcher: re-execute this command under torch.distributed.run with N ranks on
    127.0.0.1 (one process per GPU, RCCL).  Refuses when fewer than N GPUs are visible (unless --backend gloo asks for ranks
    that share GPUs, which is a functional test of the N-rank path, not a scaling measurement)."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and n_dev < args.gpus:
        print("bench.py: --gpus %d requested but only %d GPU(s) visible; refusing to report a %d-GPU number from fewer devices"
              % (args.gpus, n_dev, args.gpus), file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


_PMC_SOURCE = ("profiles/pmc_traffic.json: STATIC, not measured in this run -- (FETCH_SIZE x k + WRITE_SIZE) x 1024 per launch from "
               "the committed `rocprofv3 --pmc` passes of the same command (scripts/gpu_pmc.sh), k = the FETCH_SIZE factor "
               "calibrated on this access pattern (profiles/fetch_calibration.json)")


def _cpu_quota():
    """CPUs the cgroup grants per scheduling period (None = no limit): the GPU boxes show 256 cores and grant 16."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(p), 2)
    except Exception:
        return None


def _pmc_key(args, world):
    """The workload a committed PMC record belongs to: config, kernel mode, input order, scale, ranks."""
    return "%s|%s|%s" % (args.config, args.mode, ("shuffle-" + args.shuffle) if args.shuffle else "sorted")


def _library_sha(N):
    """sha256 (16 hex digits) of the libkindel_hip.so this run loaded: records of different builds must not be mixed
    (scripts/harvest_profiles.py refuses a counter pass whose library differs from the bench file's)"""
    try:
        with open(N.default_library().path, "rb") as fh:
            return hashlib.sha256(fh.read()).hexdigest()[:16]
    except Exception:
        return None


def _pmc_traffic(kernel, args, world):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of THIS workload (profiles/pmc_traffic.json:
    {"<config>|<mode>|<order>": {kernel: {"bytes": ...}}}), or None when no record of this workload exists (another config, a
    scaled / overridden generator, N > 1): a figure measured on another workload is not this one's traffic."""
    if world != 1 or args.scale != 1.0 or args.synth or args.window or args.slice:
        return None
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(p):
        try:
            return ((json.load(open(p)).get(_pmc_key(args, world)) or {}).get(kernel) or {}).get("bytes")
        except Exception:
            return None
    return None


def cpu_baseline(batch, contig_lens, gpu_seqs, sample, aligned_total):
    """Two CPU figures on this box's host cores, both single-threaded (the reference is):
    `reference_python`: the UNMODIFIED reference (kindel.kindel.parse_records + consensus_sequence) on a bounded sample of
    the same workload -- measured live, in this run: from /root/reference where that exists (the build container), from the
    bytecode oracle/make_ref.py compiled out of it into oracle/_ref/ elsewhere (the GPU box); only if neither is there is the
    committed figure of profiles/reference_python_baseline.json quoted, labelled as such;  value/kind "port": the C oracle (oracle/kindel_oracle.c, a statement-by-statement
    port of those loops) over the same batch, bounded to ~10-30 s by sub-sampling reads when the batch is large -- with the
    full batch it is also the full-size bit-exactness check of the GPU consensus."""
    from tools import synth
    from oracle import oracle as ko
    host = synth.to_numpy(batch)
    refpy = None
    try:
        from oracle import refbaseline, refrun
        if refrun.reference_available():
            refpy = refbaseline.time_reference(host, 0, 4.0e7)
            refpy["where"] = "this run, this box (%d host cores, cgroup quota %s)" % (os.cpu_count(), _cpu_quota())
            refpy["origin"] = ("unmodified reference, imported from /root/reference" if refrun.origin() == "source" else
                               "unmodified reference, sourceless bytecode oracle/_ref/ compiled from /root/reference by oracle/make_ref.py")
    except Exception as e:   # the reference tree is test infrastructure of the build container only
        refpy = dict(error=repr(e))
    if refpy is None:
        p = os.path.join(ROOT, "profiles", "reference_python_baseline.json")
        if os.path.exists(p):
            refpy = json.load(open(p))
            refpy["note"] = "NOT measured in this run: /root/reference does not exist on this box; figure committed from the build container"
    n = len(host["contig"])
    frac = sample if sample > 0 else min(1.0, 4.0e9 / max(aligned_total, 1))  # the oracle walks ~7e8 events/s
    if int(round(1.0 / frac)) <= 1:
        frac = 1.0           # "every 1st read" is the whole batch: keep the full-size bit-exactness check
    if frac < 1.0:
        keep = np.arange(n) % max(1, int(round(1.0 / frac))) == 0
        for k in ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig"):
            host[k] = host[k][keep]
    t0 = time.perf_counter()
    ev = 0
    same = True
    for cid in range(len(contig_lens)):
        oa = ko.parse_records(host, cid)
        seq, _ = oa.consensus_sequence()
        ev += oa.n_events_aligned
        if frac >= 1.0:
            same = same and (seq.encode() == gpu_seqs[cid])
    dt = time.perf_counter() - t0
    from kindel_amd import _native as _N
    port = dict(value=ev / dt, unit="events/s", cores=1, kind="port",
                sample="%s of the %d reads (%d aligned-base events), all contigs, pileup + consensus; %.1f s" % (
                    "all" if frac >= 1.0 else "every %d-th" % int(round(1.0 / frac)), n, ev, dt),
                bit_exact_vs_gpu=(same if frac >= 1.0 else None))
    host_info = dict(host_cores=os.cpu_count(), host_cpu_quota=_cpu_quota(), decoder_threads_default=_N.host_threads())
    if refpy and "value" in refpy:
        # the contract's baseline: the reference's own CPU path, timed beside the GPU number; the C port (the oracle) nested
        out = dict(refpy)
        out.update(host_info)
        out["c_port"] = port
        out["bit_exact_vs_gpu"] = port["bit_exact_vs_gpu"]     # (the full-size check is the port's: the Python reference walks a sample)
        return out
    port.update(host_info)
    port["reference_python"] = refpy
    return port


if __name__ == "__main__":
    main()
