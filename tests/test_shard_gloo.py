"""N > 1 path on CPU: two processes, gloo backend, each rank owns a G-space interval (emulated kernels),
the all-gather stitch must reproduce the oracle's consensus for every contig on every rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, emu_path, lens, seed, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kindel_amd import _native as N
    from kindel_amd import shard, synth
    lib = N.Library(emu_path)
    tb = synth.short_reads(lens, 10, seed=seed, shard=(rank, world))   # each rank synthesises its own interval
    batch = synth.to_numpy(tb)
    iv = shard.partition(lens, world)[rank]
    eng = N.Engine(np.asarray(lens, np.uint32), lib=lib)
    eng.set_tuning(256, 0)
    eng.set_shard(*iv)
    eng.push(batch)
    eng.finalize()
    eng.consensus_run(1)
    seqs, changes, minmax = shard.stitch(eng, iv, "cpu")
    own = int(shard.owned_mask(lens, batch["contig"], batch["pos0"], rank, world).sum())
    q.put((rank, seqs, [c.tobytes() for c in changes], minmax, own))
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


def _worker_routed(rank, world, port, emu_path, lens, depth, seed, pad, q):
    """Strong-scaling shape: ONE batch (the same on every rank, like one decoded file), work-balanced contiguous
    intervals (whole contigs where possible), every rank takes the reads that touch its interval."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kindel_amd import _native as N
    from kindel_amd import shard, synth
    lib = N.Library(emu_path)
    full = synth.to_numpy(synth.short_reads(lens, depth, seed=seed))
    ivs = shard.partition_weighted(lens, full["contig"], full["pos0"], full["seq_len"], world)
    keep = shard.reads_of_rank(lens, *shard.footprints(lens, full), rank, world, intervals=ivs)
    sub = dict(full)
    for k in ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig"):
        sub[k] = full[k][keep]
    eng = N.Engine(np.asarray(lens, np.uint32), lib=lib)
    eng.set_tuning(256, 0)
    eng.set_shard(*ivs[rank])
    eng.push(sub)
    eng.finalize()
    eng.consensus_run(1)
    seqs, changes, minmax = shard.stitch(eng, ivs[rank], "cpu", intervals=ivs, pad=pad)
    q.put((rank, seqs, [c.tobytes() for c in changes], minmax, ivs, int(keep.sum())))
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


@pytest.mark.parametrize("world,lens,depth,pad", [
    (4, [2500] * 12, 8, None),          # many contigs (config 4 in miniature): cuts land on contig boundaries
    (8, [2000] * 20, 6, None),
    (4, [40000], 6, None),              # one contig (config 3 in miniature): position intervals
    (2, [6000, 5000], 8, 64),           # a row size that is too small: the second, exactly sized gather
])
def test_routed_shards_from_one_batch(emu_lib, world, lens, depth, pad):
    from kindel_amd import shard, synth
    from oracle import oracle as ko
    seed = 31
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_routed, args=(r, world, port, emu_lib.path, lens, depth, seed, pad, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = synth.to_numpy(synth.short_reads(lens, depth, seed=seed))
    ivs = results[0][4]
    assert all(r[4] == ivs for r in results)                       # every rank computed the same cuts
    assert ivs[0][0] == 0 and all(a[1] == b[0] for a, b in zip(ivs, ivs[1:]))
    if len(lens) > 1 and len(lens) % world == 0:                   # equal contigs, a whole number per rank: every cut is a contig base
        base, _ = shard.g_layout(lens)
        assert all(iv[0] in set(int(b) for b in base) for iv in ivs)
    n_seen = [r[5] for r in results]
    assert max(n_seen) < 2.2 * len(full["contig"]) / world         # balanced work
    oracle = {}
    for cid in range(len(lens)):
        oa = ko.parse_records(full, cid)
        oracle[cid] = (oa.consensus_sequence(), oa.depth_minmax())
    for rank, seqs, changes, minmax, _, _ in results:
        for cid in range(len(lens)):
            (oseq, och), omm = oracle[cid]
            assert seqs[cid].decode() == oseq, (rank, cid)
            assert [None if c == 0 else chr(c) for c in changes[cid]] == och
            assert minmax[cid] == omm


def test_two_rank_stitch_matches_oracle(emu_lib):
    from kindel_amd import synth
    from oracle import oracle as ko
    lens, seed, world = [9000, 7000, 3000], 21, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, emu_lib.path, lens, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = synth.to_numpy(synth.short_reads(lens, 10, seed=seed))
    assert sum(r[4] for r in results) == len(full["contig"])   # every read is owned by exactly one rank
    for rank, seqs, changes, minmax, _ in results:
        for cid in range(len(lens)):
            oa = ko.parse_records(full, cid)
            oseq, och = oa.consensus_sequence()
            assert seqs[cid].decode() == oseq, (rank, cid)
            assert [None if c == 0 else chr(c) for c in changes[cid]] == och
            assert minmax[cid] == oa.depth_minmax()
