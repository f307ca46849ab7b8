"""N > 1 path on CPU: two processes, gloo backend, each rank owns a G-space interval (emulated kernels),
the all-gather stitch must reproduce the oracle's consensus for every contig on every rank."""
import os
import socket
import sys

import numpy as np
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
