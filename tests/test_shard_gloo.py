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
    from kindel_amd import shard
    from tools import synth
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
    from kindel_amd import shard
    from tools import synth
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


def _worker_attached(rank, world, port, emu_path, lens, depth, seed, pad, q):
    """The per-step shape of bench.py --gpus N: the exchange row registered with the engine (kd_set_exchange), filled on the way
    of kd_finish, the collective alone behind it -- twice, as two steps of a timed loop are."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kindel_amd import _native as N
    from kindel_amd import shard
    from tools import synth
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
    ex = shard.Exchange(eng, ivs[rank], "cpu", intervals=ivs, pad=pad).attach()
    out = np.zeros(sum(lens) * 2 + 4096, np.uint8)
    needs = []
    for _ in range(2):
        eng.reset()
        eng.push(sub)
        eng.finish(out)
        rows = ex.collect()
        needs.append(ex.need(rows))
    if needs[-1] > ex.pad:       # the announced size: every rank takes the same decision from the same headers
        ex.detach()
        ex = shard.Exchange(eng, ivs[rank], "cpu", intervals=ivs, pad=needs[-1])
        rows = ex.run()
    on_demand = shard.gather(eng, ivs[rank], "cpu", intervals=ivs, pad=ex.pad)[0]
    sizes = rows[:, :8].contiguous().view(torch.int64).reshape(-1).tolist()      # (behind a row's own bytes: whatever the buffer held)
    same = rows.shape == on_demand.shape and all(bool((rows[r, :n] == on_demand[r, :n]).all()) for r, n in enumerate(sizes))
    seqs, changes, minmax = shard.assemble(np.ascontiguousarray(rows.numpy()), lens, world, ivs[rank], intervals=ivs)
    q.put((rank, seqs, [c.tobytes() for c in changes], minmax, needs, same))
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


@pytest.mark.parametrize("world,lens,depth,pad", [
    (2, [6000, 5000], 8, None),
    (4, [30000], 6, None),
    (2, [6000, 5000], 8, 200),          # rows that are too small: headers only, the sizes announced
])
def test_attached_exchange_row(emu_lib, world, lens, depth, pad):
    from tools import synth
    from oracle import oracle as ko
    seed = 37
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_attached, args=(r, world, port, emu_lib.path, lens, depth, seed, pad, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = synth.to_numpy(synth.short_reads(lens, depth, seed=seed))
    for rank, seqs, changes, minmax, needs, same in results:
        assert needs[0] == needs[1] == results[0][4][0]            # both steps, every rank: the same announced size
        assert (needs[0] > pad) if pad else True
        assert same                                                 # the row left by kd_finish == the row on demand
        for cid in range(len(lens)):
            oa = ko.parse_records(full, cid)
            oseq, och = oa.consensus_sequence()
            assert seqs[cid].decode() == oseq, (rank, cid)
            assert [None if c == 0 else chr(c) for c in changes[cid]] == och
            assert minmax[cid] == oa.depth_minmax()


@pytest.mark.parametrize("world,lens,depth,pad", [
    (4, [2500] * 12, 8, None),          # many contigs (config 4 in miniature): cuts land on contig boundaries
    (8, [2000] * 20, 6, None),
    (4, [40000], 6, None),              # one contig (config 3 in miniature): position intervals
    (2, [6000, 5000], 8, 64),           # a row size that is too small: the second, exactly sized gather
])
def test_routed_shards_from_one_batch(emu_lib, world, lens, depth, pad):
    from kindel_amd import shard
    from tools import synth
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
    from tools import synth
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


def _worker_file(rank, world, port, emu_path, path, q, kw=None):
    """The product's multi-GPU entry on CPU: every rank decodes its share of ONE file (gloo, emulated kernels)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kindel_amd import _native as N
    from kindel_amd import kindel as K
    from kindel_amd import shard
    lib = N.Library(emu_path)
    N._default = lib
    out = shard.pileup_consensus_sharded(path, rank, world, device="cpu", lib=lib)
    try:
        res = K.bam_to_consensus_sharded(path, rank, world, device="cpu", lib=lib, **(kw or {}))
        res = ([(c.name, c.sequence) for c in res.consensuses], res.refs_reports, {k: "".join("." if c is None else c for c in v) for k, v in res.refs_changes.items()})
    except Exception as e:       # noqa: BLE001 -- handed to the parent
        res = repr(e)
    q.put((rank, out["mode"], out["stats"], [out["names"][c] for c in out["order"]], res))
    dist.barrier()
    dist.destroy_process_group()


def _run_file_ranks(emu_lib, path, world, kw=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_file, args=(r, world, port, emu_lib.path, path, q, kw)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


@pytest.mark.parametrize("key,world,block_bytes,expect_mode", [
    ("bwa_mem__1.1.sub_test", 2, 900, "sharded"),        # the reference's own HCV fixture, coordinate sorted
    ("bwa_mem__4.1.sub_test", 3, 2000, "sharded"),
    ("segemehl__2.1.sub_test", 4, 1500, "sharded"),      # indel-heavy
    ("minimap2__hxb2-gp120-mutated", 2, 1200, "whole-file"),   # SO:unsorted in the reference: every rank reads the whole file
])
def test_one_file_across_ranks_matches_the_reference_golden(emu_lib, tmp_path, key, world, block_bytes, expect_mode):
    """`kindel consensus --gpus N x.bam` underneath: every rank opens the file and decodes only the BGZF blocks of its share
    (+ the neighbouring reads that reach into its interval), one all-gather stitches -- consensus, report and change codes equal
    what the unmodified reference produced for the fixture (tests/golden), on every rank."""
    from tools import synth
    from tests import parity as P
    gold = P.golden_outputs()[key]["contigs"]
    path = str(tmp_path / (key + ".bam"))
    synth.write_bam(path, P.load_fixture(key), sort_order="unknown", block_bytes=block_bytes)
    results = _run_file_ranks(emu_lib, path, world)
    assert [r[1] for r in results] == [expect_mode] * world, results[0][1:3]
    if expect_mode == "sharded":
        total = len(P.load_fixture(key)["contig"])
        assert sum(r[2]["decoded_records"] for r in results) == total          # every record decoded by exactly one rank
        assert max(r[2]["decoded_records"] for r in results) < 1.5 * total / world + 50
        assert all(r[2]["neighbour_records"] < 0.2 * total + 200 for r in results)
    for rank, mode, stats, order, res in results:
        assert not isinstance(res, str), res
        recs, reports, changes = res
        assert order == [g["name"] for g in gold]
        assert recs == [(g["name"] + "_cns", g["consensus"]) for g in gold], rank
        for g in gold:
            assert reports[g["name"]] == g["report"].replace("{bam_path}", path)
            assert changes[g["name"]] == g["changes"]


@pytest.mark.parametrize("key,world,block_bytes", [
    ("bwa_mem__1.1.sub_test", 2, 900),
    ("bwa_mem__1.1.sub_test", 5, 400),                   # cuts inside and next to the clip-dominant regions
    ("segemehl__2.1.sub_test", 3, 1500),
    ("ext__1.issue23.debug", 4, 700),                    # the reference's --realign regression fixture (issue 23)
    ("minimap2__hxb2-gp120-mutated", 2, 1200),           # unsorted: whole-file fallback, then the same realign
    ("minimap2__1.1.multi", 3, 800),                     # three contigs
])
def test_realign_across_ranks_matches_the_reference_golden(emu_lib, tmp_path, key, world, block_bytes):
    """`kindel consensus --realign --gpus N` (round 5): the clip tables of the shards are summed (one all-reduce per contig), every rank
    finds the same clip-dominant regions and patches its part -- the consensus and the report (with its region list) equal what
    the unmodified reference printed for the fixture with realign=True, on every rank."""
    from tools import synth
    from tests import parity as P
    gold = [g for g in P.golden_outputs()[key]["contigs"] if "realign_consensus" in g]
    assert gold
    path = str(tmp_path / (key + ".bam"))
    synth.write_bam(path, P.load_fixture(key), sort_order="unknown", block_bytes=block_bytes)
    results = _run_file_ranks(emu_lib, path, world, dict(realign=True, min_overlap=7))
    for rank, mode, stats, order, res in results:
        assert not isinstance(res, str), res
        recs, reports, changes = res
        seqs = dict(recs)
        for g in gold:
            assert seqs[g["name"] + "_cns"] == g["realign_consensus"], (rank, g["name"])
            assert reports[g["name"]] == g["realign_report"].replace("{bam_path}", path)


def _worker_scan_inputs(rank, world, port, emu_path, path, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kindel_amd import _native as N
    from kindel_amd import shard
    lib = N.Library(emu_path)
    N._default = lib
    out = shard.pileup_consensus_sharded(path, rank, world, device="cpu", lib=lib,
                                         realign=dict(min_overlap=7, clip_decay_threshold=0.1, mask_ends=50, keep_scan_inputs=True))
    q.put((rank, {c: [{k: v.tolist() for k, v in side.items()} for side in pair] for c, pair in out["scan_inputs"].items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 5])
def test_scan_inputs_gathered_over_the_shards_equal_the_whole_file_scan_inputs(emu_lib, tmp_path, world):
    """What --realign across ranks scans (round 6: shard.realign_patches gathers the SPARSE per-site predicates of every rank's own
    sites, no table crosses the links): the clip-dominant candidates and extension sites of every contig, with their consensus
    characters, concatenated over the ranks == those of the single-process tables -- also at the cut sites, which two neighbouring
    contexts both commit (the halo site: only its owner reports it)."""
    from tools import synth
    from kindel_amd import _native as N
    from kindel_amd import kindel as K
    batch = synth.to_numpy(synth.short_reads([3000, 1500, 2200], 25, seed=77, clip_p=0.5))
    path = str(tmp_path / "t.bam")
    synth.write_bam(path, batch, sort_order="coordinate", block_bytes=900)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_scan_inputs, args=(r, world, port, emu_lib.path, path, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    N._default = emu_lib
    pl = K.pileup_file(path)
    try:
        n_sites = 0
        for rank, got in results:
            assert sorted(got) == sorted(pl.order)
            for cid in pl.order:
                t = pl.engine.tables(cid)
                L = int(pl.lens[cid])
                W = np.ascontiguousarray(t[0:5, :L].T)
                for side, ch in ((0, N.KD_CH_CSW), (1, N.KD_CH_CEW)):
                    want = K.cdr_scan_inputs(W, t[N.KD_CH_DEL, :L], np.ascontiguousarray(t[ch:ch + 5, :L].T), 0.1, 50, L)
                    for k in ("cand", "cand_ch", "ext", "ext_ch"):
                        assert got[cid][side][k] == want[k].tolist(), (rank, cid, side, k)
                    n_sites += len(want["ext"])
        assert n_sites > 0      # (clip_p = 0.5: there ARE extension sites)
    finally:
        pl.engine.close()


def test_one_file_across_ranks_synthetic_multi_contig(emu_lib, tmp_path):
    """Many contigs, more ranks than some shares deserve (empty intervals), reads with leading clips across the cuts: vs the oracle."""
    from tools import synth
    from oracle import oracle as ko
    batch = synth.to_numpy(synth.short_reads([4000, 2500, 300, 5000], 12, seed=41, clip_p=0.3))
    path = str(tmp_path / "m.bam")
    synth.write_bam(path, batch, sort_order="coordinate", block_bytes=1100)
    results = _run_file_ranks(emu_lib, path, 5)
    assert all(r[1] == "sharded" for r in results)
    for rank, mode, stats, order, res in results:
        assert not isinstance(res, str), res
        recs, reports, changes = res
        assert [n for n, _ in recs] == ["ctg%d_cns" % c for c in ko.contig_order(batch)] or len(recs) == 4
        for (name, seq), cid in zip(recs, ko.contig_order(batch)):
            oa = ko.parse_records(batch, cid)
            oseq, och = oa.consensus_sequence()
            assert seq == oseq, (rank, cid)
            assert changes[name[:-4]] == "".join("." if c is None else c for c in och)


def test_a_reference_exception_reaches_every_rank(emu_lib, tmp_path):
    """A base outside A,C,G,T,N in one rank's share: KeyError on every rank (kindel.py:51-52), not a hang."""
    from tools import synth
    batch = synth.to_numpy(synth.short_reads([6000], 10, seed=43))
    seq4 = batch["seq4"].copy()
    i = len(batch["contig"]) // 4                     # a read in the first rank's share
    seq4[int(batch["seq_off"][i]) + 3] = 0x3C         # 'M' (an IUPAC code) inside an aligned segment
    batch["seq4"] = seq4
    path = str(tmp_path / "bad.bam")
    synth.write_bam(path, batch, sort_order="coordinate", block_bytes=1500)
    results = _run_file_ranks_expect_error(emu_lib, path, 3)
    assert all("KeyError" in r for r in results), results


def _worker_file_err(rank, world, port, emu_path, path, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kindel_amd import _native as N
    from kindel_amd import kindel as K
    lib = N.Library(emu_path)
    try:
        K.bam_to_consensus_sharded(path, rank, world, device="cpu", lib=lib)
        q.put("no error")
    except Exception as e:       # noqa: BLE001
        q.put(type(e).__name__ + ": " + str(e))
    dist.barrier()
    dist.destroy_process_group()


def _run_file_ranks_expect_error(emu_lib, path, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_file_err, args=(r, world, port, emu_lib.path, path, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


def _worker_one_rank_fails(rank, world, port, emu_path, path, what, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kindel_amd import _native as N
    from kindel_amd import kindel as K
    lib = N.Library(emu_path)
    if rank == 1:      # this rank ALONE runs out of memory: in the whole-file fallback's decode, or when it allocates its tables
        def boom(*a, **k):
            raise MemoryError("rank 1 has no memory for " + what)
        if what == "decode":
            N.decode_file = boom
        else:
            N.Engine = boom
    try:
        K.bam_to_consensus_sharded(path, rank, world, device="cpu", lib=lib)
        q.put("no error")
    except Exception as e:       # noqa: BLE001
        q.put(type(e).__name__ + ": " + str(e))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("what", ["decode", "engine"])
def test_a_failure_of_one_rank_alone_reaches_every_rank(emu_lib, tmp_path, what):
    """One rank failing where the others do not -- the whole-file fallback's decode (a SAM text file: no BGZF spans), the engine's
    tables -- must not leave the others waiting in the next collective: the failure travels through the gather."""
    from tools import synth
    batch = synth.to_numpy(synth.short_reads([4000], 8, seed=44))
    bam = str(tmp_path / "ok.bam")
    synth.write_bam(bam, batch, sort_order="coordinate", block_bytes=1500)
    path = bam
    if what == "decode":      # the fallback is taken by a file that has no BGZF blocks to share out
        import gzip
        path = str(tmp_path / "plain.bam")
        from kindel_amd import _native as N   # (the BGZF members concatenated = one valid gzip stream; rewritten as ONE member)
        with gzip.open(bam, "rb") as src, gzip.open(path, "wb") as dst:
            dst.write(src.read())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_one_rank_fails, args=(r, 3, port, emu_lib.path, path, what, q)) for r in range(3)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(3)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r.startswith("MemoryError") and "rank 1 has no memory" in r for r in results), results


def test_random_files_across_three_ranks_equal_the_single_process_result(emu_lib):
    """tests/shard_fuzz.py: random BAM files with tiny BGZF blocks (many spans, neighbour decodes across several blocks), sorted
    (rank-sharded ingest) and unsorted (whole-file fallback), one to four contigs -- every rank's consensus and change codes equal the
    single-process run's.  1 500 such files over 2 - 8 ranks ran clean as a local campaign (round 5)."""
    from tests import shard_fuzz
    files, diffs = shard_fuzz.run_campaign(25, 90000, 3, emu_lib.path)
    assert len(files) >= 20 and not diffs, diffs


def test_files_with_clip_dominant_regions_realigned_across_four_ranks(emu_lib):
    """--realign across ranks on files that HAVE clip-dominant regions (reference_fuzz.structured_sam; tiny BGZF blocks, sorted and
    unsorted): every rank's sequences, change codes and reports (with the region lists) equal the single-process realign run's --
    which tests/test_reference_fuzz.py holds against the unmodified reference on files of the same generator."""
    import logging
    from tests import shard_fuzz
    logging.disable(logging.WARNING)
    try:
        files, diffs = shard_fuzz.run_campaign(12, 95000, 4, emu_lib.path, kw=dict(realign=True, min_overlap=7), structured=True)
    finally:
        logging.disable(logging.NOTSET)
    assert len(files) == 12 and not diffs, diffs
