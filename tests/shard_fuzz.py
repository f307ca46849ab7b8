"""TEST INFRASTRUCTURE (round 5).  `python -m tests.shard_fuzz N SEED0 WORLD [realign MIN_OVERLAP]`: a local campaign -- 1 500 files over 2, 3, 4, 5 and 8 ranks ran
clean, and 1 510 files with clip-dominant regions under --realign over 2 - 8 ranks; tests/test_shard_gloo.py keeps 25 files on three ranks.  The N-rank one-file entry (bam_to_consensus_sharded over gloo, emulated kernels) on random BAM files with tiny
BGZF blocks against the single-process result of the same file -- sorted files (rank-sharded ingest, neighbour decodes, cuts) and
unsorted ones (whole-file fallback)."""
import os, sys, random, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import torch.multiprocessing as mp

def make_files(n, seed0, tmp, structured=False):
    import numpy as np
    from tools import synth
    from tests import parity as P, reference_fuzz as RF
    files = []
    for seed in range(seed0, seed0 + n):
        rng = random.Random(seed)
        if structured:       # novel segments in the sample, reads clipped at the junctions: files --realign patches (reference_fuzz.structured_sam)
            b = P.sam_to_batch(RF.structured_sam(rng, sort=bool(seed % 4)))
            path = os.path.join(tmp, "s%d.bam" % seed)
            synth.write_bam(path, b, names=[str(x) for x in b["contig_names"]], sort_order="coordinate" if seed % 4 else "unsorted", block_bytes=rng.choice([300, 700, 2000, 65000]))
            files.append(path)
            continue
        txt = RF.rand_sam(rng, rng.randint(1, 4), realistic=True)
        if seed % 4:      # coordinate-sort the records (header order of contigs), like `samtools sort`
            lines = txt.rstrip("\n").split("\n")
            hdr = [l for l in lines if l.startswith("@")]
            body = [l for l in lines if not l.startswith("@")]
            body.sort(key=lambda l: (int(l.split("\t")[2][1:]), int(l.split("\t")[3])))
            txt = "\n".join(hdr + body) + "\n"
        b = P.sam_to_batch(txt)
        if len(b["contig"]) == 0: continue
        path = os.path.join(tmp, "f%d.bam" % seed)
        synth.write_bam(path, b, names=[str(x) for x in b["contig_names"]], sort_order="coordinate" if seed % 4 else "unsorted", block_bytes=rng.choice([150, 300, 700, 2000, 65000]))
        files.append(path)
    return files

def worker(rank, world, port, emu_path, files, q, kw=None):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kindel_amd import _native as N
    from kindel_amd import kindel as K
    lib = N.Library(emu_path); N._default = lib
    out = []
    for path in files:
        try:
            res = K.bam_to_consensus_sharded(path, rank, world, device="cpu", lib=lib, **(kw or {}))
            out.append(("ok", [(c.name, c.sequence) for c in res.consensuses], {k: list(v) for k, v in res.refs_changes.items()}, dict(res.refs_reports)))
        except Exception as e:
            out.append(("raise", type(e).__name__, str(e)[:100]))
    q.put((rank, out))
    dist.barrier(); dist.destroy_process_group()

def run_campaign(n, seed0, world, emu_path, kw=None, structured=False):
    """-> (files, differences): every rank's result of every file against the single-process result of the same file"""
    import socket
    from kindel_amd import _native as N
    N._default = N.Library(emu_path)
    from kindel_amd import kindel as K
    tmp = tempfile.mkdtemp()
    files = make_files(n, seed0, tmp, structured)
    single = []
    for p in files:
        try:
            r = K.bam_to_consensus(p, **(kw or {}))
            single.append(("ok", [(c.name, c.sequence) for c in r.consensuses], {k: list(v) for k, v in r.refs_changes.items()}, dict(r.refs_reports)))
        except Exception as e:      # noqa: BLE001
            single.append(("raise", type(e).__name__, str(e)[:100]))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, emu_path, files, q, kw)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=3000) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    diffs = []
    for k, p in enumerate(files):
        for r in range(world):
            a, b = single[k], res[r][k]
            same = (a[0] == b[0] == "ok" and a[1] == b[1] and a[2] == b[2] and a[3] == b[3]) or (a[0] == b[0] == "raise" and a[1] == b[1])
            if not same:
                diffs.append("%s rank %d: single %s, sharded %s" % (os.path.basename(p), r, a[0], b[:2] if b[0] == "raise" else "ok"))
                break
    return files, diffs


if __name__ == "__main__":
    import __graft_entry__ as g
    n, seed0, world = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    kw = dict(realign=True, min_overlap=int(sys.argv[5]) if len(sys.argv) > 5 else 7) if len(sys.argv) > 4 and sys.argv[4] == "realign" else None
    if kw:
        import logging
        logging.disable(logging.WARNING)
    files, diffs = run_campaign(n, seed0, world, g.build_emu(), kw, structured=bool(kw))
    for d in diffs:
        print("DIFF", d, flush=True)
    print("shard fuzz done:", len(files), "files, world", world, "diffs:", len(diffs), flush=True)
