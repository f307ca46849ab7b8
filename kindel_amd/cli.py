"""`kindel` command line (mirror of /root/reference/kindel/cli.py:9-66 with argparse, argh being absent).

Same sub-commands, flags, short options and defaults; `consensus` prints the reports to stderr
and the FASTA records (unwrapped) to stdout exactly like cli.py:30-33.
"""
import argparse
import sys

from . import __version__


def _spawn_ranks(n, argv):
    """`kindel consensus --gpus N` without a launcher: re-run this command as N ranks (one process per GPU) under
    torch.distributed.run on 127.0.0.1; rank 0 prints.  Refuses when fewer than N GPUs are visible."""
    import os
    import socket
    import subprocess
    import torch
    if os.environ.get("KINDEL_DIST_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < n:
        print("kindel: --gpus %d requested but only %d GPU(s) visible" % (n, torch.cuda.device_count()), file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "kindel_amd"] + list(argv)
    return subprocess.call(cmd, env=env)


def consensus(args):
    import os
    from . import kindel
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        sys.exit(_spawn_ranks(args.gpus, args.argv))
    if world > 1:
        # one process per GPU: every rank decodes its share of the file, piles up its interval, one all-gather stitches
        import torch
        import torch.distributed as dist
        rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
        backend = os.environ.get("KINDEL_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
            device, dev_index = "cuda:%d" % local, local
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
            dev_index = local % max(1, torch.cuda.device_count()) if torch.cuda.is_available() else 0
            device = "cpu"
        try:
            result = kindel.bam_to_consensus_sharded(args.bam_path, rank, world, device=device, dev_index=dev_index, realign=args.realign,
                                                     min_depth=args.min_depth, min_overlap=args.min_overlap,
                                                     clip_decay_threshold=args.clip_decay_threshold, mask_ends=args.mask_ends,
                                                     trim_ends=args.trim_ends, uppercase=args.uppercase)
        finally:
            dist.destroy_process_group()
        if rank != 0:
            return
    else:
        result = kindel.bam_to_consensus(args.bam_path, args.realign, args.min_depth, args.min_overlap,
                                         args.clip_decay_threshold, args.mask_ends, args.trim_ends, args.uppercase)
    print("\n".join([r for r in result.refs_reports.values()]), file=sys.stderr)
    for consensus_record in result.consensuses:
        print(f">{consensus_record.name}")
        print(consensus_record.sequence)


def weights(args):
    from . import kindel
    weights_df = kindel.weights(args.bam_path, args.relative, not args.no_confidence, args.confidence_alpha)
    weights_df.to_csv(sys.stdout, sep="\t", index=False)


def features(args):
    from . import kindel
    kindel.features(args.bam_path).to_csv(sys.stdout, sep="\t", index=False)


def variants(args):
    from . import kindel
    df = kindel.variants(args.bam_path, args.abs_threshold, args.rel_threshold, not args.all_alleles, args.absolute)
    df.to_csv(sys.stdout, sep="\t", index=False)


def plot(args):
    from . import kindel
    return kindel.plotly_clips(args.bam_path)


def version(args):
    print(f"kindel {__version__}")


def build_parser():
    p = argparse.ArgumentParser(prog="kindel")
    sub = p.add_subparsers(dest="command")
    c = sub.add_parser("consensus", help="Infer consensus sequence(s) from alignment in SAM/BAM format")
    c.add_argument("bam_path", help="path to SAM/BAM file")
    c.add_argument("-r", "--realign", action="store_true", default=False,
                   help="attempt to reconstruct reference around soft-clip boundaries")
    c.add_argument("--min-depth", type=int, default=1, help="substitute Ns at coverage depths beneath this value")
    c.add_argument("--min-overlap", type=int, default=7, help="match length required to close soft-clipped gaps")
    c.add_argument("-c", "--clip-decay-threshold", type=float, default=0.1,
                   help="read depth fraction at which to cease clip extension")
    c.add_argument("--mask-ends", type=int, default=50, help="ignore clip dominant positions within n positions of termini")
    c.add_argument("-t", "--trim-ends", action="store_true", default=False,
                   help="trim ambiguous nucleotides (Ns) from sequence ends")
    c.add_argument("-u", "--uppercase", action="store_true", default=False, help="close gaps using uppercase alphabet")
    c.add_argument("--gpus", type=int, default=1, help="(kindel_amd) GPUs of this node to spread the reference positions over: "
                   "one process per GPU, every rank decodes its share of the file, one all-gather stitches the FASTA")
    c.set_defaults(func=consensus)
    w = sub.add_parser("weights", help="Returns table of per-site nucleotide frequencies and coverage")
    w.add_argument("bam_path", help="path to SAM/BAM file")
    w.add_argument("-r", "--relative", action="store_true", default=False, help="output relative nucleotide frequencies")
    w.add_argument("-n", "--no-confidence", action="store_true", default=False,
                   help="skip confidence interval calculation")
    w.add_argument("-c", "--confidence-alpha", type=float, default=0.01, help="confidence interval alpha value")
    w.set_defaults(func=weights)
    f = sub.add_parser("features", help="Returns table of per-site nucleotide frequencies and coverage including indels")
    f.add_argument("bam_path", help="path to SAM/BAM file")
    f.set_defaults(func=features)
    va = sub.add_parser("variants", help="Output variants exceeding specified absolute and relative frequency "
                        "thresholds (extension: announced in the reference README, not implemented there)")
    va.add_argument("bam_path", help="path to SAM/BAM file")
    va.add_argument("-a", "--abs-threshold", type=int, default=1, help="minimum allele count")
    va.add_argument("-r", "--rel-threshold", type=float, default=0.01, help="minimum allele frequency")
    va.add_argument("--all-alleles", action="store_true", default=False, help="also list the majority allele")
    va.add_argument("--absolute", action="store_true", default=False, help="omit the frequency column")
    va.set_defaults(func=variants)
    pl = sub.add_parser("plot", help="Plot sitewise soft clipping frequency across reference and genome")
    pl.add_argument("bam_path", help="path to SAM/BAM file")
    pl.set_defaults(func=plot)
    v = sub.add_parser("version", help="Show version")
    v.set_defaults(func=version)
    return p


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    args.argv = list(sys.argv[1:] if argv is None else argv)
    if not getattr(args, "func", None):
        parser.print_help()
        return 1
    args.func(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
