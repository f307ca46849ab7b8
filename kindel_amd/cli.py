"""`kindel` command line (mirror of /root/reference/kindel/cli.py:9-66 with argparse, argh being absent).

Same sub-commands, flags, short options and defaults; `consensus` prints the reports to stderr
and the FASTA records (unwrapped) to stdout exactly like cli.py:30-33.
"""
import argparse
import sys

from . import __version__


def consensus(args):
    from . import kindel
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
    if not getattr(args, "func", None):
        parser.print_help()
        return 1
    args.func(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
