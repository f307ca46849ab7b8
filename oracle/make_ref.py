"""TEST / MEASUREMENT INFRASTRUCTURE ONLY -- byte-compile the UNMODIFIED reference into oracle/_ref/ (a build product).

    python -m oracle.make_ref          (also run by __graft_entry__.build() whenever /root/reference exists)

The GPU box has no /root/reference, but bench.py's `cpu_baseline` leg is asked to time the reference's pure-Python path
(kindel.kindel.parse_records + consensus_sequence, /root/reference/kindel/kindel.py:21-128, :384-430) ON THAT NODE, in
the same run.  The C oracle's recipe compiles C sources where they lie into oracle/_ref/*.so; this is the same recipe for
a Python reference: `py_compile` turns /root/reference/kindel/{__init__,kindel,cli}.py, read where they lie, into
sourceless bytecode oracle/_ref/kindel/*.pyc.  oracle/_ref/ is git-ignored (no reference source or derivative enters the
history) and is NOT in .gpurunignore, so the bytecode travels to the GPU box like the built .so files do.  Both boxes run
the same image (CPython 3.10): the bytecode's magic number is checked at import.

The same recipe stages the reference's own INPUT files -- the 15 htslib-written BAMs and 3 SAMs of /root/reference/tests/data_* --
into oracle/_ref/fixtures/ (git-ignored, travels): the `-m gpu` suite then runs the device-side ingest (k_gpu_inflate, k_bam_*) on
third-party files ON the MI355X, not only on BAMs this repo's own writer produced.  Inputs only; what the reference's tests
compare the outputs with is committed as tests/golden/reference_fasta.json.

Only oracle/refrun.py imports the bytecode, and only tests/, smoke() and bench.py's cpu_baseline leg use refrun.
"""
import os
import py_compile
import sys

REF_PKG = "/root/reference/kindel"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref", "kindel")
FILES = ("__init__.py", "kindel.py", "cli.py")
REF_TESTS = "/root/reference/tests"
FIXTURES = os.path.join(HERE, "_ref", "fixtures")


def available():
    return all(os.path.isfile(os.path.join(REF_PKG, f)) for f in FILES)


def built():
    return all(os.path.isfile(os.path.join(OUT, f + "c")) for f in FILES)


def build(force=False):
    """-> path of oracle/_ref (to be put on sys.path), or None when the reference tree is absent and nothing was built before"""
    if not available():
        return os.path.dirname(OUT) if built() else None
    os.makedirs(OUT, exist_ok=True)
    for f in FILES:
        src, dst = os.path.join(REF_PKG, f), os.path.join(OUT, f + "c")
        if force or not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
            # dfile: the path tracebacks show -- the reference's own, so that file:line citations stay meaningful
            py_compile.compile(src, cfile=dst, dfile=src, doraise=True)
    stage_fixtures()
    with open(os.path.join(os.path.dirname(OUT), "README"), "w") as fh:
        fh.write("Build product of oracle/make_ref.py: sourceless bytecode of the unmodified reference (%s), CPython %s.\n"
                 "Git-ignored; test / measurement infrastructure only.\n" % (REF_PKG, sys.version.split()[0]))
    return os.path.dirname(OUT)


def stage_fixtures():
    """The reference's test INPUTS (*.bam, *.sam) -> oracle/_ref/fixtures/<data_dir>/ (byte-identical copies, git-ignored)."""
    import glob
    import shutil
    n = 0
    for src in sorted(glob.glob(os.path.join(REF_TESTS, "data_*", "*.bam")) + glob.glob(os.path.join(REF_TESTS, "data_*", "*.sam"))):
        dst = os.path.join(FIXTURES, os.path.basename(os.path.dirname(src)), os.path.basename(src))
        if not os.path.exists(dst) or os.path.getsize(dst) != os.path.getsize(src) or os.path.getmtime(dst) < os.path.getmtime(src):
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            shutil.copyfile(src, dst)
        n += 1
    return n


def fixture(rel):
    """Path of a staged reference input, e.g. fixture("data_bwa_mem/1.1.sub_test.bam"): the reference tree's own file where that
    exists, the staged copy elsewhere (the GPU box), None if neither."""
    for root in (REF_TESTS, FIXTURES):
        p = os.path.join(root, rel)
        if os.path.isfile(p):
            return p
    return None


def fixtures(pattern="*.bam"):
    import glob
    for root in (REF_TESTS, FIXTURES):
        got = sorted(glob.glob(os.path.join(root, "data_*", pattern)))
        if got:
            return got
    return []


if __name__ == "__main__":
    p = build(force=True)
    print("oracle/_ref:", p if p else "reference tree not present, nothing built")
