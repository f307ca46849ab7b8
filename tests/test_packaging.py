"""Drop-in packaging: the `kindel` executable and the `kindel` import name, as bede/kindel's own packaging
provides them (/root/reference/pyproject.toml:37-38; its tests call `kindel consensus <bam> > out.fa`, tests/test_kindel.py:119-121)."""
import os
import stat
import subprocess
import sys

import pytest

from tools import synth
from tests import parity as P
from tests import refcheck as RC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _entry_points():
    try:
        import tomllib as toml
    except ImportError:
        import tomli as toml
    with open(os.path.join(ROOT, "pyproject.toml"), "rb") as fh:
        return toml.load(fh)


def test_pyproject_declares_the_reference_entry_point_and_import_name():
    cfg = _entry_points()
    assert cfg["project"]["scripts"] == {"kindel": "kindel_amd.cli:main"}
    assert "kindel" in cfg["tool"]["setuptools"]["packages"]
    shim = os.path.join(ROOT, cfg["tool"]["setuptools"]["package-dir"]["kindel"], "__init__.py")
    assert os.path.isfile(shim)


def _console_script(tmp_path):
    """What pip generates for [project.scripts] kindel = "module:func": a launcher on PATH."""
    mod, func = _entry_points()["project"]["scripts"]["kindel"].split(":")
    bindir = tmp_path / "bin"
    bindir.mkdir()
    exe = bindir / "kindel"
    exe.write_text("#!%s\nimport sys\nfrom %s import %s\nsys.exit(%s())\n" % (sys.executable, mod, func, func))
    exe.chmod(exe.stat().st_mode | stat.S_IXUSR)
    env = dict(os.environ)
    env["PATH"] = str(bindir) + os.pathsep + env.get("PATH", "")
    env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "dropin")] + ([env["PYTHONPATH"]] if env.get("PYTHONPATH") else []))
    return env


def _run_like_the_reference_tests(env, tmp_path, key, extra=""):
    bam = RC.bam_of(tmp_path, key)
    out = tmp_path / (key + ".fa")
    subprocess.run("kindel consensus %s %s > %s" % (extra, bam, out), shell=True, check=True, env=env)     # test_kindel.py:119-121
    seqs, name = {}, None
    for line in open(out):
        if line.startswith(">"):
            name = line[1:].split()[0]; seqs[name] = ""
        else:
            seqs[name] += line.strip()
    return seqs


def _check_import_name(env):
    imp = subprocess.run([sys.executable, "-c", "import kindel; from kindel import kindel as k, cli; import kindel.kindel as kk; "
                          "assert k is kk and callable(k.bam_to_consensus) and callable(cli.main); print(kindel.__version__, k.__name__)"],
                         env=env, capture_output=True, text=True, check=True)
    assert imp.stdout.split() == ["1.2.1", "kindel_amd.kindel"]


def test_kindel_executable_and_import_name(tmp_path):
    """The generated console script starts and the import name resolves (no GPU needed for that; the record loop itself has no CPU
    path -- `kindel consensus` is run by the -m gpu test below, in-process CLI runs on the kernel emulator by tests/test_host_api.py)."""
    env = _console_script(tmp_path)
    out = subprocess.run("kindel version", shell=True, check=True, env=env, capture_output=True, text=True)
    assert "1.2.1" in out.stdout
    _check_import_name(env)


@pytest.mark.gpu
def test_kindel_consensus_executable_on_the_gpu(hip_lib, tmp_path):
    """`kindel consensus <bam> > out.fa` exactly as /root/reference/tests/test_kindel.py:119-121 runs it, against the reference's
    golden FASTA files."""
    env = _console_script(tmp_path)
    for key in ("bwa_mem__1.1.sub_test", "minimap2__1.1.multi"):
        got = _run_like_the_reference_tests(env, tmp_path, key)
        for name, seq in RC.REF_FASTA[key]["default"].items():
            assert got[name].upper() == seq.upper(), (key, name)
    _check_import_name(env)
