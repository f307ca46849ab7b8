"""TEST INFRASTRUCTURE ONLY -- import the *unmodified* reference module.

``load_reference()`` returns /root/reference/kindel/kindel.py as a module, with the three
absent third-party imports (simplesam, dnaio, argh) satisfied by oracle/ref_shims/ and
tqdm silenced.  Only usable in the build container (the GPU box has no /root/reference).
"""
import importlib
import os
import sys

REF_ROOT = "/root/reference"


def reference_available():
    return os.path.isfile(os.path.join(REF_ROOT, "kindel", "kindel.py"))


def load_reference():
    if not reference_available():
        raise RuntimeError("reference tree not present")
    here = os.path.dirname(os.path.abspath(__file__))
    repo = os.path.dirname(here)
    for p in (os.path.join(here, "ref_shims"), REF_ROOT, repo):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.setdefault("TQDM_DISABLE", "1")
    mod = importlib.import_module("kindel.kindel")
    assert mod.__file__.startswith(REF_ROOT), mod.__file__
    import tqdm

    class _Quiet:
        @staticmethod
        def tqdm(it, **kw):
            return it

    mod.tqdm = _Quiet
    return mod
