"""TEST INFRASTRUCTURE ONLY -- import the *unmodified* reference module.

``load_reference()`` returns kindel.kindel of the reference as a module, with the three absent third-party imports
(simplesam, dnaio, argh) satisfied by oracle/ref_shims/ and tqdm silenced: from /root/reference (the build container),
or from the sourceless bytecode oracle/make_ref.py compiled from it into oracle/_ref/ (the GPU box, which has no
/root/reference: the build product travels with the snapshot like the built .so files).  ``origin()`` says which.
"""
import importlib
import os
import sys

REF_ROOT = "/root/reference"
_HERE = os.path.dirname(os.path.abspath(__file__))
_REF_BUILT = os.path.join(_HERE, "_ref")


def _source_available():
    return os.path.isfile(os.path.join(REF_ROOT, "kindel", "kindel.py"))


def _bytecode_available():
    return os.path.isfile(os.path.join(_REF_BUILT, "kindel", "kindel.pyc"))


def reference_available():
    return _source_available() or _bytecode_available()


def origin():
    """'source' (/root/reference), 'bytecode' (oracle/_ref, compiled from it by oracle/make_ref.py) or None"""
    return "source" if _source_available() else "bytecode" if _bytecode_available() else None


def load_reference():
    if not reference_available():
        raise RuntimeError("reference not present: neither %s nor %s (python -m oracle.make_ref)" % (REF_ROOT, _REF_BUILT))
    root = REF_ROOT if _source_available() else _REF_BUILT
    repo = os.path.dirname(_HERE)
    if repo not in sys.path:
        sys.path.insert(0, repo)
    for p in (os.path.join(_HERE, "ref_shims"), root):     # BEHIND everything else: the reference tree has a `tests` package too
        if p not in sys.path:
            sys.path.append(p)
    os.environ.setdefault("TQDM_DISABLE", "1")
    mod = importlib.import_module("kindel.kindel")
    assert os.path.abspath(mod.__file__).startswith(root), mod.__file__
    import tqdm

    class _Quiet:
        @staticmethod
        def tqdm(it, **kw):
            return it

    mod.tqdm = _Quiet
    return mod
