"""Drop-in import name: ``from kindel import kindel`` / ``import kindel.cli`` resolve to the MI355X engine (kindel_amd).
Mirrors /root/reference/kindel/__init__.py:3 (the version constant the CLI prints)."""
import sys as _sys

from kindel_amd import __version__, cli, kindel  # noqa: F401

_sys.modules[__name__ + ".kindel"] = kindel
_sys.modules[__name__ + ".cli"] = cli
