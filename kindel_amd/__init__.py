"""kindel_amd -- MI355X-native pileup + majority-consensus engine, drop-in for bede/kindel's hot path.

``from kindel_amd import kindel`` mirrors ``from kindel import kindel``
(/root/reference/kindel/__init__.py:3 keeps the version constant used by the CLI).
"""
__version__ = "1.2.1"
