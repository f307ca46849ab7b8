"""TEST INFRASTRUCTURE ONLY -- stand-in for simplesam==0.1.3.2 (absent here, no network).

Lets the unmodified reference (/root/reference/kindel/kindel.py:136-145) iterate SAM/BAM
records without samtools.  Contains no pileup/consensus arithmetic: it only parses files
(via oracle/samio_py.py) and exposes the attributes the reference reads
(header["@SQ"], rname, pos, mapped, seq, cigars).
"""
from collections import OrderedDict

from oracle import samio_py


class Reader:
    def __init__(self, fh):
        path = fh.name
        text, refs, recs = samio_py.read_alignment_file(path)
        self.header = OrderedDict()
        # kindel/kindel.py:138-141 expects {"SN:name": ["LN:len", ...]}
        self.header["@SQ"] = OrderedDict(("SN:%s" % n, ["LN:%d" % l]) for n, l in refs)
        self._recs = recs

    def __iter__(self):
        return iter(self._recs)
