"""TEST INFRASTRUCTURE ONLY -- pure-Python SAM/BAM reader used by the oracle tooling.

This file is part of ``oracle/`` : it is imported only by ``tests/``, by the golden
generator (``oracle/make_golden.py``) and by the ``simplesam`` stand-in under
``oracle/ref_shims/`` that lets the *unmodified* reference module
(/root/reference/kindel/kindel.py) run in a container that has neither samtools nor
simplesam.  The product decoder is the C++ one inside libkindel_hip.so
(kindel_amd/csrc/kd_decode.cpp); this reader exists so that decoder can be checked
against an independent implementation.

Format sources: SAM/BAM specification (SAMv1 section 4.2 "The BAM format", 4.2.3 SEQ
nibble table "=ACMGRSVTWYHKDBN", section 1.4 mandatory SAM fields).  The record
attributes exposed (rname, pos, flag, mapped, seq, cigar, cigars) are the five the
reference touches: kindel/kindel.py:42-48 and :144-145.
"""
import gzip
import struct

import numpy as np

NIB2CHR = "=ACMGRSVTWYHKDBN"
CIGOPS = "MIDNSHP=X"
_CHR2NIB = {c: i for i, c in enumerate(NIB2CHR)}
_CHR2NIB.update({c.lower(): i for i, c in enumerate(NIB2CHR)})


class Rec:
    __slots__ = ("qname", "flag", "rname", "pos", "cigar", "seq", "refid")

    def __init__(self, qname, flag, rname, pos, cigar, seq, refid):
        self.qname, self.flag, self.rname, self.pos = qname, flag, rname, pos
        self.cigar, self.seq, self.refid = cigar, seq, refid

    @property
    def mapped(self):
        return not (self.flag & 0x4)

    @property
    def cigars(self):
        # simplesam 0.1.3.2 behaviour relied on by kindel/kindel.py:47 ("StopIteration ->
        # RuntimeError"): CIGAR '*' yields (0, None) and then the generator dies with
        # RuntimeError (PEP 479) when asked for a second item.
        if self.cigar == "*":
            yield (0, None)
            raise RuntimeError("generator raised StopIteration")
        num = 0
        for ch in self.cigar:
            if ch.isdigit():
                num = num * 10 + ord(ch) - 48
            else:
                yield (num, ch)
                num = 0


def _is_bam(raw):
    return raw[:2] == b"\x1f\x8b"


def read_alignment_file(path):
    """-> (header_text, [(name, length)], [Rec])  for a SAM text file or a BAM file."""
    with open(path, "rb") as fh:
        raw = fh.read()
    if _is_bam(raw):
        return _read_bam(gzip.decompress(raw))
    return _read_sam(raw.decode("ascii", errors="replace"))


def _sq_from_text(text):
    refs = []
    for line in text.splitlines():
        if line.startswith("@SQ"):
            name, ln = None, None
            for f in line.split("\t")[1:]:
                if f.startswith("SN:"):
                    name = f[3:]
                elif f.startswith("LN:"):
                    ln = int(f[3:])
            refs.append((name, ln))
    return refs


def _read_sam(text):
    header_lines, recs = [], []
    for line in text.split("\n"):
        if not line:
            continue
        if line[0] == "@":
            header_lines.append(line)
            continue
        f = line.split("\t")
        recs.append(Rec(f[0], int(f[1]), f[2], int(f[3]), f[5], f[9], None))
    header = "\n".join(header_lines) + ("\n" if header_lines else "")
    refs = _sq_from_text(header)
    ids = {n: i for i, (n, _) in enumerate(refs)}
    for r in recs:
        r.refid = ids.get(r.rname, -1)
    return header, refs, recs


def _read_bam(d):
    assert d[:4] == b"BAM\x01", "not a BAM stream"
    (l_text,) = struct.unpack_from("<i", d, 4)
    text = d[8 : 8 + l_text].split(b"\0", 1)[0].decode("ascii", errors="replace")
    o = 8 + l_text
    (n_ref,) = struct.unpack_from("<i", d, o)
    o += 4
    refs = []
    for _ in range(n_ref):
        (l_name,) = struct.unpack_from("<i", d, o)
        name = d[o + 4 : o + 4 + l_name - 1].decode()
        (l_ref,) = struct.unpack_from("<i", d, o + 4 + l_name)
        refs.append((name, l_ref))
        o += 8 + l_name
    recs = []
    n = len(d)
    while o + 4 <= n:
        (bs,) = struct.unpack_from("<i", d, o)
        refid, pos, l_rn, _mq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", d, o + 4)
        p = o + 36
        qname = d[p : p + l_rn - 1].decode()
        p += l_rn
        cig = struct.unpack_from("<%dI" % n_cig, d, p)
        p += 4 * n_cig
        cigar = "".join("%d%s" % (c >> 4, CIGOPS[c & 15]) for c in cig) if n_cig else "*"
        sb = d[p : p + (l_seq + 1) // 2]
        if l_seq:
            chars = []
            for b in sb:
                chars.append(NIB2CHR[b >> 4])
                chars.append(NIB2CHR[b & 15])
            seq = "".join(chars[:l_seq])
        else:
            seq = "*"
        rname = refs[refid][0] if refid >= 0 else "*"
        recs.append(Rec(qname, flag, rname, pos + 1, cigar, seq, refid))
        o += 4 + bs
    text_refs = _sq_from_text(text)
    return text, (text_refs if text_refs else refs), recs


def records_to_batch(refs, recs):
    """Pack records into the SoA batch layout of include/kindel_hip.h (kd_batch).

    Records whose rname is '*' are dropped (kindel/kindel.py:147-148).  SEQ '*' becomes
    seq_len 0.  Returns a dict of numpy arrays + 'contig_names', 'contig_lens'.
    """
    ids = {n: i for i, (n, _) in enumerate(refs)}
    contig, pos0, flag, seq_off, seq_len, cig_off, n_cig = [], [], [], [], [], [], []
    seq4 = bytearray()
    cigar = []
    for r in recs:
        if r.rname == "*":
            continue
        contig.append(ids[r.rname])
        pos0.append(r.pos - 1)
        flag.append(r.flag)
        s = "" if r.seq == "*" else r.seq
        seq_off.append(len(seq4))
        seq_len.append(len(s))
        nib = [_CHR2NIB[c] for c in s]
        if len(nib) & 1:
            nib.append(0)
        seq4.extend((nib[i] << 4) | nib[i + 1] for i in range(0, len(nib), 2))
        cig_off.append(len(cigar))
        k = 0
        if r.cigar != "*":
            num = 0
            for ch in r.cigar:
                if ch.isdigit():
                    num = num * 10 + ord(ch) - 48
                else:
                    cigar.append((num << 4) | CIGOPS.index(ch))
                    num = 0
                    k += 1
        n_cig.append(k)
    return dict(
        contig=np.asarray(contig, np.uint32),
        pos0=np.asarray(pos0, np.int32),
        flag=np.asarray(flag, np.uint32),
        seq_off=np.asarray(seq_off, np.uint64),
        seq_len=np.asarray(seq_len, np.uint32),
        cig_off=np.asarray(cig_off, np.uint64),
        n_cig=np.asarray(n_cig, np.uint32),
        seq4=np.frombuffer(bytes(seq4), np.uint8).copy(),
        cigar=np.asarray(cigar, np.uint32),
        contig_names=np.asarray([n for n, _ in refs]),
        contig_lens=np.asarray([l for _, l in refs], np.uint32),
    )


def load_batch(path):
    header, refs, recs = read_alignment_file(path)
    return records_to_batch(refs, recs)
