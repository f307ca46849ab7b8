"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of oracle/kindel_oracle.c.

Mirrors the shapes of the reference's ``parse_records`` / ``consensus_sequence``
(/root/reference/kindel/kindel.py:21-128, :384-430) on the SoA read batch of
include/kindel_hip.h.  Used by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by kindel_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libkindel_oracle.so")

KO_E_BASE, KO_E_RANGE, KO_E_CIGAR, KO_E_PATCH, KO_E_NOMEM = -1, -2, -3, -4, -5
_EXC = {KO_E_BASE: KeyError, KO_E_RANGE: IndexError, KO_E_CIGAR: RuntimeError,
        KO_E_PATCH: AttributeError, KO_E_NOMEM: MemoryError}


def build(force=False):
    src = os.path.join(_HERE, "kindel_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libkindel_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        p = C.c_void_p
        L.ko_parse_records.restype = p
        L.ko_parse_records.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64] + [p] * 9 + [
            C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
        L.ko_free.argtypes = [p]
        for f in ("ko_weights", "ko_clip_start_weights", "ko_clip_end_weights", "ko_clip_starts",
                  "ko_clip_ends", "ko_deletions"):
            getattr(L, f).restype = C.POINTER(C.c_uint32)
            getattr(L, f).argtypes = [p]
        L.ko_ins_n.restype = C.c_uint64
        L.ko_ins_n.argtypes = [p]
        L.ko_ins_bytes.restype = C.c_uint64
        L.ko_ins_bytes.argtypes = [p]
        L.ko_ins_enumerate.argtypes = [p] * 6
        L.ko_ins_totals.argtypes = [p, p]
        L.ko_stats.argtypes = [p, p]
        L.ko_derived.argtypes = [p] * 6
        L.ko_depth_minmax.argtypes = [p, p]
        L.ko_consensus_sequence.restype = C.c_int
        L.ko_consensus_sequence.argtypes = [p, C.c_uint32, C.c_int, p, p, p, C.c_int, C.c_int,
                                            p, C.c_uint64, C.POINTER(C.c_uint64), p]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleAln:
    """Tables of one contig, reference-shaped: weights[L,5] in A,T,G,C,N order etc."""

    def __init__(self, handle, L):
        self._h, self.L = handle, L
        lb = lib()

        def arr(fn, n):
            return np.ctypeslib.as_array(getattr(lb, fn)(handle), shape=(n,)).copy()

        self.weights = arr("ko_weights", L * 5).reshape(L, 5)
        self.clip_start_weights = arr("ko_clip_start_weights", L * 5).reshape(L, 5)
        self.clip_end_weights = arr("ko_clip_end_weights", L * 5).reshape(L, 5)
        self.clip_starts = arr("ko_clip_starts", L + 1)
        self.clip_ends = arr("ko_clip_ends", L + 1)
        self.deletions = arr("ko_deletions", L + 1)
        n, nb = lb.ko_ins_n(handle), lb.ko_ins_bytes(handle)
        sites = np.zeros(n, np.uint32)
        counts = np.zeros(n, np.uint32)
        lens = np.zeros(n, np.uint32)
        offs = np.zeros(n, np.uint64)
        byts = np.zeros(max(nb, 1), np.uint8)
        lb.ko_ins_enumerate(handle, _ptr(sites), _ptr(counts), _ptr(lens), _ptr(offs), _ptr(byts))
        raw = byts.tobytes()
        #: list of (site, string, count) -- site ascending, dict insertion order within a site
        self.insertions = [(int(s), raw[int(o):int(o) + int(l)].decode(), int(c))
                           for s, c, l, o in zip(sites, counts, lens, offs)]
        self.ins_totals = np.zeros(L + 1, np.uint32)
        lb.ko_ins_totals(handle, _ptr(self.ins_totals))
        st = np.zeros(3, np.uint64)
        lb.ko_stats(handle, _ptr(st))
        self.n_reads_used, self.n_events_aligned, self.n_events_walked = (int(x) for x in st)

    def derived(self):
        out = [np.zeros(self.L, np.uint32) for _ in range(5)]
        lib().ko_derived(self._h, *[_ptr(o) for o in out])
        return dict(zip(("aligned_depth", "consensus_depth", "clip_start_depth",
                         "clip_end_depth", "clip_depth"), out))

    def depth_minmax(self):
        mm = np.zeros(2, np.uint32)
        lib().ko_depth_minmax(self._h, _ptr(mm))
        return int(mm[0]), int(mm[1])

    def consensus_sequence(self, cdr_patches=None, trim_ends=False, min_depth=1, uppercase=False):
        """-> (str, changes list of None|'D'|'N'|'I')  -- kindel.py:384-430.
        cdr_patches: iterable of objects/tuples with (start, end, seq)."""
        pl = [(p[0], p[1], p[2]) if isinstance(p, tuple) else (p.start, p.end, p.seq)
              for p in (cdr_patches or [])]
        n = len(pl)
        ps = np.asarray([p[0] for p in pl], np.int64)
        pe = np.asarray([p[1] for p in pl], np.int64)
        seqs = (C.c_char_p * max(n, 1))(*[None if p[2] is None else p[2].encode() for p in pl])
        cap = self.L + sum(len(s) * c for _, s, c in self.insertions) + \
            sum(len(p[2] or "") for p in pl) + 16
        out = C.create_string_buffer(cap)
        olen = C.c_uint64(0)
        changes = np.zeros(max(self.L, 1), np.uint8)
        rc = lib().ko_consensus_sequence(self._h, min_depth, n, _ptr(ps), _ptr(pe), seqs,
                                         int(trim_ends), int(uppercase), out, cap,
                                         C.byref(olen), _ptr(changes))
        if rc:
            raise _EXC[rc]("oracle consensus_sequence failed (%d)" % rc)
        ch = [None if c == 0 else chr(c) for c in changes[: self.L]]
        return out.raw[: olen.value].decode(), ch

    def __del__(self):
        try:
            lib().ko_free(self._h)
        except Exception:
            pass


def parse_records(batch, contig_id):
    """Oracle tables for contig ``contig_id`` of an SoA batch (dict of numpy arrays)."""
    err = C.c_int(0)
    err_read = C.c_uint64(0)
    L = int(batch["contig_lens"][contig_id])
    keep = [np.ascontiguousarray(batch[k]) for k in
            ("contig", "pos0", "flag", "seq_off", "seq_len", "cig_off", "n_cig", "seq4", "cigar")]
    h = lib().ko_parse_records(contig_id, L, len(keep[0]), *[_ptr(a) for a in keep],
                               C.byref(err), C.byref(err_read))
    if not h:
        raise _EXC[err.value]("oracle parse_records: read %d" % err_read.value)
    return OracleAln(h, L)


def contig_order(batch):
    """Contig ids in order of first appearance in the batch (kindel.py:143-151)."""
    c = np.asarray(batch["contig"])
    if c.size == 0:
        return []
    _, first = np.unique(c, return_index=True)
    return [int(c[i]) for i in sorted(first)]


def parse_bam_batch(batch):
    """-> {contig_id: OracleAln} in first-appearance order (parse_bam, kindel.py:131-153)."""
    return {cid: parse_records(batch, cid) for cid in contig_order(batch)}
