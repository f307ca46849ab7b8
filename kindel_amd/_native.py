"""ctypes binding of libkindel_hip.so (C-ABI: include/kindel_hip.h).

This is the only place the Python host code touches the native engine.  The library is the
HIP/gfx950 build kept in-tree next to this file; if it is missing or cannot be loaded the
import of the hot path fails loudly -- there is no CPU fallback in this package.

Reference correspondence: ``Engine`` is the device-side replacement of the body of
``parse_records`` / ``consensus_sequence`` (/root/reference/kindel/kindel.py:21-128, :384-430);
``decode_file`` replaces ``simplesam.Reader`` as used by ``parse_bam`` (:136-148).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libkindel_hip.so")

KD_E_UNSUPPORTED = -10
KD_OK, KD_E_BASE, KD_E_RANGE, KD_E_CIGAR, KD_E_HIP, KD_E_NOMEM, KD_E_ARG, KD_E_IO, KD_E_INTERNAL, KD_E_NOREF = (
    0, -1, -2, -3, -4, -5, -6, -7, -8, -9)
KD_MODE_AUTO, KD_MODE_GLOBAL, KD_MODE_WINDOW, KD_MODE_STRIP, KD_MODE_COOP = 0, 1, 2, 3, 4
(KD_CH_A, KD_CH_T, KD_CH_G, KD_CH_C, KD_CH_N, KD_CH_DEL, KD_CH_CSW, KD_CH_CEW, KD_CH_CLIP_STARTS,
 KD_CH_CLIP_ENDS, KD_CH_INS_TOTAL, KD_NCH) = (0, 1, 2, 3, 4, 5, 6, 11, 16, 17, 18, 19)

#: every symbol include/kindel_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = (
    "kd_abi_version kd_create kd_destroy kd_last_error kd_reset kd_set_mode kd_set_tuning kd_get_tuning kd_contig_base "
    "kd_total_sites kd_set_shard kd_push_batch kd_push_batch_device kd_sync kd_finalize kd_get_stats "
    "kd_get_batch_info kd_get_tables kd_get_insertions kd_consensus_run kd_consensus_fetch kd_consensus_fetch_all kd_consensus_device kd_changes_device kd_consensus_offsets kd_exchange_row kd_set_exchange "
    "kd_profile_enable kd_profile_get kd_profile_reset kd_decode_open kd_decode_batch kd_decode_n_contigs "
    "kd_decode_contig_name kd_decode_contig_len kd_decode_n_records kd_decode_close kd_decode_last_error "
    "kd_stream_open kd_stream_n_contigs kd_stream_contig_name kd_stream_contig_len kd_stream_next kd_stream_n_records "
    "kd_stream_last_error kd_stream_close kd_stream_set_contig_map kd_push_stream kd_decode_push_file kd_get_contig_first kd_host_threads kd_host_inflate kd_host_crc32 "
    "kd_bgzf_index kd_decode_open_span kd_step kd_finish kd_bgzf_plan_open kd_bgzf_plan_n_contigs kd_bgzf_plan_contig_name kd_bgzf_plan_contig_len "
    "kd_bgzf_plan_view kd_bgzf_plan_close kd_push_bam_gpu"
).split()

#: the reference exception each error code stands for (kindel.py:47,51-52,57,61,67,72,75,79)
_EXC = {KD_E_BASE: KeyError, KD_E_RANGE: IndexError, KD_E_CIGAR: RuntimeError, KD_E_NOMEM: MemoryError,
        KD_E_IO: OSError, KD_E_NOREF: KeyError}


class KindelNativeError(RuntimeError):
    pass


class kd_batch(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("contig", C.c_void_p), ("pos0", C.c_void_p), ("flag", C.c_void_p),
                ("seq_off", C.c_void_p), ("seq_len", C.c_void_p), ("cig_off", C.c_void_p), ("n_cig", C.c_void_p),
                ("seq4", C.c_void_p), ("seq4_bytes", C.c_uint64), ("cigar", C.c_void_p), ("cigar_words", C.c_uint64)]


_BATCH_FIELDS = (("contig", np.uint32), ("pos0", np.int32), ("flag", np.uint32), ("seq_off", np.uint64),
                 ("seq_len", np.uint32), ("cig_off", np.uint64), ("n_cig", np.uint32), ("seq4", np.uint8),
                 ("cigar", np.uint32))


class Library:
    """A loaded C-ABI library with typed prototypes."""

    def __init__(self, path=DEFAULT_LIB):
        if not os.path.exists(path):
            raise ImportError(
                "kindel_amd: native library %s not found. Build it with `python -c \"import __graft_entry__ as g; "
                "g.build()\"` (hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
        try:
            self.dll = C.CDLL(path)
        except OSError as e:
            raise ImportError("kindel_amd: cannot load %s: %s" % (path, e))
        self.path = path
        # where the contexts' "device" memory lives: the test emulator (tests/emu, the kernels as host code) keeps it in host memory
        self.emulated = hasattr(self.dll, "emu_switch")
        L, p, u32, u64 = self.dll, C.c_void_p, C.c_uint32, C.c_uint64
        L.kd_abi_version.restype = C.c_int
        L.kd_create.argtypes = [C.POINTER(p), C.c_int, u32, p, p]
        L.kd_destroy.argtypes = [p]
        L.kd_destroy.restype = None
        L.kd_last_error.argtypes = [p]
        L.kd_last_error.restype = C.c_char_p
        L.kd_reset.argtypes = [p]
        L.kd_set_mode.argtypes = [p, C.c_int]
        L.kd_set_tuning.argtypes = [p, u32, u32]
        L.kd_get_tuning.argtypes = [p, C.POINTER(C.c_uint32)]
        L.kd_contig_base.argtypes = [p, u32]
        L.kd_contig_base.restype = u64
        L.kd_total_sites.argtypes = [p]
        L.kd_total_sites.restype = u64
        L.kd_set_shard.argtypes = [p, u64, u64]
        L.kd_push_batch.argtypes = [p, C.POINTER(kd_batch)]
        L.kd_push_batch_device.argtypes = [p, C.POINTER(kd_batch)]
        L.kd_sync.argtypes = [p]
        L.kd_finalize.argtypes = [p, C.POINTER(u64)]
        L.kd_get_stats.argtypes = [p, p]
        L.kd_get_batch_info.argtypes = [p, p]
        L.kd_get_tables.argtypes = [p, u32, u32, p, p]
        L.kd_get_insertions.argtypes = [p, u32, C.POINTER(u64), C.POINTER(u64), p, p, p, p, p]
        L.kd_consensus_run.argtypes = [p, u32, u32, p, p]
        L.kd_consensus_fetch.argtypes = [p, u32, p, u64, C.POINTER(u64), p, p, p]
        L.kd_consensus_device.argtypes = [p, C.POINTER(p), C.POINTER(u64)]
        L.kd_changes_device.argtypes = [p, C.POINTER(p)]
        L.kd_exchange_row.argtypes = [p, p, u64, C.POINTER(u64)]
        L.kd_set_exchange.argtypes = [p, p, u64]
        L.kd_consensus_offsets.argtypes = [p, p, p]
        L.kd_consensus_fetch_all.argtypes = [p, p, u64, C.POINTER(u64), p, p]
        L.kd_step.argtypes = [p, C.POINTER(kd_batch), u32, p, u64, C.POINTER(u64), p]
        L.kd_finish.argtypes = [p, u32, p, u64, C.POINTER(u64), p]
        L.kd_profile_enable.argtypes = [p, C.c_int]
        L.kd_profile_get.argtypes = [p, C.POINTER(u32), p, p, p]
        L.kd_profile_reset.argtypes = [p]
        L.kd_bgzf_plan_open.argtypes = [C.POINTER(p), C.c_char_p]
        L.kd_bgzf_plan_n_contigs.argtypes = [p]
        L.kd_bgzf_plan_n_contigs.restype = u32
        L.kd_bgzf_plan_contig_name.argtypes = [p, u32]
        L.kd_bgzf_plan_contig_name.restype = C.c_char_p
        L.kd_bgzf_plan_contig_len.argtypes = [p, u32]
        L.kd_bgzf_plan_contig_len.restype = u32
        L.kd_bgzf_plan_view.argtypes = [p, C.POINTER(p), C.POINTER(u64), C.POINTER(p), C.POINTER(u32), C.POINTER(u64), C.POINTER(u64)]
        L.kd_bgzf_plan_close.argtypes = [p]
        L.kd_bgzf_plan_close.restype = None
        L.kd_push_bam_gpu.argtypes = [p, p, p]
        L.kd_decode_open.argtypes = [C.POINTER(p), C.c_char_p, C.c_int]
        L.kd_bgzf_index.argtypes = [C.c_char_p, C.POINTER(u64), p, u64]
        L.kd_decode_open_span.argtypes = [C.POINTER(p), C.c_char_p, C.c_int, u64, u64, p]
        L.kd_decode_batch.argtypes = [p]
        L.kd_decode_batch.restype = C.POINTER(kd_batch)
        L.kd_decode_n_contigs.argtypes = [p]
        L.kd_decode_n_contigs.restype = u32
        L.kd_decode_contig_name.argtypes = [p, u32]
        L.kd_decode_contig_name.restype = C.c_char_p
        L.kd_decode_contig_len.argtypes = [p, u32]
        L.kd_decode_contig_len.restype = u32
        L.kd_decode_n_records.argtypes = [p]
        L.kd_decode_n_records.restype = u64
        L.kd_decode_close.argtypes = [p]
        L.kd_decode_close.restype = None
        L.kd_decode_last_error.restype = C.c_char_p
        L.kd_stream_open.argtypes = [C.POINTER(p), C.c_char_p, C.c_int, u64]
        L.kd_stream_n_contigs.argtypes = [p]
        L.kd_stream_n_contigs.restype = u32
        L.kd_stream_contig_name.argtypes = [p, u32]
        L.kd_stream_contig_name.restype = C.c_char_p
        L.kd_stream_contig_len.argtypes = [p, u32]
        L.kd_stream_contig_len.restype = u32
        L.kd_stream_next.argtypes = [p, C.POINTER(C.POINTER(kd_batch))]
        L.kd_stream_n_records.argtypes = [p]
        L.kd_stream_n_records.restype = u64
        L.kd_stream_last_error.argtypes = [p]
        L.kd_stream_last_error.restype = C.c_char_p
        L.kd_stream_close.argtypes = [p]
        L.kd_stream_set_contig_map.argtypes = [p, p, u32]
        L.kd_stream_close.restype = None
        L.kd_push_stream.argtypes = [p, p, C.POINTER(u64)]
        L.kd_decode_push_file.argtypes = [p, C.c_char_p, C.c_int, u64, C.POINTER(u64)]
        L.kd_get_contig_first.argtypes = [p, p]
        if L.kd_abi_version() != 2:
            raise ImportError("kindel_amd: ABI version mismatch in %s" % path)


_default = None


def default_library():
    """The product library (HIP, gfx950).  Raises ImportError when it is not built."""
    global _default
    if _default is None:
        _default = Library(DEFAULT_LIB)
    return _default


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class _DecodedFile:
    """Owner of a kd_decode handle: closed when the last array viewing its buffers is gone."""

    def __init__(self, lib, handle):
        self.lib, self.handle = lib, handle

    def __del__(self):
        if self.handle is not None:
            self.lib.dll.kd_decode_close(self.handle)
            self.handle = None


class Stream:
    """A SAM / BAM file read in chunks (kd_stream_*): the header is known at once, the records arrive batch by batch."""

    def __init__(self, path, threads=0, chunk_bytes=0, lib=None):
        self.lib = lib or default_library()
        self._h = C.c_void_p()
        rc = self.lib.dll.kd_stream_open(C.byref(self._h), os.fsencode(str(path)), int(threads), int(chunk_bytes))
        if rc:
            self._h = None
            raise _EXC.get(rc, KindelNativeError)("%s: %s" % (path, self.lib.dll.kd_stream_last_error(None).decode()))
        n = self.lib.dll.kd_stream_n_contigs(self._h)
        self.contig_names = [self.lib.dll.kd_stream_contig_name(self._h, i).decode() for i in range(n)]
        self.contig_lens = np.asarray([self.lib.dll.kd_stream_contig_len(self._h, i) for i in range(n)], np.uint32)

    def _raise(self, rc):
        msg = self.lib.dll.kd_stream_last_error(self._h).decode()
        if rc == KD_E_NOREF:
            raise KeyError(msg)
        raise _EXC.get(rc, KindelNativeError)(msg)

    def next_batch(self):
        """-> batch dict of COPIES (numpy), or None at the end of the file"""
        b = C.POINTER(kd_batch)()
        rc = self.lib.dll.kd_stream_next(self._h, C.byref(b))
        if rc:
            self._raise(rc)
        if not b:
            return None
        b = b.contents
        n = int(b.n_reads)
        sizes = dict(contig=n, pos0=n, flag=n, seq_off=n, seq_len=n, cig_off=n, n_cig=n, seq4=int(b.seq4_bytes) + 8,
                     cigar=int(b.cigar_words) + 2)
        out = {}
        for name, dt in _BATCH_FIELDS:
            m = sizes[name]
            addr = getattr(b, name)
            if m and addr:
                buf = (C.c_char * (m * np.dtype(dt).itemsize)).from_address(addr)
                out[name] = np.frombuffer(buf, dtype=dt, count=m).copy()
            else:
                out[name] = np.zeros(0, dt)
        out["contig_names"] = np.asarray(self.contig_names)
        out["contig_lens"] = self.contig_lens
        return out

    def set_contig_map(self, mapping):
        """Records come out with contig = mapping[refID] from the next batch on (uint32 per @SQ line, 0xffffffff = no record may
        lie there); None restores the identity.  kd_stream_set_contig_map."""
        m = np.zeros(0, np.uint32) if mapping is None else np.ascontiguousarray(mapping, np.uint32)
        rc = self.lib.dll.kd_stream_set_contig_map(self._h, _ptr(m) if len(m) else None, len(m))
        if rc:
            self._raise(rc)

    def n_records(self):
        return int(self.lib.dll.kd_stream_n_records(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self.lib.dll.kd_stream_close(self._h)
            self._h = None

    __del__ = close


_tools = None


def tools_library():
    """TEST / BENCH TOOLS (tools/libkindel_tools.so: the BAM writer), built by __graft_entry__.build(); not the product library."""
    global _tools
    if _tools is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "libkindel_tools.so")
        if not os.path.exists(path):
            raise ImportError("%s not built: python -c 'import __graft_entry__ as g; g.build()'" % path)
        _tools = C.CDLL(path)
        _tools.kd_write_bam.argtypes = [C.c_char_p, C.POINTER(kd_batch), C.c_uint32, C.POINTER(C.c_char_p), C.c_void_p, C.c_char_p, C.c_int, C.c_int]
        _tools.kd_tools_last_error.restype = C.c_char_p
    return _tools


def write_bam(path, batch, names=None, sort_order="coordinate", threads=0, level=1, lib=None):
    """Host batch (dict of numpy arrays) -> BGZF-compressed BAM through the native writer of the TOOLS library (parallel deflate;
    test / bench inputs).  `lib` is ignored (kept for the callers that pass the product / emulator library around)."""
    dll = tools_library()
    arrs = {name: np.ascontiguousarray(batch[name], dt) for name, dt in _BATCH_FIELDS}
    b = kd_batch()
    b.n_reads = len(arrs["contig"])
    for name, _ in _BATCH_FIELDS:
        setattr(b, name, arrs[name].ctypes.data)
    b.seq4_bytes, b.cigar_words = arrs["seq4"].size, arrs["cigar"].size
    lens = np.ascontiguousarray(batch["contig_lens"], np.uint32)
    names = [str(x) for x in (names if names is not None else batch.get("contig_names", ["ctg%d" % i for i in range(len(lens))]))]
    cnames = (C.c_char_p * max(len(names), 1))(*[n.encode() for n in names])
    rc = dll.kd_write_bam(os.fsencode(str(path)), C.byref(b), len(lens), cnames, _ptr(lens), sort_order.encode(), int(threads), int(level))
    if rc:
        raise _EXC.get(rc, KindelNativeError)("%s: %s" % (path, dll.kd_tools_last_error().decode()))


def host_threads(lib=None):
    """Host threads the native decoder uses by default: visible cores capped by the cgroup CPU quota."""
    lib = lib or default_library()
    lib.dll.kd_host_threads.restype = C.c_uint32
    return int(lib.dll.kd_host_threads())


def host_crc32(data, lib=None):
    """CRC-32 of `data` as the BGZF reader computes it (kd_host_crc32; zlib's convention)."""
    lib = lib or default_library()
    f = lib.dll.kd_host_crc32
    f.argtypes = [C.c_char_p, C.c_uint64]
    f.restype = C.c_uint32
    return int(f(bytes(data), len(data)))


def host_inflate(data, out_len, lib=None):
    """Raw DEFLATE -> bytes of exactly `out_len` (the BGZF reader's block decoder, kd_host_inflate); ValueError if malformed."""
    lib = lib or default_library()
    src = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(0, np.uint8)
    out = np.empty(int(out_len) + 64, np.uint8)
    out[int(out_len):] = 0xA5          # canary: the decoder never writes past out_len
    f = lib.dll.kd_host_inflate
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    rc = f(src.ctypes.data if len(src) else None, len(src), out.ctypes.data, int(out_len))
    if not (out[int(out_len):] == 0xA5).all():
        raise AssertionError("kd_host_inflate wrote past the end of its output")
    if rc != 0:
        raise ValueError("kd_host_inflate: malformed DEFLATE stream (rc %d)" % rc)
    return out[:int(out_len)].tobytes()


def bgzf_index(path, lib=None):
    """-> uint64[n_blocks]: compressed file offset of every BGZF block's data (kd_bgzf_index); OSError if the file is not BGZF"""
    lib = lib or default_library()
    n = C.c_uint64(0)
    rc = lib.dll.kd_bgzf_index(os.fsencode(str(path)), C.byref(n), None, 0)
    if rc:
        raise _EXC.get(rc, KindelNativeError)("%s: %s" % (path, lib.dll.kd_decode_last_error().decode()))
    off = np.zeros(n.value, np.uint64)
    rc = lib.dll.kd_bgzf_index(os.fsencode(str(path)), C.byref(n), _ptr(off), off.size)
    if rc:
        raise _EXC.get(rc, KindelNativeError)("%s: %s" % (path, lib.dll.kd_decode_last_error().decode()))
    return off


def _batch_of_handle(lib, h):
    owner = _DecodedFile(lib, h)
    b = lib.dll.kd_decode_batch(h).contents
    n = int(b.n_reads)
    sizes = dict(contig=n, pos0=n, flag=n, seq_off=n, seq_len=n, cig_off=n, n_cig=n,
                 seq4=int(b.seq4_bytes), cigar=int(b.cigar_words))
    out = {}
    for name, dt in _BATCH_FIELDS:
        cnt = sizes[name]
        addr = getattr(b, name)
        if cnt and addr:
            buf = (C.c_char * (cnt * np.dtype(dt).itemsize)).from_address(addr)
            buf._kd_owner = owner                      # array -> .base (buf) -> owner -> kd_decode_close
            a = np.frombuffer(buf, dtype=dt, count=cnt)
            a.flags.writeable = False
            out[name] = a
        else:
            out[name] = np.zeros(0, dt)
    nc = lib.dll.kd_decode_n_contigs(h)
    out["contig_names"] = np.asarray([lib.dll.kd_decode_contig_name(h, i).decode() for i in range(nc)])
    out["contig_lens"] = np.asarray([lib.dll.kd_decode_contig_len(h, i) for i in range(nc)], np.uint32)
    out["n_records"] = int(lib.dll.kd_decode_n_records(h))
    return out


def decode_span(path, block_lo, block_hi, threads=0, lib=None):
    """The records that begin in BGZF blocks [block_lo, block_hi) of a BAM file (kd_decode_open_span) -> batch dict as
    decode_file() plus span = (start offset, end offset, records, reaches_eof).  OSError when the record chain of the span does
    not end on the next span's start (the caller falls back to the whole file)."""
    lib = lib or default_library()
    h = C.c_void_p()
    info = np.zeros(4, np.uint64)
    rc = lib.dll.kd_decode_open_span(C.byref(h), os.fsencode(str(path)), int(threads), int(block_lo), int(block_hi), _ptr(info))
    if rc == KD_E_NOREF:
        raise KeyError(lib.dll.kd_decode_last_error().decode())
    if rc:
        raise _EXC.get(rc, KindelNativeError)("%s: %s" % (path, lib.dll.kd_decode_last_error().decode()))
    out = _batch_of_handle(lib, h)
    out["span"] = tuple(int(x) for x in info)
    return out


def decode_file(path, threads=0, lib=None):
    """SAM/BAM file -> SoA batch dict (numpy, host) via the native decoder.

    Keys: contig,pos0,flag,seq_off,seq_len,cig_off,n_cig,seq4,cigar + contig_names, contig_lens,
    n_records.  Records with RNAME '*' are dropped (kindel.py:147-148).
    The arrays are read-only VIEWS of the decoder's own buffers (no copy: at a few 10^6 reads/s the copies cost as
    much as the decode); the decoder handle lives as long as any of them does."""
    lib = lib or default_library()
    h = C.c_void_p()
    rc = lib.dll.kd_decode_open(C.byref(h), os.fsencode(str(path)), int(threads))
    if rc == KD_E_NOREF:   # refs_lens[ref_id] of kindel.py:151: KeyError(<the unknown reference name>)
        raise KeyError(lib.dll.kd_decode_last_error().decode())
    if rc:
        raise _EXC.get(rc, KindelNativeError)("%s: %s" % (path, lib.dll.kd_decode_last_error().decode()))
    owner = _DecodedFile(lib, h)
    b = lib.dll.kd_decode_batch(h).contents
    n = int(b.n_reads)
    sizes = dict(contig=n, pos0=n, flag=n, seq_off=n, seq_len=n, cig_off=n, n_cig=n,
                 seq4=int(b.seq4_bytes), cigar=int(b.cigar_words))
    out = {}
    for name, dt in _BATCH_FIELDS:
        cnt = sizes[name]
        addr = getattr(b, name)
        if cnt and addr:
            buf = (C.c_char * (cnt * np.dtype(dt).itemsize)).from_address(addr)
            buf._kd_owner = owner                      # array -> .base (buf) -> owner -> kd_decode_close
            a = np.frombuffer(buf, dtype=dt, count=cnt)
            a.flags.writeable = False
            out[name] = a
        else:
            out[name] = np.zeros(0, dt)
    nc = lib.dll.kd_decode_n_contigs(h)
    out["contig_names"] = np.asarray([lib.dll.kd_decode_contig_name(h, i).decode() for i in range(nc)])
    out["contig_lens"] = np.asarray([lib.dll.kd_decode_contig_len(h, i) for i in range(nc)], np.uint32)
    out["n_records"] = int(lib.dll.kd_decode_n_records(h))
    return out


class UnsupportedByGpuIngest(KindelNativeError):
    """The device-side ingest cannot read this file (SAM text, plain gzip, CG-tag CIGARs, a record chain it could not verify):
    the host decoder does."""


class BgzfPlan:
    """The host's share of the device-side ingest (kd_bgzf_plan_*): the BAM file mapped, its BGZF block table, its header."""

    def __init__(self, path, lib=None):
        self.lib = lib or default_library()
        h = C.c_void_p()
        rc = self.lib.dll.kd_bgzf_plan_open(C.byref(h), os.fsencode(str(path)))
        if rc:
            msg = "%s: %s" % (path, self.lib.dll.kd_decode_last_error().decode())
            raise (UnsupportedByGpuIngest if rc == KD_E_UNSUPPORTED else _EXC.get(rc, KindelNativeError))(msg)
        self._h = h
        n = self.lib.dll.kd_bgzf_plan_n_contigs(h)
        self.contig_names = [self.lib.dll.kd_bgzf_plan_contig_name(h, i).decode() for i in range(n)]
        self.contig_lens = np.asarray([self.lib.dll.kd_bgzf_plan_contig_len(h, i) for i in range(n)], np.uint32)

    def close(self):
        if self._h:
            self.lib.dll.kd_bgzf_plan_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Engine:
    """One kd_ctx: device tables for a set of contigs on one GPU."""

    def __init__(self, contig_lens, device=0, stream=None, lib=None, mode=KD_MODE_AUTO):
        self.lib = lib or default_library()
        self.contig_lens = np.ascontiguousarray(contig_lens, np.uint32)
        self.device_index = int(device)
        self._h = C.c_void_p()
        rc = self.lib.dll.kd_create(C.byref(self._h), int(device), len(self.contig_lens), _ptr(self.contig_lens),
                                    C.c_void_p(stream) if stream else None)
        if rc:
            self._h = None
            raise _EXC.get(rc, KindelNativeError)(
                "kd_create failed (%d): %s" % (rc, self.lib.dll.kd_last_error(None).decode()))
        if mode != KD_MODE_AUTO:
            self.set_mode(mode)
        self._keep = None

    # -- plumbing --
    def _check(self, rc, what):
        if rc:
            msg = "%s failed (%d): %s" % (what, rc, self.lib.dll.kd_last_error(self._h).decode())
            raise _EXC.get(rc, KindelNativeError)(msg)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.dll.kd_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_mode(self, mode):
        self._check(self.lib.dll.kd_set_mode(self._h, mode), "kd_set_mode")

    def set_tuning(self, window_sites=0, slice_reads=0):
        self._check(self.lib.dll.kd_set_tuning(self._h, window_sites, slice_reads), "kd_set_tuning")

    def tuning(self):
        """-> (sites per LDS window, reads per work item or 0 = chosen per batch) in effect"""
        out = (C.c_uint32 * 2)()
        self._check(self.lib.dll.kd_get_tuning(self._h, out), "kd_get_tuning")
        return int(out[0]), int(out[1])

    def set_shard(self, g_lo, g_hi):
        self._check(self.lib.dll.kd_set_shard(self._h, g_lo, g_hi), "kd_set_shard")
        self._shard = (int(g_lo), int(g_hi))

    @property
    def memory_device(self):
        """torch device string of the memory this context's device pointers point into ("cuda:<index>"; "cpu" under the test emulator)."""
        return "cpu" if self.lib.emulated else "cuda:%d" % self.device_index

    def shard_interval(self):
        """The emit interval [g_lo, g_hi) of this context (all of G-space unless set_shard was called)."""
        return getattr(self, "_shard", None) or (0, self.total_sites())

    def reset(self):
        self._check(self.lib.dll.kd_reset(self._h), "kd_reset")

    def contig_base(self, c):
        return int(self.lib.dll.kd_contig_base(self._h, c))

    def total_sites(self):
        return int(self.lib.dll.kd_total_sites(self._h))

    # -- pileup --
    @staticmethod
    def _struct(arrs, n):
        b = kd_batch()
        b.n_reads = n
        for name, _ in _BATCH_FIELDS:
            setattr(b, name, arrs[name])
        return b

    def push(self, batch):
        """Host batch (dict of numpy arrays, see decode_file)."""
        arrs = {name: np.ascontiguousarray(batch[name], dt) for name, dt in _BATCH_FIELDS}
        b = self._struct({k: v.ctypes.data for k, v in arrs.items()}, len(arrs["contig"]))
        b.seq4_bytes = arrs["seq4"].size
        b.cigar_words = arrs["cigar"].size
        self._check(self.lib.dll.kd_push_batch(self._h, C.byref(b)), "kd_push_batch")

    def push_device(self, ptrs, n_reads, seq4_bytes, cigar_words):
        """Device-resident batch: ptrs maps field name -> device address (e.g. tensor.data_ptr()).  The library runs on its own
        stream: the arrays must be complete (torch.cuda.synchronize() after the kernels that wrote them) when this is called."""
        b = self._struct(ptrs, n_reads)
        b.seq4_bytes, b.cigar_words = seq4_bytes, cigar_words
        self._check(self.lib.dll.kd_push_batch_device(self._h, C.byref(b)), "kd_push_batch_device")

    def step_device(self, ptrs, n_reads, seq4_bytes, cigar_words, out, min_depth=1):
        """One whole step over a device-resident batch (kd_step: reset + record loop + insertion reduction + consensus + all
        contigs' consensus bytes into `out`, ideally pinned).  -> contig_off uint64[n_contigs + 1]"""
        # (a step loop calls this with the same batch again and again: the kd_batch struct is kept while the addresses and sizes are the same)
        key = (tuple(ptrs[name] for name, _ in _BATCH_FIELDS), n_reads, seq4_bytes, cigar_words)
        if getattr(self, "_step_key", None) != key:
            b = self._struct(ptrs, n_reads)
            b.seq4_bytes, b.cigar_words = seq4_bytes, cigar_words
            self._step_key, self._step_struct = key, b
        b = self._step_struct
        ln = C.c_uint64(0)
        off = np.zeros(len(self.contig_lens) + 1, np.uint64)
        self._n_patches = 0
        self._check(self.lib.dll.kd_step(self._h, C.byref(b), int(min_depth), _ptr(out), out.size, C.byref(ln), _ptr(off)), "kd_step")
        return off

    def finish(self, out, min_depth=1):
        """Everything behind the pushes in one call and one host round trip (kd_finish): insertion reduction, consensus of all
        contigs, their bytes into `out`.  -> contig_off uint64[n_contigs + 1].  Raises like finalize()."""
        ln = C.c_uint64(0)
        off = np.zeros(len(self.contig_lens) + 1, np.uint64)
        self._n_patches = 0
        self._check(self.lib.dll.kd_finish(self._h, int(min_depth), _ptr(out), out.size, C.byref(ln), _ptr(off)), "kd_finish")
        return off

    def push_stream(self, stream):
        """Every remaining batch of a Stream: decode of batch k+1 overlapped with copy + kernels of batch k.
        -> dict(batches, decode_s, push_s, wall_s)"""
        st = (C.c_uint64 * 4)()
        rc = self.lib.dll.kd_push_stream(self._h, stream._h, st)
        if rc == KD_E_NOREF:
            raise KeyError(self.lib.dll.kd_last_error(self._h).decode())
        self._check(rc, "kd_push_stream")
        return dict(batches=int(st[0]), decode_s=st[1] / 1e6, push_s=st[2] / 1e6, wall_s=st[3] / 1e6)

    def push_bam_gpu(self, plan):
        """The whole BAM file of a BgzfPlan through the DEVICE-side ingest (kd_push_bam_gpu): BGZF inflate, record walk and the
        batch arrays on the GPU, then the pileup.  UnsupportedByGpuIngest: read the file with the host decoder instead.
        -> dict(records, kept, inflated_bytes, blocks, ingest_s, push_s)"""
        st = (C.c_uint64 * 8)()
        rc = self.lib.dll.kd_push_bam_gpu(self._h, plan._h, st)
        if rc == KD_E_UNSUPPORTED:
            raise UnsupportedByGpuIngest(self.lib.dll.kd_last_error(self._h).decode())
        self._check(rc, "kd_push_bam_gpu")
        return dict(records=int(st[0]), kept=int(st[1]), inflated_bytes=int(st[2]), blocks=int(st[3]), ingest_s=st[5] / 1e6, push_s=st[6] / 1e6)

    def contig_first(self):
        """uint64[n_contigs]: index of each contig's first record over all pushed records, 2^64-1 = none"""
        out = np.zeros(len(self.contig_lens), np.uint64)
        self._check(self.lib.dll.kd_get_contig_first(self._h, _ptr(out)), "kd_get_contig_first")
        return out

    def sync(self):
        self._check(self.lib.dll.kd_sync(self._h), "kd_sync")

    def finalize(self):
        bad = C.c_uint64(0)
        self._check(self.lib.dll.kd_finalize(self._h, C.byref(bad)), "kd_finalize")

    def stats(self):
        out = np.zeros(4, np.uint64)
        self._check(self.lib.dll.kd_get_stats(self._h, _ptr(out)), "kd_get_stats")
        return dict(reads=int(out[0]), aligned=int(out[1]), walked=int(out[2]), ins_events=int(out[3]))

    def batch_info(self):
        out = np.zeros(8, np.uint64)
        self._check(self.lib.dll.kd_get_batch_info(self._h, _ptr(out)), "kd_get_batch_info")
        keys = ("windowed", "regular", "cold", "irregular", "long_cigar", "work_items", "max_span", "unsorted")
        return dict(zip(keys, (int(x) for x in out)))

    # -- tables --
    def tables(self, contig, channels=None):
        """-> uint32 array [len(channels), L+1] (row i = channel channels[i])."""
        ch = np.arange(KD_NCH, dtype=np.uint32) if channels is None else np.ascontiguousarray(channels, np.uint32)
        L1 = int(self.contig_lens[contig]) + 1
        out = np.zeros((len(ch), L1), np.uint32)
        self._check(self.lib.dll.kd_get_tables(self._h, contig, len(ch), _ptr(ch), _ptr(out)), "kd_get_tables")
        return out

    def insertions(self, contig):
        """-> (site[u32], count[u32], strings[list of str]) of the insertion dicts of one contig."""
        nk, nb = C.c_uint64(0), C.c_uint64(0)
        self._check(self.lib.dll.kd_get_insertions(self._h, contig, C.byref(nk), C.byref(nb), None, None, None,
                                                   None, None), "kd_get_insertions")
        n = nk.value
        site, count, ln = (np.zeros(n, np.uint32) for _ in range(3))
        off = np.zeros(n, np.uint64)
        byts = np.zeros(max(nb.value, 1), np.uint8)
        if n:
            self._check(self.lib.dll.kd_get_insertions(self._h, contig, C.byref(nk), C.byref(nb), _ptr(site),
                                                       _ptr(count), _ptr(ln), _ptr(off), _ptr(byts)),
                        "kd_get_insertions")
        raw = byts.tobytes()
        strings = [raw[int(o):int(o) + int(l)].decode() for o, l in zip(off, ln)]
        return site, count, strings

    # -- consensus --
    def consensus_run(self, min_depth=1, patches=()):
        """patches: iterable of (g_start, g_end) skip ranges in G-space."""
        ps = np.asarray([p[0] for p in patches], np.uint64)
        pe = np.asarray([p[1] for p in patches], np.uint64)
        self._n_patches = len(ps)
        self._check(self.lib.dll.kd_consensus_run(self._h, int(min_depth), len(ps), _ptr(ps), _ptr(pe)),
                    "kd_consensus_run")

    def consensus_fetch_into(self, contig, out):
        """Copy contig's consensus bytes into `out` (a uint8 numpy array, ideally backed by pinned host memory so
        the device-to-host copy needs no staging); -> number of bytes."""
        ln = C.c_uint64(0)
        self._check(self.lib.dll.kd_consensus_fetch(self._h, contig, _ptr(out), out.size, C.byref(ln), None, None, None),
                    "kd_consensus_fetch")
        return ln.value

    def consensus_fetch_all_into(self, out, changes=None):
        """All contigs' consensus bytes in one device-to-host copy into `out` (uint8, ideally pinned);
        `changes` (uint8[total_sites], G-space) optional.  -> contig_off uint64[n_contigs + 1]."""
        ln = C.c_uint64(0)
        off = np.zeros(len(self.contig_lens) + 1, np.uint64)
        self._check(self.lib.dll.kd_consensus_fetch_all(self._h, _ptr(out), out.size, C.byref(ln), _ptr(off),
                                                        _ptr(changes) if changes is not None else None),
                    "kd_consensus_fetch_all")
        return off

    def consensus_fetch(self, contig, want_changes=True):
        """-> (bytes, changes uint8[L] | None, (min_depth, max_depth), patch_off uint64[n_patches])."""
        ln = C.c_uint64(0)
        self._check(self.lib.dll.kd_consensus_fetch(self._h, contig, None, 0, C.byref(ln), None, None, None),
                    "kd_consensus_fetch")
        L = int(self.contig_lens[contig])
        seq = np.zeros(max(ln.value, 1), np.uint8)
        changes = np.zeros(max(L, 1), np.uint8) if want_changes else None
        mm = np.zeros(2, np.uint32)
        poff = np.zeros(max(self._n_patches, 1), np.uint64)
        self._check(self.lib.dll.kd_consensus_fetch(self._h, contig, _ptr(seq), seq.size, C.byref(ln),
                                                    _ptr(changes) if want_changes else None, _ptr(mm), _ptr(poff)),
                    "kd_consensus_fetch")
        return (seq[: ln.value].tobytes(), changes[:L] if want_changes else None, (int(mm[0]), int(mm[1])),
                poff[: self._n_patches])

    def consensus_meta(self, contig):
        """Host-side results of the last run for one contig, no device traffic:
        -> (n_bytes, (min_depth, max_depth), patch_off uint64[n_patches])."""
        ln = C.c_uint64(0)
        mm = np.zeros(2, np.uint32)
        poff = np.zeros(max(self._n_patches, 1), np.uint64)
        self._check(self.lib.dll.kd_consensus_fetch(self._h, contig, None, 0, C.byref(ln), None, _ptr(mm), _ptr(poff)),
                    "kd_consensus_fetch")
        return ln.value, (int(mm[0]), int(mm[1])), poff[: self._n_patches]

    def consensus_device(self):
        p, n = C.c_void_p(), C.c_uint64(0)
        self._check(self.lib.dll.kd_consensus_device(self._h, C.byref(p), C.byref(n)), "kd_consensus_device")
        return p.value, n.value

    def changes_device(self):
        p = C.c_void_p()
        self._check(self.lib.dll.kd_changes_device(self._h, C.byref(p)), "kd_changes_device")
        return p.value

    def exchange_row(self, dev_ptr, cap):
        """This shard's exchange row (kindel_hip.h: kd_exchange_row) into `cap` bytes of device memory at dev_ptr -> row bytes
        (> cap: it did not fit, only the header was written)."""
        n = C.c_uint64(0)
        self._check(self.lib.dll.kd_exchange_row(self._h, C.c_void_p(dev_ptr), cap, C.byref(n)), "kd_exchange_row")
        return n.value

    def set_exchange(self, dev_ptr, cap):
        """Register (dev_ptr = 0 / None: unregister) the device buffer kd_finish / kd_step leave the exchange row in."""
        self._check(self.lib.dll.kd_set_exchange(self._h, C.c_void_p(dev_ptr or None), cap if dev_ptr else 0), "kd_set_exchange")

    def consensus_offsets(self):
        """-> (contig_off uint64[n+1], depth_minmax uint32[n,2]) of the last consensus_run"""
        n = len(self.contig_lens)
        off = np.zeros(n + 1, np.uint64)
        mm = np.zeros((n, 2), np.uint32)
        self._check(self.lib.dll.kd_consensus_offsets(self._h, _ptr(off), _ptr(mm)), "kd_consensus_offsets")
        return off, mm

    # -- profiling --
    def profile_enable(self, on=True):   # True / 1: every launch; 2: only k_window's launches; False / 0: off
        self._check(self.lib.dll.kd_profile_enable(self._h, int(on)), "kd_profile_enable")

    def profile_reset(self):
        self._check(self.lib.dll.kd_profile_reset(self._h), "kd_profile_reset")

    def profile(self):
        """-> {kernel name: (launches, total_ms)}"""
        n = C.c_uint32(0)
        self._check(self.lib.dll.kd_profile_get(self._h, C.byref(n), None, None, None), "kd_profile_get")
        k = n.value
        names = C.create_string_buffer(64 * max(k, 1))
        launches = np.zeros(max(k, 1), np.uint64)
        ms = np.zeros(max(k, 1), np.float64)
        n = C.c_uint32(k)
        self._check(self.lib.dll.kd_profile_get(self._h, C.byref(n), names, _ptr(launches), _ptr(ms)),
                    "kd_profile_get")
        out = {}
        for i in range(min(k, n.value)):
            nm = names.raw[i * 64:(i + 1) * 64].split(b"\0", 1)[0].decode()
            out[nm] = (int(launches[i]), float(ms[i]))
        return out
