"""Host-side mirror of the reference module ``kindel.kindel`` on top of the HIP engine.

Same function names, arguments, defaults, return shapes and exceptions as
/root/reference/kindel/kindel.py, so ``from kindel_amd import kindel`` is a drop-in for
``from kindel import kindel``:

    parse_bam            :131-153   decode (native C++) -> device pileup -> ``alignment`` tuples
    parse_records        :21-128    same, from an iterable of simplesam-like records
    consensus            :369-381   (host helper on one weight dict; unchanged semantics)
    consensus_sequence   :384-430   device per-site pass + host splice/trim/upper
    build_report         :437-485
    bam_to_consensus     :488-555
    weights / features   :558-664   integer columns from device tables, float columns numpy/scipy
    cdr_* / merge_*      :156-366   --realign host logic over device tables (O(L) numpy scans)

The record loop and the per-site loop run in kindel_amd/csrc/kd_kernels.h on the GPU through
the C-ABI (kindel_amd/_native.py).  There is no CPU implementation of either loop here.

Attribution.  This module is an interface mirror of bede/kindel (GPLv3, (c) Bede Constantinides).  Most of it is a new
implementation behind the same names (the CDR / --realign scans are vectorised numpy re-derivations, the tables come from
the device).  Three small pieces necessarily follow the reference line by line, because their OUTPUT must be identical to the
byte or to the float and so leaves no room to differ: ``consensus()`` (:369-381, six lines), the REPORT text of
``_report`` / ``build_report`` (:437-485, its wording, field order and separators), and the pandas tail of ``weights()``
(:586-630: column names, their order, the rounding and the Jeffreys-interval call).  The warning text of ``merge_cdrps`` is the
reference's for the same reason.  Those lines are derived from the GPLv3 reference and are marked where they stand.
"""
import logging
import os
import stat
from collections import OrderedDict, namedtuple

import numpy as np

from . import _native as N

Region = namedtuple("Region", ["start", "end", "seq", "direction"])

_CH = "ATGCN"  # key order of the reference's weight dicts (kindel.py:29)
_CHB = np.frombuffer(b"ATGCN", np.uint8)


class Sequence:
    """Stand-in for dnaio.Sequence (kindel.py:433-434): .name, .sequence, .qualities"""

    def __init__(self, name, sequence, qualities=None):
        self.name, self.sequence, self.qualities = name, sequence, qualities

    def __repr__(self):
        s = self.sequence
        return "Sequence(name=%r, sequence=%r)" % (self.name, s if len(s) < 60 else s[:57] + "...")


# --------------------------------------------------------------------------------------
# array-backed views that look like the reference's lists of dicts
# --------------------------------------------------------------------------------------
class SiteDicts:
    """list-of-dicts view of an [L,5] uint32 table: view[i] == {"A":..,"T":..,"G":..,"C":..,"N":..}"""

    def __init__(self, table, owner=None, role=None):
        self.table, self._owner, self._role = table, owner, role

    def __len__(self):
        return self.table.shape[0]

    def _one(self, i):
        r = self.table[i]
        return {"A": int(r[0]), "T": int(r[1]), "G": int(r[2]), "C": int(r[3]), "N": int(r[4])}

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._one(k) for k in range(*i.indices(len(self)))]
        return self._one(i)

    def __iter__(self):
        return (self._one(i) for i in range(len(self)))


class InsertionDicts:
    """list-of-dicts view of the insertion multiset: view[site] == {"ACG": 3, ...} (L+1 slots)"""

    def __init__(self, n_slots, site, count, strings, owner=None):
        self._n, self._owner = n_slots, owner
        self.site, self.count, self.strings = site, count, strings
        self._by_site = None

    def _index(self):
        if self._by_site is None:
            d = {}
            for s, c, t in zip(self.site.tolist(), self.count.tolist(), self.strings):
                d.setdefault(s, {})[t] = c
            self._by_site = d
        return self._by_site

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError("list index out of range")
        return dict(self._index().get(i, {}))

    def __iter__(self):
        return (self[i] for i in range(self._n))

    def totals(self):
        out = np.zeros(self._n, np.int64)
        np.add.at(out, self.site.astype(np.int64), self.count.astype(np.int64))
        return out


alignment = namedtuple(
    "alignment",
    ["ref_id", "weights", "insertions", "deletions", "clip_starts", "clip_ends", "clip_start_weights",
     "clip_end_weights", "clip_start_depth", "clip_end_depth", "clip_depth", "consensus_depth"],
)


class Pileup:
    """Device-resident result of the record loop for one input (all contigs)."""

    def __init__(self, engine, names, lens, order, bam_path=None):
        self.engine, self.names, self.lens, self.order, self.bam_path = engine, names, lens, order, bam_path
        self._tables = {}

    def close(self):
        """Release the context (device tables, stream, pinned buffers) NOW.  The host copies already fetched stay usable; whatever
        needs the device afterwards raises.  A Pileup without records has no context (engine None)."""
        if self.engine is not None:
            self.engine.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def tables(self, cid):
        """uint32 [KD_NCH, L+1] of contig cid (cached host copy)."""
        if cid not in self._tables:
            self._tables[cid] = self.engine.tables(cid)
        return self._tables[cid]

    def alignment(self, cid):
        # (not cached here: the views refer back to this Pileup -- consensus_sequence() needs the device tables behind them --
        # and a cache would close the cycle Pileup -> alignment -> Pileup, leaving the context to the cycle collector instead of
        # to the reference count of the last alignment alive)
        t = self.tables(cid)
        L = int(self.lens[cid])
        W = np.ascontiguousarray(t[0:5, :L].T)
        S = np.ascontiguousarray(t[N.KD_CH_CSW:N.KD_CH_CSW + 5, :L].T)
        E = np.ascontiguousarray(t[N.KD_CH_CEW:N.KD_CH_CEW + 5, :L].T)
        site, count, strings = self.engine.insertions(cid)
        owner = (self, cid)
        csd = (S[:, 0] + S[:, 1] + S[:, 2] + S[:, 3]).astype(np.int64)  # A,T,G,C without N (kindel.py:90-95)
        ced = (E[:, 0] + E[:, 1] + E[:, 2] + E[:, 3]).astype(np.int64)
        aln = alignment(
            self.names[cid],
            SiteDicts(W, owner, "weights"),
            InsertionDicts(L + 1, site, count, strings, owner),
            t[N.KD_CH_DEL].astype(np.int64).tolist(),
            t[N.KD_CH_CLIP_STARTS].astype(np.int64).tolist(),
            t[N.KD_CH_CLIP_ENDS].astype(np.int64).tolist(),
            SiteDicts(S, owner, "csw"),
            SiteDicts(E, owner, "cew"),
            csd.tolist(),
            ced.tolist(),
            (csd + ced).tolist(),                       # kindel.py:96
            W.max(axis=1).astype(np.int64) if L else np.zeros(0, np.int64),  # aligned - discordant, :83-89
        )
        return aln


def _first_appearance(contig):
    if len(contig) == 0:
        return []
    _, first = np.unique(contig, return_index=True)
    return [int(contig[i]) for i in np.sort(first)]


def pileup_batch(batch, bam_path=None, device=0, lib=None, mode=N.KD_MODE_AUTO, all_contigs=False):
    """Run the device record loop over one decoded batch -> Pileup (raises what the reference raises).

    Only the contigs that HAVE records get device tables -- like the reference, which calls parse_records per RNAME
    seen (kindel.py:143-151) and never allocates for header-only @SQ lines: a human-genome header with reads on chrM
    costs chrM-sized tables, not 3 Gbp of them.  Contig ids are remapped to the dense ids of that subset (header
    order); `names` / `lens` / `order` of the Pileup speak the dense ids.  all_contigs: every contig of `contig_lens` gets
    tables whether or not a record names it -- parse_records(ref_id, ref_len, records) allocates ref_len sites before it looks
    at a record (kindel.py:29-39) and returns an all-zero alignment for an empty or all-unmapped record list."""
    names_all = [str(x) for x in batch["contig_names"]]
    lens_all = np.asarray(batch["contig_lens"], np.uint32)
    contig = np.asarray(batch["contig"])
    if len(contig) == 0 and not all_contigs:
        return Pileup(None, [], np.zeros(0, np.uint32), [], bam_path)   # no records: parse_bam returns {} (kindel.py:143-152)
    used = np.unique(contig)
    if all_contigs or len(used) == len(lens_all):
        sub, names, lens = batch, names_all, lens_all
    else:
        remap = np.zeros(len(lens_all), np.uint32)
        remap[used] = np.arange(len(used), dtype=np.uint32)
        sub = dict(batch)
        sub["contig"] = remap[contig]
        names = [names_all[int(i)] for i in used]
        lens = lens_all[used]
    eng = N.Engine(lens, device=device, lib=lib, mode=mode)
    eng.push(sub)
    eng.finalize()
    return Pileup(eng, names, lens, _first_appearance(np.asarray(sub["contig"])), bam_path)


#: a header whose contigs add up to at most this many sites is laid out whole and the file is streamed at once (decode of batch
#: k+1 overlapped with copy + kernels of batch k); a larger header (a human genome's) is first SCANNED for the contigs that have
#: records (one streamed decode pass, nothing kept), then streamed over those alone -- the reference lays out per RNAME seen
#: (kindel.py:143-151) -- so that neither the header nor the file has to fit anywhere
STREAM_MAX_SITES = 1 << 26      # 67 M sites = 5 GB of tables at 76 B/site


def _is_regular_file(path):
    try:
        return stat.S_ISREG(os.stat(path).st_mode)
    except OSError:
        return True      # (let the decoder report what is wrong with the path)


def _contigs_in_use(bam_path, threads, chunk_bytes, lib, with_order=False):
    """One streamed pass over the file: the @SQ entries that have records, in file order of the header.
    with_order: -> (those, the same entries in order of first appearance in the file: kindel.py:143-151)."""
    st = N.Stream(bam_path, threads=threads, chunk_bytes=chunk_bytes, lib=lib)
    try:
        seen = np.zeros(len(st.contig_lens), bool)
        order = []
        while True:
            b = st.next_batch()
            if b is None:
                break
            c = np.asarray(b["contig"])
            if with_order and len(c):
                u, first = np.unique(c, return_index=True)
                order.extend(int(x) for x in u[np.argsort(first, kind="stable")] if not seen[x])
            seen[np.unique(c)] = True
        return (np.flatnonzero(seen), order) if with_order else np.flatnonzero(seen)
    finally:
        st.close()


def pileup_file(bam_path, device=0, lib=None, threads=0, chunk_bytes=0, stream=None, ingest=None, contigs=None):
    """Decode + device record loop of one SAM / BAM file -> Pileup.  stream: False = the whole file decoded as ONE batch (only the
    contigs with records laid out); True = streamed over the whole header, whatever its size; None (default) = streamed, a header
    beyond STREAM_MAX_SITES scanned first for the contigs in use.  ingest: "host" (default: the native host decoder feeds the GPU) or "gpu"
    (opt-in, also KINDEL_INGEST=gpu: the BGZF blocks are inflated and the BAM records walked ON the GPU, kd_push_bam_gpu; a file
    that path cannot read -- SAM text, plain gzip, CG-tag CIGARs, a header larger than STREAM_MAX_SITES -- takes the host path).
    contigs: header indices -- only these @SQ entries are laid out and only their records counted (one of several passes over a
    file whose reference does not fit the GPU at once: bam_to_consensus); streamed through the host decoder."""
    if contigs is not None:
        stream, ingest = True, "host"
    if stream is None and not _is_regular_file(bam_path):
        # the default route may open the file a second time (a large header: _contigs_in_use; the opt-in device-side ingest falling
        # back to the host decoder); an input that can be read only once (a pipe, /dev/stdin) is decoded in ONE pass as a whole,
        # like the reference reads it (kindel.py:136-145)
        stream = False
    if (ingest or os.environ.get("KINDEL_INGEST", "host")) == "gpu" and stream is not False:
        try:
            with N.BgzfPlan(bam_path, lib=lib) as plan:
                total = int(plan.contig_lens.astype(np.uint64).sum())
                if len(plan.contig_lens) and total <= STREAM_MAX_SITES:
                    eng = N.Engine(plan.contig_lens, device=device, lib=lib)
                    try:
                        info = eng.push_bam_gpu(plan)
                        eng.finalize()            # (the reference's deferred exceptions surface here: the tables go with them)
                        first = eng.contig_first()
                    except BaseException:
                        eng.close()
                        raise
                    used = np.flatnonzero(first != np.uint64(0xFFFFFFFFFFFFFFFF))
                    order = [int(c) for c in used[np.argsort(first[used], kind="stable")]]   # first appearance, kindel.py:143-151
                    pl = Pileup(eng, list(plan.contig_names), plan.contig_lens, order, bam_path)
                    pl.ingest = dict(info, path="gpu")
                    return pl
        except N.UnsupportedByGpuIngest:
            pass
    if stream is not False:
        st = N.Stream(bam_path, threads=threads, chunk_bytes=chunk_bytes, lib=lib)
        try:
            total = int(st.contig_lens.astype(np.uint64).sum())
            if len(st.contig_lens) == 0:
                if st.next_batch() is not None:
                    raise KeyError("no @SQ lines in header")
                return Pileup(None, [], np.zeros(0, np.uint32), [], bam_path)
            names, lens = list(st.contig_names), st.contig_lens
            if contigs is not None or (total > STREAM_MAX_SITES and not stream):
                # a header far larger than what the records touch: lay out the contigs in use only (a second pass over the
                # file: regular files only, see above) -- or the caller's choice of contigs
                keep = np.asarray(sorted(int(c) for c in contigs), np.int64) if contigs is not None else _contigs_in_use(bam_path, threads, chunk_bytes, lib)
                if len(keep) == 0:
                    return Pileup(None, [], np.zeros(0, np.uint32), [], bam_path)
                cmap = np.full(len(lens), 0xFFFFFFFE if contigs is not None else 0xFFFFFFFF, np.uint32)      # (records elsewhere: dropped / an error)
                cmap[keep] = np.arange(len(keep), dtype=np.uint32)
                st.set_contig_map(cmap)
                names, lens = [names[int(i)] for i in keep], lens[keep]
            eng = N.Engine(lens, device=device, lib=lib)
            try:
                info = eng.push_stream(st)
                eng.finalize()                    # (the reference's deferred exceptions surface here: the tables go with them)
                first = eng.contig_first()
            except BaseException:
                eng.close()
                raise
            used = np.flatnonzero(first != np.uint64(0xFFFFFFFFFFFFFFFF))
            order = [int(c) for c in used[np.argsort(first[used], kind="stable")]]   # first appearance, kindel.py:143-151
            pl = Pileup(eng, names, lens, order, bam_path)
            pl.ingest = info
            return pl
        finally:
            st.close()
    return pileup_batch(N.decode_file(bam_path, threads=threads, lib=lib), bam_path=bam_path, device=device, lib=lib)


def parse_bam(bam_path):
    """Returns alignment information for each reference sequence as an OrderedDict (kindel.py:131-153)"""
    pl = pileup_file(bam_path)
    return OrderedDict((pl.names[cid], pl.alignment(cid)) for cid in pl.order)


_NIB = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
_OPS = {c: i for i, c in enumerate("MIDNSHP=X")}


def parse_records(ref_id, ref_len, records):
    """parse_records(ref_id, ref_len, records) of kindel.py:21-128 for an iterable of
    simplesam-like records (.pos, .mapped, .seq, .cigars); packs them and runs the device loop."""
    contig, pos0, flag, seq_off, seq_len, cig_off, n_cig, seq4, cigar = [], [], [], [], [], [], [], bytearray(), []
    for r in records:
        s = "" if r.seq == "*" else r.seq
        ops = []
        try:
            for ln, op in r.cigars:
                if op is not None:
                    ops.append((int(ln) << 4) | _OPS.get(op, 15))
        except RuntimeError:
            ops = []  # CIGAR '*': the engine raises RuntimeError for a mapped read with real SEQ (kindel.py:47)
        contig.append(0); pos0.append(r.pos - 1); flag.append(0 if r.mapped else 4)
        seq_off.append(len(seq4)); seq_len.append(len(s)); cig_off.append(len(cigar)); n_cig.append(len(ops))
        nib = [_NIB.get(c.upper(), 0) for c in s] + [0]
        seq4.extend((nib[i] << 4) | nib[i + 1] for i in range(0, len(s), 2))
        cigar.extend(ops)
    batch = dict(contig=np.asarray(contig, np.uint32), pos0=np.asarray(pos0, np.int32), flag=np.asarray(flag, np.uint32),
                 seq_off=np.asarray(seq_off, np.uint64), seq_len=np.asarray(seq_len, np.uint32),
                 cig_off=np.asarray(cig_off, np.uint64), n_cig=np.asarray(n_cig, np.uint32),
                 seq4=np.frombuffer(bytes(seq4) + b"\0", np.uint8), cigar=np.asarray(cigar + [0], np.uint32),
                 contig_names=np.asarray([ref_id]), contig_lens=np.asarray([ref_len], np.uint32))
    return pileup_batch(batch, all_contigs=True).alignment(0)


# --------------------------------------------------------------------------------------
# realign: clip-dominant regions (kindel.py:156-366), host scans over device tables
# --------------------------------------------------------------------------------------
def _cns_chars(tab5):
    """consensus(w)[0] per site for an [n,5] table: first max in A,T,G,C,N order, 'N' if empty (kindel.py:373-375)"""
    idx = np.argmax(tab5, axis=1)
    ch = _CHB[idx].copy()
    ch[tab5.sum(axis=1) == 0] = ord("N")
    return ch


def _tab(x):
    return x.table if isinstance(x, SiteDicts) else np.asarray([[d[c] for c in _CH] for d in x], np.int64).reshape(-1, 5)


def _masked(L, mask_ends):
    m = np.zeros(L, bool)
    m[:mask_ends] = True          # positions[:mask_ends]
    m[L - mask_ends if mask_ends else 0:] = True  # positions[-mask_ends:] (mask_ends == 0 masks everything, :168)
    return m


def _masked_at(pos, L, mask_ends):
    """_masked(L, mask_ends)[pos] for an array of positions"""
    pos = np.asarray(pos, np.int64)
    return (pos < mask_ends) | (pos >= (L - mask_ends if mask_ends else 0))


def cdr_scan_inputs(W, d, X, clip_decay_threshold, mask_ends, L, offset=0, depth=None):
    """What one direction's clip-dominant-region scan reads of a contig, SPARSE: for the rows [offset, offset + n) of the contig's
    tables (W: [n,5] weights, d: [n] deletions, X: [n,5] clip start / end weights; L = the whole contig's length)
      cand      the unmasked sites whose clip depth dominates  (csd / (sum(w) + d + 1) > 0.5, kindel.py:183 / :244)   + their
      cand_ch   consensus(X[site])[0]
      ext       the sites an extension runs over               (csd > (sum(w) + d) * clip_decay_threshold, :202 / :256)  + their
      ext_ch    consensus(X[site])[0]
    as int64 positions in the contig (ascending) and uint8 characters.  Every predicate is a function of ONE site's counters, so
    the rows of a shard give the shard's part and the parts of all shards, concatenated, are the contig's (shard.realign_patches:
    a few bytes per clip-dominant site cross the links instead of the tables)."""
    W, X = np.asarray(W, np.int64), np.asarray(X, np.int64)
    d = np.asarray(d, np.int64)
    if depth is None:
        depth = X[:, 0] + X[:, 1] + X[:, 2] + X[:, 3]                   # the alignment's clip_start_depth / clip_end_depth: A,T,G,C without N (kindel.py:90-95)
    depth = np.asarray(depth, np.int64)
    wsum = W.sum(axis=1)
    pos = np.arange(offset, offset + W.shape[0], dtype=np.int64)
    cand = (2 * depth > wsum + d + 1) & ~_masked_at(pos, L, mask_ends)
    ext = depth > (wsum + d).astype(np.float64) * clip_decay_threshold
    chars = _cns_chars(X)
    return dict(cand=pos[cand], cand_ch=chars[cand].astype(np.uint8), ext=pos[ext], ext_ch=chars[ext].astype(np.uint8))


def cdr_scan_concat(parts):
    """The scan inputs of a contig from its shards' parts (disjoint row ranges, any order)."""
    out = {}
    for k in ("cand", "ext"):
        p = np.concatenate([np.asarray(x[k], np.int64) for x in parts]) if parts else np.zeros(0, np.int64)
        c = np.concatenate([np.asarray(x[k + "_ch"], np.uint8) for x in parts]) if parts else np.zeros(0, np.uint8)
        o = np.argsort(p, kind="stable")
        out[k], out[k + "_ch"] = p[o], c[o]
    return out


def _ext_runs(ext):
    """-> (first, last): for every entry of the ascending position list `ext`, the index of the first / last entry of its run of
    consecutive positions"""
    n = len(ext)
    if n == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    brk = np.flatnonzero(np.diff(ext) != 1)                   # a run ends at brk[k], the next starts at brk[k] + 1
    starts = np.concatenate([[0], brk + 1])
    ends = np.concatenate([brk, [n - 1]])
    run = np.searchsorted(starts, np.arange(n), side="right") - 1
    return starts[run], ends[run]


def cdr_start_regions(L, inp):
    """cdr_start_consensuses (kindel.py:156-213) over sparse scan inputs: a region opens at the first unmasked clip-dominant site
    no earlier region covers and runs to the first site that fails the extension test (that site is its `end`: not covered, its
    base not included) -- or to the contig's last site, whose base IS included while `end` still names it (:196-208)."""
    ext, ext_ch = inp["ext"], inp["ext_ch"]
    _, last = _ext_runs(ext)
    regions = []
    for pos in inp["cand"].tolist():
        if any(r.start <= pos < r.end for r in regions):
            continue
        i = int(np.searchsorted(ext, pos))
        if i == len(ext) or int(ext[i]) != pos:              # the extension stops at once
            regions.append(Region(pos, pos, "", "\u2192"))
            continue
        j = int(last[i])
        end_site = int(ext[j]) + 1                            # the first site behind the run
        seq = ext_ch[i:j + 1].tobytes().decode()
        regions.append(Region(pos, end_site if end_site <= L - 1 else L - 1, seq, "\u2192"))
    for region in regions:
        logging.debug(region)
    return regions


def cdr_end_regions(L, inp):
    """cdr_end_consensuses (kindel.py:216-275) over sparse scan inputs: from the right; a region ends behind an unmasked
    clip-dominant site, its extension walks down from the site in front of it, and the region's `start` is the site the walk
    stops at (covered, its base not included) -- or site 0 with its base included when the walk never stops (:251-270)."""
    ext, ext_ch = inp["ext"], inp["ext_ch"]
    first, _ = _ext_runs(ext)
    regions = []
    for pos, ch in zip(inp["cand"][::-1].tolist(), inp["cand_ch"][::-1].tolist()):
        if any(r.start <= pos < r.end for r in regions):
            continue
        end_pos = pos + 1
        if pos == 0:  # reversed_weights[L:] is empty: for-else with nothing accumulated (:251,268-270)
            regions.append(Region(pos, end_pos, "", "\u2190"))
            continue
        i = int(np.searchsorted(ext, pos - 1))
        if i == len(ext) or int(ext[i]) != pos - 1:          # the walk stops at once, at pos - 1: nothing accumulated
            regions.append(Region(pos - 1, end_pos, "", "\u2190"))
            continue
        f = int(first[i])
        b = int(ext[f]) - 1                                   # the site the walk stops at (-1: it never stops)
        seq = ext_ch[f:i + 1].tobytes().decode() + chr(ch)    # bases b + 1 .. pos - 1 (extension sites) and pos
        regions.append(Region(b if b >= 0 else 0, end_pos, seq, "\u2190"))
    for region in regions:
        logging.debug(region)
    return regions


def cdr_start_consensuses(weights, deletions, clip_start_weights, clip_start_depth, clip_decay_threshold, mask_ends):
    """Returns list of Region instances for right clipped consensuses of clip-dominant region (kindel.py:156-213)"""
    W, S = _tab(weights), _tab(clip_start_weights)
    L = W.shape[0]
    return cdr_start_regions(L, cdr_scan_inputs(W, np.asarray(deletions, np.int64)[:L], S, clip_decay_threshold, mask_ends, L,
                                                depth=np.asarray(clip_start_depth, np.int64)[:L]))


def cdr_end_consensuses(weights, deletions, clip_end_weights, clip_end_depth, clip_decay_threshold, mask_ends):
    """Returns list of Region instances for left clipped consensuses of clip-dominant region (kindel.py:216-275)"""
    W, E = _tab(weights), _tab(clip_end_weights)
    L = W.shape[0]
    return cdr_end_regions(L, cdr_scan_inputs(W, np.asarray(deletions, np.int64)[:L], E, clip_decay_threshold, mask_ends, L,
                                              depth=np.asarray(clip_end_depth, np.int64)[:L]))


def pair_cdrs(fwd_cdrs, rev_cdrs):
    """kindel.py:305-316: every right-clipped region with the first left-clipped region it shares a site with"""
    paired_cdrs = []
    for f in fwd_cdrs:
        for r in rev_cdrs:
            if max(f.start, r.start) < min(f.end, r.end):  # set(range) & set(range) non-empty, :314
                paired_cdrs.append((f, r))
                break
    return paired_cdrs


def cdrp_consensuses(weights, deletions, clip_start_weights, clip_end_weights, clip_start_depth, clip_end_depth,
                     clip_decay_threshold, mask_ends):
    """Returns list of 2-tuples of L&R clipped consensus sequences around clip-dominant regions (kindel.py:278-320)"""
    fwd_cdrs = cdr_start_consensuses(weights, deletions, clip_start_weights, clip_start_depth,
                                     clip_decay_threshold, mask_ends)
    rev_cdrs = cdr_end_consensuses(weights, deletions, clip_end_weights, clip_end_depth,
                                   clip_decay_threshold, mask_ends)
    return pair_cdrs(fwd_cdrs, rev_cdrs)


def merge_by_lcs(s1, s2, min_overlap):
    """Returns superstring of s1 and s2 about an exact overlap of len > min_overlap (kindel.py:323-347)"""
    a = np.frombuffer(s1.encode(), np.uint8)
    b = np.frombuffer(s2.encode(), np.uint8)
    longest, x_longest = 0, 0
    prev = np.zeros(len(b) + 1, np.int64)
    for x in range(1, len(a) + 1):  # row-wise form of the reference's DP; first strict maximum wins
        row = np.zeros(len(b) + 1, np.int64)
        eq = b == a[x - 1]
        row[1:][eq] = prev[:-1][eq] + 1
        m = int(row.max()) if len(b) else 0
        if m > longest:
            longest, x_longest = m, x
        prev = row
    lcs = s1[x_longest - longest:x_longest]
    logging.debug(f"merge_by_lcs(): s1: {s1}, s2: {s2}, min_overlap: {min_overlap}")
    if len(lcs) < min_overlap:
        return None
    return s1.split(lcs, 1)[0] + lcs + s2.split(lcs, 1)[1]


def merge_cdrps(cdrps, min_overlap):
    """Returns merged clip-dominant region pairs as Region instances (kindel.py:350-366)"""
    merged = []
    for fwd_cdr, rev_cdr in cdrps:
        merged_seq = merge_by_lcs(fwd_cdr.seq, rev_cdr.seq, min_overlap)
        if not merged_seq:
            logging.warning(
                f"No overlap found for clip dominant region spanning positions {fwd_cdr.start}-{rev_cdr.end} (min_overlap = {min_overlap})")
        merged.append(Region(fwd_cdr.start, rev_cdr.end, merged_seq, None))
    return merged


# --------------------------------------------------------------------------------------
# consensus
# --------------------------------------------------------------------------------------
def consensus(weight):   # (follows kindel.py:369-381 line by line: see the module header)
    """Returns tuple of consensus base, weight and flag indicating a tie for consensus (kindel.py:369-381)"""
    base, frequency = max(weight.items(), key=lambda x: x[1]) if sum(weight.values()) else ("N", 0)
    weight_sans_consensus = {k: d for k, d in weight.items() if k != base}
    tie = True if frequency and frequency in weight_sans_consensus.values() else False
    aligned_depth = sum(weight.values())
    proportion = round(frequency / aligned_depth, 2) if aligned_depth else 0
    return (base, frequency, proportion, tie)


def _patch_plan(L, cdr_patches):
    """The skip/patch control flow of consensus_sequence (kindel.py:393-401) as data:
    -> [(start, end, text)]: sites [start, end) emit nothing, `text` is spliced at start."""
    plan = []
    if not cdr_patches:
        return plan
    starts = sorted({r.start for r in cdr_patches if r.seq and 0 <= r.start < L})
    pos = 0
    for s in starts:
        if s < pos:
            continue  # inside a span being skipped: never examined
        patch = next(r for r in cdr_patches if r.start == s)
        text = patch.seq.lower()  # AttributeError if the first patch at s has seq None, as in the reference
        span = patch.end - patch.start
        if span - 1 < 0:  # skip_positions goes negative and stays truthy: nothing else is ever emitted
            plan.append((s, L, text))
            break
        plan.append((s, min(L, s + span), text))
        pos = s + span
    return plan


def _device_consensus(pl, cid, cdr_patches, trim_ends, min_depth, uppercase):
    L = int(pl.lens[cid])
    base = pl.engine.contig_base(cid)
    plan = _patch_plan(L, cdr_patches)
    pl.engine.consensus_run(min_depth, [(base + s, base + e) for s, e, _ in plan])
    raw, ch, mm, poff = pl.engine.consensus_fetch(cid)
    if plan:
        parts, prev = [], 0
        for (s, e, text), o in zip(plan, poff.tolist()):
            parts.append(raw[prev:o]); parts.append(text.encode()); prev = o
        parts.append(raw[prev:])
        raw = b"".join(parts)
    seq = raw.decode("ascii")
    if trim_ends:
        seq = seq.strip("N")        # kindel.py:425-426
    if uppercase:
        seq = seq.upper()           # kindel.py:427-428
    return seq, ch, mm


def _device_consensus_all(pl, patches_by_cid, trim_ends, min_depth, uppercase):
    """All contigs of the input with ONE consensus run and ONE device-to-host copy (the reference loops
    consensus_sequence over the contigs, kindel.py:515-551) -> {cid: (seq, changes, depth_minmax)}."""
    eng = pl.engine
    if not pl.order:
        return {}
    plans = {cid: _patch_plan(int(pl.lens[cid]), patches_by_cid.get(cid)) for cid in pl.order}
    flat, owner = [], []
    for cid in pl.order:
        base = eng.contig_base(cid)
        for s, e, _ in plans[cid]:
            flat.append((base + s, base + e)); owner.append(cid)
    eng.consensus_run(min_depth, flat)
    off, _ = eng.consensus_offsets()
    buf = np.empty(int(off[-1]) + 16, np.uint8)
    chg = np.empty(max(eng.total_sites(), 1), np.uint8)
    off = eng.consensus_fetch_all_into(buf, chg)
    out = {}
    for cid in pl.order:
        L = int(pl.lens[cid])
        base = eng.contig_base(cid)
        raw = buf[int(off[cid]): int(off[cid + 1])].tobytes()
        _, mm, poff = eng.consensus_meta(cid)
        plan = plans[cid]
        if plan:
            mine = [o for o, c in zip(poff.tolist(), owner) if c == cid]
            parts, prev = [], 0
            for (s, e, text), o in zip(plan, mine):
                parts.append(raw[prev:o]); parts.append(text.encode()); prev = o
            parts.append(raw[prev:])
            raw = b"".join(parts)
        seq = raw.decode("ascii")
        if trim_ends:
            seq = seq.strip("N")        # kindel.py:425-426
        if uppercase:
            seq = seq.upper()           # kindel.py:427-428
        out[cid] = (seq, chg[base: base + L].copy(), mm)
    return out


_CHG = {0: None, ord("D"): "D", ord("N"): "N", ord("I"): "I"}


def _changes_list(ch):
    out = [None] * len(ch)
    for i in np.flatnonzero(ch).tolist():
        out[i] = _CHG[int(ch[i])]
    return out


def consensus_sequence(weights, insertions, deletions, cdr_patches, trim_ends, min_depth, uppercase):
    """consensus_sequence of kindel.py:384-430.  `weights` must come from this package's
    parse_bam()/parse_records() (it carries a handle to the device tables)."""
    owner = getattr(weights, "_owner", None)
    if owner is None:
        raise TypeError("kindel_amd.consensus_sequence needs the weights object returned by kindel_amd.parse_bam(); "
                        "arbitrary list-of-dict tables have no device pileup behind them")
    pl, cid = owner
    seq, ch, _ = _device_consensus(pl, cid, cdr_patches, trim_ends, min_depth, uppercase)
    return seq, _changes_list(ch)


def consensus_seqrecord(consensus, ref_id):
    return Sequence(name=f"{ref_id}_cns", sequence=consensus, qualities=None)


def _report(ref_id, depth_minmax, ch, cdr_patches, bam_path, realign, min_depth, min_overlap, clip_decay_threshold,
            trim_ends, uppercase):
    def sites(code):
        return ", ".join(map(str, (np.flatnonzero(ch == ord(code)) + 1).tolist()))

    cdr_patches_fmt = ["{}-{}: {}".format(r.start, r.end, r.seq) for r in cdr_patches] if cdr_patches else ""
    report = "========================= REPORT ===========================\n"
    report += "reference: {}\n".format(ref_id)
    report += "options:\n"
    report += "- bam_path: {}\n".format(bam_path)
    report += "- min_depth: {}\n".format(min_depth)
    report += "- realign: {}\n".format(realign)
    report += "    - min_overlap: {}\n".format(min_overlap)
    report += "    - clip_decay_threshold: {}\n".format(clip_decay_threshold)
    report += "- trim_ends: {}\n".format(trim_ends)
    report += "- uppercase: {}\n".format(uppercase)
    report += "observations:\n"
    report += "- min, max observed depth: {}, {}\n".format(depth_minmax[0], depth_minmax[1])
    report += "- ambiguous sites: {}\n".format(sites("N"))
    report += "- insertion sites: {}\n".format(sites("I"))
    report += "- deletion sites: {}\n".format(sites("D"))
    report += "- clip-dominant regions: {}\n".format(", ".join(cdr_patches_fmt))
    return report


def build_report(ref_id, weights, changes, cdr_patches, bam_path, realign, min_depth, min_overlap,
                 clip_decay_threshold, trim_ends, uppercase):
    """build_report of kindel.py:437-485 (host text formatting)."""
    W = _tab(weights)
    ad = W[:, 0] + W[:, 1] + W[:, 2] + W[:, 3]  # A,T,G,C: no N (:450)
    ch = np.asarray([0 if c is None else ord(c) for c in changes], np.uint8)
    return _report(ref_id, (int(ad.min()), int(ad.max())), ch, cdr_patches, bam_path, realign, min_depth, min_overlap,
                   clip_decay_threshold, trim_ends, uppercase)


def _consensus_of_pileup(pl, bam_path, realign, min_depth, min_overlap, clip_decay_threshold, mask_ends, trim_ends, uppercase):
    """The per-contig loop of kindel.py:501-551 over one Pileup (closed on the way out) -> [(ref_id, SeqRecord, changes, report)] in
    pl.order."""
    try:
        patches = {}
        for cid in pl.order:
            if realign:
                aln = pl.alignment(cid)
                cdrps = cdrp_consensuses(aln.weights, aln.deletions, aln.clip_start_weights, aln.clip_end_weights,
                                         aln.clip_start_depth, aln.clip_end_depth, clip_decay_threshold, mask_ends)
                patches[cid] = merge_cdrps(cdrps, min_overlap)
            else:
                patches[cid] = None
        done = _device_consensus_all(pl, patches, trim_ends, min_depth, uppercase)
    finally:
        pl.close()      # everything below is host data: the context -- tables, stream, pinned buffers -- goes now (no context: no records)
    out = []
    for cid in pl.order:
        ref_id = pl.names[cid]
        seq, ch, mm = done[cid]
        report = _report(ref_id, mm, ch, patches[cid], bam_path, realign, min_depth, min_overlap, clip_decay_threshold,
                         trim_ends, uppercase)
        out.append((ref_id, consensus_seqrecord(seq, ref_id), _changes_list(ch), report))
    return out


def _contig_groups(lens, used, n_groups):
    """`used` (header indices, ascending) cut into at most n_groups runs of about equal total sites; a run is never empty."""
    if n_groups >= len(used):
        return [[int(c)] for c in used]
    sites = np.asarray([int(lens[int(c)]) + 1 for c in used], np.float64)
    cum = np.cumsum(sites)
    cuts = np.searchsorted(cum, cum[-1] * np.arange(1, n_groups) / n_groups, side="left") + 1
    edges = sorted(set([0] + [int(min(x, len(used))) for x in cuts] + [len(used)]))
    return [[int(c) for c in used[a:b]] for a, b in zip(edges[:-1], edges[1:]) if b > a]


def bam_to_consensus(bam_path, realign=False, min_depth=1, min_overlap=9, clip_decay_threshold=0.1, mask_ends=50,
                     trim_ends=False, uppercase=False):
    """bam_to_consensus of kindel.py:488-555: same arguments, same result tuple.

    The reference allocates per RNAME as it meets it (kindel.py:143-151) and holds one contig's tables at a time in effect; here the
    contigs in use are laid out together on the device (76 B of tables per site and 20 more of working arrays: 288 GB hold ~2.5 G
    sites).  A reference that does not fit -- MemoryError from the device allocation -- is processed in GROUPS of contigs, one
    streamed pass over the file per group (halved until a group fits); the result is the same tuple, contigs in order of first
    appearance.  Regular files only: a pipe cannot be read twice."""
    args = (realign, min_depth, min_overlap, clip_decay_threshold, mask_ends, trim_ends, uppercase)
    try:
        rows = _consensus_of_pileup(pileup_file(bam_path), bam_path, *args)
    except MemoryError:
        if not _is_regular_file(bam_path):
            raise
        rows = _consensus_in_groups(bam_path, args)
    consensuses, refs_changes, refs_reports = [], {}, {}
    for ref_id, rec, changes, report in rows:
        consensuses.append(rec)
        refs_reports[ref_id] = report
        refs_changes[ref_id] = changes
    result = namedtuple("result", ["consensuses", "refs_changes", "refs_reports"])
    return result(consensuses, refs_changes, refs_reports)


def _consensus_in_groups(bam_path, args):
    """bam_to_consensus for a reference whose tables do not fit the device at once: the contigs in use (one scanning pass: which, and
    in which order they first appear) in groups of consecutive header entries, every group a streamed pass of its own over the file
    with only its contigs laid out.  Twice as many groups after every MemoryError, down to one contig per group.  (A file with failing
    reads on several contigs raises the exception of the first GROUP that has one -- header order -- where the one pass raises the first
    contig's in order of appearance: the same exception types, possibly another read's.)"""
    st = N.Stream(bam_path)
    try:
        names, lens = list(st.contig_names), st.contig_lens
    finally:
        st.close()
    used, order = _contigs_in_use(bam_path, 0, 0, None, with_order=True)
    n_groups = 2
    while True:
        groups = _contig_groups(lens, used, n_groups)
        try:
            by_name = {}
            for grp in groups:
                for row in _consensus_of_pileup(pileup_file(bam_path, contigs=grp), bam_path, *args):
                    by_name[row[0]] = row
            return [by_name[names[c]] for c in order if names[c] in by_name]
        except MemoryError:
            if n_groups >= len(used):      # one contig per group and it still does not fit
                raise
            n_groups = min(2 * n_groups, len(used))


def bam_to_consensus_sharded(bam_path, rank, world, device="cpu", dev_index=0, group=None, realign=False, min_depth=1, min_overlap=9,
                             clip_decay_threshold=0.1, mask_ends=50, trim_ends=False, uppercase=False, threads=0, lib=None):
    """bam_to_consensus (kindel.py:488-555) with the per-contig loop (:143-151, :501-551) spread over `world` ranks, one GPU
    each (kindel_amd/shard.py: sharded ingest, shard-local pileup, one all-gather).  Call on every rank of an initialised
    torch.distributed group; every rank returns the same result tuple.  realign=True: every rank evaluates the clip-dominant-region
    predicates on its own sites, the sparse results are gathered (shard.realign_patches: no table crosses the links), every rank runs
    the same scans over them and patches its part."""
    from . import shard
    out = shard.pileup_consensus_sharded(bam_path, rank, world, device=device, dev_index=dev_index, group=group, min_depth=min_depth,
                                         threads=threads, lib=lib,
                                         realign=dict(min_overlap=min_overlap, clip_decay_threshold=clip_decay_threshold, mask_ends=mask_ends) if realign else None)
    consensuses, refs_changes, refs_reports = [], {}, {}
    for cid in out["order"]:
        ref_id = out["names"][cid]
        seq = out["seqs"][cid].decode("ascii")
        if trim_ends:
            seq = seq.strip("N")        # kindel.py:425-426
        if uppercase:
            seq = seq.upper()           # kindel.py:427-428
        ch = out["changes"][cid]
        consensuses.append(consensus_seqrecord(seq, ref_id))
        refs_reports[ref_id] = _report(ref_id, out["minmax"][cid], ch, (out.get("patches") or {}).get(cid), bam_path, realign, min_depth, min_overlap,
                                       clip_decay_threshold, trim_ends, uppercase)
        refs_changes[ref_id] = _changes_list(ch)
    result = namedtuple("result", ["consensuses", "refs_changes", "refs_reports"])
    return result(consensuses, refs_changes, refs_reports)


# --------------------------------------------------------------------------------------
# tables as DataFrames (kindel.py:558-664): integer columns from the device, floats on host
# --------------------------------------------------------------------------------------
def weights(bam_path, relative=False, confidence=True, confidence_alpha=0.01):
    """Returns DataFrame of per-site nucleotide frequencies, depth, consensus, clip start/end,
    confidence intervals and entropy (kindel.py:558-630)."""
    import pandas as pd
    import scipy.stats

    pl = pileup_file(bam_path)
    try:
        for cid in pl.order:
            pl.tables(cid)      # (host copies)
    finally:
        pl.close()              # the rest is host arithmetic: the context goes now
    frames = []
    for cid in pl.order:
        t = pl.tables(cid).astype(np.int64)
        L = int(pl.lens[cid])
        ins = t[N.KD_CH_INS_TOTAL]
        frames.append(pd.DataFrame(OrderedDict([
            ("chrom", np.full(L, pl.names[cid], dtype=object)),
            ("pos", np.arange(1, L + 1, dtype=np.int64)),
            ("A", t[N.KD_CH_A, :L]), ("C", t[N.KD_CH_C, :L]), ("G", t[N.KD_CH_G, :L]), ("T", t[N.KD_CH_T, :L]),
            ("N", t[N.KD_CH_N, :L]),
            ("insertions", ins[1:L + 1]),                  # sum(aln.insertions[i]), i = 1..L   (:581)
            ("deletions", t[N.KD_CH_DEL, :L]),             # aln.deletions[i - 1]              (:582)
            ("clip_starts", t[N.KD_CH_CLIP_STARTS, :L]),   # (:583)
            ("clip_ends", t[N.KD_CH_CLIP_ENDS, :L]),       # (:584)
        ])))
    weights_df = pd.concat(frames, ignore_index=True) if frames else pd.DataFrame(
        columns=["chrom", "pos", "A", "C", "G", "T", "N", "insertions", "deletions", "clip_starts", "clip_ends"])
    six = ["A", "C", "G", "T", "N", "deletions"]
    weights_df["depth"] = weights_df[six].sum(axis=1)
    consensus_depths_df = weights_df[six].max(axis=1)
    weights_df["consensus"] = consensus_depths_df.divide(weights_df.depth)
    rel_weights_df = pd.DataFrame()
    for nt in six:
        rel_weights_df[[nt]] = weights_df[[nt]].divide(weights_df.depth, axis=0)
    rel_weights_df = rel_weights_df.round({nt: 4 for nt in six})
    with np.errstate(invalid="ignore", divide="ignore"):
        weights_df["shannon"] = scipy.stats.entropy(rel_weights_df[["A", "C", "G", "T"]].values, axis=1)
    if confidence:
        c = consensus_depths_df.to_numpy()
        n = weights_df["depth"].to_numpy()
        lo, hi = scipy.stats.beta.interval(1 - confidence_alpha, c + 0.5, n - c + 0.5)  # Jeffreys, :569-574
        weights_df["lower_ci"] = lo
        weights_df["upper_ci"] = hi
    if relative:
        for nt in ["A", "C", "G", "T", "N"]:
            weights_df[[nt]] = rel_weights_df[[nt]]
    return weights_df.round(dict(consensus=3, lower_ci=3, upper_ci=3, shannon=3))


def features(bam_path):
    """Returns DataFrame of relative per-site nucleotide frequencies, insertions, deletions and
    entropy (kindel.py:633-664).  Deliberate divergence: the reference indexes the *last* contig's
    indel tables with a global row counter (:644-646) and so crashes on multi-contig input; here
    every contig uses its own tables (identical for single-contig input)."""
    import pandas as pd
    import scipy.stats

    pl = pileup_file(bam_path)
    try:
        for cid in pl.order:
            pl.tables(cid)      # (host copies)
    finally:
        pl.close()
    frames = []
    for cid in pl.order:
        t = pl.tables(cid).astype(np.int64)
        L = int(pl.lens[cid])
        frames.append(pd.DataFrame(OrderedDict([
            ("chrom", np.full(L, pl.names[cid], dtype=object)), ("pos", np.arange(1, L + 1, dtype=np.int64)),
            ("A", t[N.KD_CH_A, :L]), ("C", t[N.KD_CH_C, :L]), ("G", t[N.KD_CH_G, :L]), ("T", t[N.KD_CH_T, :L]),
            ("N", t[N.KD_CH_N, :L]), ("i", t[N.KD_CH_INS_TOTAL, :L]), ("d", t[N.KD_CH_DEL, :L])])))
    df = pd.concat(frames, ignore_index=True)
    df["depth"] = df[["A", "C", "G", "T", "N", "d"]].sum(axis=1)
    consensus_depths = df[["A", "C", "G", "T", "N"]].max(axis=1)
    df["consensus"] = consensus_depths.divide(df.depth)
    for nt in ["A", "C", "G", "T", "N", "i", "d"]:
        df[[nt]] = df[[nt]].divide(df.depth, axis=0)
    with np.errstate(invalid="ignore", divide="ignore"):
        df["shannon"] = scipy.stats.entropy(df[["A", "C", "G", "T", "i", "d"]].values.astype(float), axis=1)
    return df.round(3)


def variants(bam_path, abs_threshold=1, rel_threshold=0.01, only_variants=True, absolute=False):
    """EXTENSION -- parity unpinned.  The reference advertises a `variants` sub-command ("Output variants
    exceeding specified absolute and relative frequency thresholds", README.md:106) but ships no
    implementation (cli.py:64-66 registers consensus/weights/features/plot/version only), so there is nothing
    to be bit-exact against.  This is a thresholded filter over the same device tables `weights()` uses:

    one row per (site, allele) whose count >= abs_threshold and count / depth >= rel_threshold, where
    depth = A+C+G+T+N+deletions (the `depth` column of weights(), kindel.py:586), alleles are A,C,G,T,
    "-" (deletion of the site) and "+<TEXT>" (insertion keyed before the site, upper-case as stored, :55-58);
    `ref` is the site's majority allele over A,T,G,C,N (first max in that order, "N" when empty, :369-381).
    only_variants drops rows whose allele equals `ref`; absolute=True reports counts instead of frequencies.
    Columns: chrom, pos (1-based), ref, alt, type (snv|del|ins), count, depth, frequency."""
    import pandas as pd

    pl = pileup_file(bam_path)
    ins_of = {}
    try:
        for cid in pl.order:
            pl.tables(cid)      # (host copies)
            ins_of[cid] = pl.engine.insertions(cid)
    finally:
        pl.close()
    frames = []
    for cid in pl.order:
        t = pl.tables(cid).astype(np.int64)
        L = int(pl.lens[cid])
        if L == 0:
            continue
        five = t[0:5, :L]                                        # A,T,G,C,N
        depth = five.sum(axis=0) + t[N.KD_CH_DEL, :L]
        ref = np.array(list("ATGCN"))[np.argmax(five, axis=0)]
        ref[five.sum(axis=0) == 0] = "N"
        safe = np.maximum(depth, 1)

        def rows(sites, alt, kind, count):
            if len(sites) == 0:
                return None
            d = depth[sites]
            return pd.DataFrame(OrderedDict([
                ("chrom", np.full(len(sites), pl.names[cid], dtype=object)), ("pos", sites + 1),
                ("ref", ref[sites]), ("alt", alt), ("type", np.full(len(sites), kind, dtype=object)),
                ("count", count), ("depth", d), ("frequency", np.round(count / np.maximum(d, 1), 4))]))

        for ch, nt in ((N.KD_CH_A, "A"), (N.KD_CH_C, "C"), (N.KD_CH_G, "G"), (N.KD_CH_T, "T")):
            c = t[ch, :L]
            keep = (c >= max(abs_threshold, 1)) & (c >= rel_threshold * safe) & (depth > 0)
            if only_variants:
                keep &= ref != nt
            sites = np.flatnonzero(keep)
            frames.append(rows(sites, np.full(len(sites), nt, dtype=object), "snv", c[sites]))
        c = t[N.KD_CH_DEL, :L]
        sites = np.flatnonzero((c >= max(abs_threshold, 1)) & (c >= rel_threshold * safe))
        frames.append(rows(sites, np.full(len(sites), "-", dtype=object), "del", c[sites]))
        site, count, strings = ins_of[cid]
        site, count = np.asarray(site, np.int64), np.asarray(count, np.int64)
        ok = site < L                                            # slot L is never emitted (kindel.py:390)
        ok &= (count >= max(abs_threshold, 1)) & (count >= rel_threshold * safe[np.minimum(site, L - 1)])
        idx = np.flatnonzero(ok)
        frames.append(rows(site[idx], np.array(["+" + strings[i] for i in idx], dtype=object), "ins", count[idx]))
    frames = [f for f in frames if f is not None]
    cols = ["chrom", "pos", "ref", "alt", "type", "count", "depth", "frequency"]
    if not frames:
        return pd.DataFrame(columns=cols)
    df = pd.concat(frames, ignore_index=True)
    order = {name: i for i, name in enumerate(pl.names[c] for c in pl.order)}
    df = df.assign(_c=df.chrom.map(order)).sort_values(["_c", "pos", "type", "alt"], kind="stable").drop(columns="_c")
    if absolute:
        df = df.drop(columns="frequency")
    return df.reset_index(drop=True)


def plotly_clips(bam_path):
    """kindel.py:667-703: HTML depth / soft-clip plot of the first contig (needs plotly)."""
    import plotly.graph_objs as go
    import plotly.offline as py

    with pileup_file(bam_path) as pl:
        aln = pl.alignment(pl.order[0])
        t = pl.tables(pl.order[0]).astype(np.int64)
    aligned_depth = t[0:5, :-1].sum(axis=0).tolist()
    ins = t[N.KD_CH_INS_TOTAL].tolist()
    x_axis = list(range(1, len(aligned_depth) + 1))
    traces = [
        go.Scattergl(x=x_axis, y=aligned_depth, mode="lines", name="Aligned depth"),
        go.Scattergl(x=x_axis, y=aln.clip_depth, mode="lines", name="Soft clip total depth"),
        go.Scattergl(x=x_axis, y=aln.clip_start_depth, mode="lines", name="Soft clip start depth"),
        go.Scattergl(x=x_axis, y=aln.clip_end_depth, mode="lines", name="Soft clip end depth"),
        go.Scattergl(x=x_axis, y=aln.clip_starts, mode="markers", name="Soft clip starts"),
        go.Scattergl(x=x_axis, y=aln.clip_ends, mode="markers", name="Soft clip ends"),
        go.Scattergl(x=x_axis, y=ins, mode="markers", name="Insertions"),
        go.Scattergl(x=x_axis, y=aln.deletions, mode="markers", name="Deletions"),
    ]
    layout = go.Layout(xaxis=dict(type="linear", autorange=True), yaxis=dict(type="linear", autorange=True))
    fig = go.Figure(data=traces, layout=layout)
    out_fn = os.path.splitext(os.path.split(bam_path)[1])[0]
    py.plot(fig, filename=out_fn + ".plot.html", auto_open=False)
