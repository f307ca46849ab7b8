"""kd_host_inflate (kindel_amd/csrc/kd_inflate.h): the BGZF reader's raw-DEFLATE block decoder against zlib.

The decoder's contract: the stream must decode to exactly the announced size, malformed / truncated input is refused, and
nothing is written outside the output buffer (the binding puts a canary behind it).  zlib is the checker here.
"""
import os
import random
import zlib

import pytest

from kindel_amd import _native as N


def raw_deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=9):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strategy)
    return c.compress(data) + c.flush()


def _payloads():
    rng = random.Random(7)
    bam_like = b"".join(
        bytes([rng.randrange(256) for _ in range(8)]) + b"r%07d\0" % i + bytes(rng.choice(b"\x11\x12\x14\x18\x21\x22\x24\x28\x41\x42\x44\x48\x81\x82\x84\x88")
                                                                                   for _ in range(75)) + b"I" * 150
        for i in range(400))
    return {
        "empty": b"",
        "one_byte": b"a",
        "run": b"a" * 70000,                                   # distance 1, maximum-length matches
        "short_period": b"abcde" * 9000,                       # distances < 8
        "random": bytes(rng.getrandbits(8) for _ in range(65536)),   # stored blocks at any level
        "acgt": bytes(rng.choice(b"ACGT") for _ in range(150000)),
        "acgt_runs": b"".join(bytes([rng.choice(b"ACGT")]) * rng.randint(1, 40) for _ in range(12000)),
        "bam_like": bam_like,
        "all_bytes": bytes(range(256)) * 200,
        "far_matches": os.urandom(300) + b"x" * 32000 + os.urandom(300) * 2,
    }


PAYLOADS = _payloads()


@pytest.mark.parametrize("name", sorted(PAYLOADS))
@pytest.mark.parametrize("level", [0, 1, 6, 9])
@pytest.mark.parametrize("strategy", [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE],
                         ids=["default", "fixed", "huffman_only", "rle"])
def test_round_trip_against_zlib(emu_lib, name, level, strategy):
    data = PAYLOADS[name]
    z = raw_deflate(data, level, strategy)
    assert N.host_inflate(z, len(data), lib=emu_lib) == data
    for wrong in {len(data) + 1, max(len(data) - 1, 0)} - {len(data)}:
        with pytest.raises(ValueError):
            N.host_inflate(z, wrong, lib=emu_lib)
    if len(data) > 0 and len(z) > 4:
        with pytest.raises(ValueError):
            N.host_inflate(z[:len(z) // 2], len(data), lib=emu_lib)


def test_many_small_blocks_and_window_sizes(emu_lib):
    rng = random.Random(3)
    data = PAYLOADS["bam_like"]
    for mem in (1, 4, 9):
        c = zlib.compressobj(6, zlib.DEFLATED, -15, mem)
        z = b""
        at = 0
        while at < len(data):       # Z_FULL_FLUSH: many deflate blocks (incl. empty stored ones) in one stream
            n = rng.randint(1, 3000)
            z += c.compress(data[at:at + n]) + c.flush(zlib.Z_FULL_FLUSH)
            at += n
        z += c.flush()
        assert N.host_inflate(z, len(data), lib=emu_lib) == data


def test_corrupted_streams_never_crash_or_overrun(emu_lib):
    rng = random.Random(11)
    data = PAYLOADS["bam_like"]
    z = raw_deflate(data, 6)
    refused = 0
    for _ in range(1500):
        zz = bytearray(z)
        for _ in range(rng.randint(1, 4)):
            zz[rng.randrange(len(zz))] ^= 1 << rng.randrange(8)
        try:
            out = N.host_inflate(bytes(zz), len(data), lib=emu_lib)   # the binding asserts the canary behind the output
            assert len(out) == len(data)
        except ValueError:
            refused += 1
    assert refused > 500          # (a flipped bit can still give a well-formed stream of the same length: no CRC at this level)


def test_bgzf_reader_uses_it(emu_lib, tmp_path):
    from tools import synth
    batch = synth.to_numpy(synth.short_reads([3000], 40, seed=2))
    p = str(tmp_path / "x.bam")
    N.write_bam(p, batch, lib=emu_lib)
    from tests.test_decoder import same_batch
    same_batch(N.decode_file(p, lib=emu_lib), batch)


def test_crc32_against_zlib(emu_lib):
    """kd_host_crc32 (kd_crc32.h: carry-less-multiply folding + zlib for the tail) == zlib.crc32 at every length around the
    folding's block sizes, and on buffers larger than a BGZF block."""
    rng = random.Random(5)
    blob = bytes(rng.getrandbits(8) for _ in range(70000))
    for n in list(range(0, 300)) + [1023, 1024, 1025, 4095, 65279, 65280, 65536, 70000]:
        for off in (0, 1, 7):
            assert N.host_crc32(blob[off:off + n], lib=emu_lib) == zlib.crc32(blob[off:off + n]), (n, off)
    assert N.host_crc32(b"\0" * 100000, lib=emu_lib) == zlib.crc32(b"\0" * 100000)
    assert N.host_crc32(b"\xff" * 65280, lib=emu_lib) == zlib.crc32(b"\xff" * 65280)


class _Bits:
    def __init__(self):
        self.acc = 0
        self.n = 0

    def put(self, value, nbits):          # LSB first (header fields, extra bits)
        self.acc |= (value & ((1 << nbits) - 1)) << self.n
        self.n += nbits

    def code(self, code, nbits):          # a Huffman code word: most significant bit first
        for k in range(nbits - 1, -1, -1):
            self.put((code >> k) & 1, 1)

    def bytes(self):
        return self.acc.to_bytes((self.n + 7) // 8, "little")


def _dynamic_block(lit_len_a, lit_len_eob, cl_lens):
    """One final dynamic-Huffman block that says b"a": literal 'a' and end-of-block are the only literal / length symbols (code
    lengths `lit_len_a`, `lit_len_eob`), no distance code; cl_lens = lengths of the code-length symbols (0, 2 or 1, 18)."""
    sym_len = sorted({lit_len_a, lit_len_eob})          # code-length symbols in use besides 0 and 18
    b = _Bits()
    b.put(1, 1); b.put(2, 2)                            # BFINAL, BTYPE = dynamic
    b.put(0, 5); b.put(0, 5); b.put(19 - 4, 4)          # HLIT = 257, HDIST = 1, HCLEN = 19
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    for s in order:
        b.put(cl_lens.get(s, 0), 3)
    # canonical code of the code-length alphabet
    items = sorted((l, s) for s, l in cl_lens.items() if l)
    code, prev, cl_code = 0, items[0][0], {}
    for l, s in items:
        code <<= l - prev
        prev = l
        cl_code[s] = (code, l)
        code += 1

    def zeros(n):
        while n:
            k = min(n, 138)
            assert k >= 11
            b.code(*cl_code[18]); b.put(k - 11, 7)
            n -= k
    zeros(97); b.code(*cl_code[lit_len_a]); zeros(138); zeros(20); b.code(*cl_code[lit_len_eob]); b.code(*cl_code[0])
    # the data: 'a', end of block (canonical: shorter code first, then symbol order)
    if lit_len_a == lit_len_eob:
        b.code(0, lit_len_a); b.code(1, lit_len_eob)
    elif lit_len_a < lit_len_eob:
        b.code(0, lit_len_a); b.code(1 << (lit_len_eob - lit_len_a), lit_len_eob)
    else:
        b.code(1 << (lit_len_a - lit_len_eob), lit_len_a); b.code(0, lit_len_eob)
    assert sym_len
    return b.bytes()


def test_incomplete_codes_are_refused_like_zlib(emu_lib):
    """zlib (and htslib with it) refuses an incomplete literal / length code and an incomplete code-length code even if the
    stream never uses the missing words (inflate_table: "incomplete set"); so does this decoder."""
    good = _dynamic_block(1, 1, {0: 2, 1: 2, 18: 1})                 # 'a' and EOB: two 1-bit words, a complete code
    assert zlib.decompress(good, -15) == b"a"
    assert N.host_inflate(good, 1, lib=emu_lib) == b"a"
    bad_lit = _dynamic_block(2, 2, {0: 2, 2: 2, 18: 1})              # two 2-bit words: half of the code space unassigned
    with pytest.raises(zlib.error):
        zlib.decompress(bad_lit, -15)
    with pytest.raises(ValueError):
        N.host_inflate(bad_lit, 1, lib=emu_lib)
    bad_cl = _dynamic_block(1, 1, {0: 2, 1: 2, 18: 2})               # code-length code with three 2-bit words
    with pytest.raises(zlib.error):
        zlib.decompress(bad_cl, -15)
    with pytest.raises(ValueError):
        N.host_inflate(bad_cl, 1, lib=emu_lib)


def test_bgzf_crc_is_verified(emu_lib, tmp_path):
    """A BGZF block whose CRC-32 trailer does not match its inflated bytes is refused (htslib: "CRC32 checksum mismatch"),
    whole-file and streamed; the untouched file decodes."""
    import struct
    from tools import synth
    batch = synth.to_numpy(synth.short_reads([3000], 40, seed=4))
    p = str(tmp_path / "x.bam")
    N.write_bam(p, batch, lib=emu_lib)
    raw = bytearray(open(p, "rb").read())
    assert N.decode_file(p, lib=emu_lib)["contig"].size == batch["contig"].size
    bsize = struct.unpack_from("<H", raw, 16)[0] + 1                 # first block: header with the BC subfield at offset 12
    crc_at = bsize - 8
    assert zlib.crc32(zlib.decompress(bytes(raw[18:crc_at]), -15)) == struct.unpack_from("<I", raw, crc_at)[0]
    raw[crc_at] ^= 0x40
    q = str(tmp_path / "bad_crc.bam")
    open(q, "wb").write(bytes(raw))
    with pytest.raises(Exception):
        N.decode_file(q, lib=emu_lib)
    with pytest.raises(Exception):
        st = N.Stream(q, lib=emu_lib)
        while st.next_batch() is not None:
            pass
