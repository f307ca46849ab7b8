"""EXPERIMENT RECORD / TEST TOOL (round 5): `python bam_mutations.py N SEED0 OUTDIR [BAM]` -- N mutated copies of a BAM file whose BGZF
framing stays VALID (the uncompressed stream is mutated -- bit flips, truncation, random runs, extreme length fields, deletions,
insertions, mostly behind the header -- and re-blocked with correct CRCs), so the mutations reach the BAM record parser instead of
dying at the block CRC.  Fed to scripts/exp/decoder_asan.cpp and scripts/exp/emu_mutated_files.py."""
import gzip, zlib, struct, random, sys, os
raw = gzip.open(sys.argv[4] if len(sys.argv) > 4 else '/tmp/tsan/s.bam','rb').read()
def bgzf(data, bs):
    out = bytearray()
    for o in range(0, len(data), bs):
        chunk = data[o:o+bs]
        c = zlib.compressobj(6, zlib.DEFLATED, -15); comp = c.compress(chunk) + c.flush()
        bsize = len(comp) + 25
        out += struct.pack('<4BI2BH2BHH', 31,139,8,4, 0, 0,255, 6, 66,67, 2, bsize) + comp + struct.pack('<II', zlib.crc32(chunk) & 0xffffffff, len(chunk))
    out += bytes.fromhex('1f8b08040000000000ff0600424302001b0003000000000000000000')
    return bytes(out)
n, seed0, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
os.makedirs(outdir, exist_ok=True)
# header length: find end of header to aim most mutations at the records
l_text = struct.unpack_from('<i', raw, 4)[0]; p = 8 + l_text; n_ref = struct.unpack_from('<i', raw, p)[0]; p += 4
for _ in range(n_ref):
    l = struct.unpack_from('<i', raw, p)[0]; p += 4 + l + 4
hdr_end = p
for it in range(n):
    rng = random.Random(seed0 + it)
    m = bytearray(raw)
    kind = rng.randrange(6)
    lo = 0 if rng.random() < 0.15 else hdr_end
    if kind == 0:
        for _ in range(rng.randint(1, 6)): m[rng.randrange(lo, len(m))] ^= 1 << rng.randrange(8)
    elif kind == 1:
        m = m[:rng.randrange(lo, len(m))]
    elif kind == 2:
        a = rng.randrange(lo, len(m))
        for j in range(a, min(len(m), a + rng.randint(1, 40))): m[j] = rng.randrange(256)
    elif kind == 3:   # a length-like field set to an extreme
        a = rng.randrange(lo, len(m) - 4) & ~3
        struct.pack_into('<I', m, a, rng.choice([0, 1, 0xffffffff, 0x7fffffff, 0x80000000, 65535, 65536, 1 << 24]))
    elif kind == 4:
        a = rng.randrange(lo, len(m)); del m[a:a + rng.randint(1, 500)]
    else:
        a = rng.randrange(lo, len(m)); m[a:a] = bytes(rng.randrange(256) for _ in range(rng.randint(1, 200)))
    open(os.path.join(outdir, 'm%05d.bam' % it), 'wb').write(bgzf(bytes(m), rng.choice([700, 4000, 60000])))
print("wrote", n)
