/*
 * kindel_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into or called by the product).
 *
 * Plain-C, single-threaded restatement of the reference hot path
 *     /root/reference/kindel/kindel.py  parse_records :21-128, consensus :369-381,
 *     consensus_sequence :384-430, depth min/max of build_report :450,477-479
 * operating on the same structure-of-arrays read batch the product's C-ABI takes
 * (include/kindel_hip.h, kd_batch): BAM-native 4-bit bases and u32 CIGAR words.
 *
 * It deliberately mirrors the reference statement by statement (sequential record loop,
 * sequential CIGAR loop, one increment per base) -- including Python's negative-index
 * wrap-around on lists and the exception the reference would raise -- so that it can act
 * as the checker for the HIP kernels at sizes where the Python reference is too slow.
 * PARITY IS PINNED: oracle/make_golden.py runs this file and the UNMODIFIED reference side by side (in the
 * build container) on every reference fixture and on the quirk cases of oracle/quirk_cases.py and asserts
 * equality of every table, insertion dict, consensus string and exception type; tests/golden/ holds the
 * reference's own outputs (tests/test_oracle_golden.py re-checks this file against them on any box), and
 * oracle/refbaseline.py compares the consensus again on the sample it times the reference on.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KO_OK 0
#define KO_E_BASE (-1)  /* KeyError:   base outside A,T,G,C,N          kindel.py:52,72,79 */
#define KO_E_RANGE (-2) /* IndexError: list index out of range         kindel.py:51-52,57,61,67,75 */
#define KO_E_CIGAR (-3) /* RuntimeError: mapped read with CIGAR '*'    kindel.py:47 */
#define KO_E_PATCH (-4) /* AttributeError: first patch at pos has seq None  kindel.py:397-398 */
#define KO_E_NOMEM (-5)

/* dict key order of the reference's per-site weight dicts, kindel.py:29 */
enum { CH_A = 0, CH_T = 1, CH_G = 2, CH_C = 3, CH_N = 4 };
static const char CH2CHR[5] = {'A', 'T', 'G', 'C', 'N'};
/* SAMv1 4.2.3 nibble alphabet */
static const char NIB2CHR[17] = "=ACMGRSVTWYHKDBN";

typedef struct ins_node {
    struct ins_node *next;
    uint32_t count, len;
    char *s; /* upper-case ASCII, not NUL terminated */
} ins_node;

typedef struct ko_aln {
    uint32_t L;
    uint32_t *weights;            /* [L][5]  A,T,G,C,N            kindel.py:29 */
    uint32_t *clip_start_weights; /* [L][5]                       kindel.py:30-32 */
    uint32_t *clip_end_weights;   /* [L][5]                       kindel.py:33-35 */
    uint32_t *clip_starts;        /* [L+1]                        kindel.py:36 */
    uint32_t *clip_ends;          /* [L+1]                        kindel.py:37 */
    uint32_t *deletions;          /* [L+1]                        kindel.py:39 */
    ins_node **ins_head, **ins_tail; /* [L+1] dicts in insertion order, kindel.py:38 */
    uint64_t n_ins_keys, n_ins_bytes;
    uint64_t n_reads_used, n_events_aligned, n_events_walked;
} ko_aln;

static int chan_of_nib(unsigned nib) {
    switch (nib) {
    case 1: return CH_A;
    case 2: return CH_C;
    case 4: return CH_G;
    case 8: return CH_T;
    case 15: return CH_N;
    default: return -1;
    }
}

static inline unsigned nib_at(const uint8_t *seq, uint64_t q) {
    uint8_t b = seq[q >> 1];
    return (q & 1) ? (b & 15u) : (b >> 4);
}

/* Python list indexing: negative wraps once, otherwise IndexError */
static inline int pyidx(int64_t i, int64_t n, int64_t *out) {
    if (i < 0) i += n;
    if (i < 0 || i >= n) return 0;
    *out = i;
    return 1;
}

void ko_free(ko_aln *a) {
    if (!a) return;
    if (a->ins_head) {
        for (uint64_t i = 0; i <= a->L; i++) {
            ins_node *n = a->ins_head[i];
            while (n) {
                ins_node *nx = n->next;
                free(n->s);
                free(n);
                n = nx;
            }
        }
    }
    free(a->weights);
    free(a->clip_start_weights);
    free(a->clip_end_weights);
    free(a->clip_starts);
    free(a->clip_ends);
    free(a->deletions);
    free(a->ins_head);
    free(a->ins_tail);
    free(a);
}

static int ins_add(ko_aln *a, int64_t site, const uint8_t *seq, uint64_t q0, uint64_t q1) {
    uint32_t len = (uint32_t)(q1 - q0);
    char stackbuf[256];
    char *buf = len <= sizeof stackbuf ? stackbuf : (char *)malloc(len);
    if (!buf) return KO_E_NOMEM;
    for (uint32_t i = 0; i < len; i++) buf[i] = NIB2CHR[nib_at(seq, q0 + i)];
    ins_node *n = a->ins_head[site];
    for (; n; n = n->next)
        if (n->len == len && memcmp(n->s, buf, len) == 0) break;
    if (n) {
        n->count++;
    } else {
        n = (ins_node *)calloc(1, sizeof *n);
        if (!n) return KO_E_NOMEM;
        n->s = (char *)malloc(len ? len : 1);
        memcpy(n->s, buf, len);
        n->len = len;
        n->count = 1;
        if (a->ins_tail[site]) a->ins_tail[site]->next = n;
        else a->ins_head[site] = n;
        a->ins_tail[site] = n;
        a->n_ins_keys++;
        a->n_ins_bytes += len;
    }
    if (buf != stackbuf) free(buf);
    return KO_OK;
}

/*
 * parse_records(ref_id, ref_len, records)  -- kindel.py:21-81.
 * `records` = the reads of the batch whose contig == contig_id, in batch order
 * (parse_bam groups by rname preserving order, kindel.py:143-151).
 * On a reference exception returns NULL and sets *err / *err_read (batch index).
 */
ko_aln *ko_parse_records(uint32_t contig_id, uint32_t ref_len, uint64_t n_reads,
                         const uint32_t *contig, const int32_t *pos0, const uint32_t *flag,
                         const uint64_t *seq_off, const uint32_t *seq_len,
                         const uint64_t *cig_off, const uint32_t *n_cig, const uint8_t *seq4,
                         const uint32_t *cigar, int *err, uint64_t *err_read) {
    const int64_t L = ref_len;
    ko_aln *a = (ko_aln *)calloc(1, sizeof *a);
    *err = KO_OK;
    if (!a) { *err = KO_E_NOMEM; return NULL; }
    a->L = ref_len;
    a->weights = (uint32_t *)calloc((size_t)L * 5 + 1, 4);
    a->clip_start_weights = (uint32_t *)calloc((size_t)L * 5 + 1, 4);
    a->clip_end_weights = (uint32_t *)calloc((size_t)L * 5 + 1, 4);
    a->clip_starts = (uint32_t *)calloc((size_t)L + 1, 4);
    a->clip_ends = (uint32_t *)calloc((size_t)L + 1, 4);
    a->deletions = (uint32_t *)calloc((size_t)L + 1, 4);
    a->ins_head = (ins_node **)calloc((size_t)L + 1, sizeof(ins_node *));
    a->ins_tail = (ins_node **)calloc((size_t)L + 1, sizeof(ins_node *));
    if (!a->weights || !a->clip_start_weights || !a->clip_end_weights || !a->clip_starts ||
        !a->clip_ends || !a->deletions || !a->ins_head || !a->ins_tail) {
        ko_free(a); *err = KO_E_NOMEM; return NULL;
    }
#define FAIL(code) do { *err = (code); if (err_read) *err_read = i; ko_free(a); return NULL; } while (0)
    for (uint64_t i = 0; i < n_reads; i++) {
        if (contig[i] != contig_id) continue;
        int64_t q = 0;                 /* :41 */
        int64_t r = pos0[i];           /* :42  record.pos - 1 */
        const int64_t sl = seq_len[i];
        if ((flag[i] & 0x4u) || sl <= 1) continue; /* :43-46 */
        const uint8_t *seq = seq4 + seq_off[i];
        const uint32_t *cg = cigar + cig_off[i];
        if (n_cig[i] == 0) FAIL(KO_E_CIGAR); /* :47 CIGAR '*' -> RuntimeError */
        a->n_reads_used++;
        for (uint32_t k = 0; k < n_cig[i]; k++) {
            const int64_t len = cg[k] >> 4;
            const unsigned op = cg[k] & 15u;
            int64_t ix;
            if (op == 0 || op == 7 || op == 8) { /* M = X  :49-54 */
                for (int64_t j = 0; j < len; j++) {
                    if (q >= sl) FAIL(KO_E_RANGE);
                    int ch = chan_of_nib(nib_at(seq, (uint64_t)q));
                    if (!pyidx(r, L, &ix)) FAIL(KO_E_RANGE);
                    if (ch < 0) FAIL(KO_E_BASE);
                    a->weights[ix * 5 + ch]++;
                    r++; q++;
                }
                a->n_events_aligned += (uint64_t)len;
                a->n_events_walked += (uint64_t)len;
            } else if (op == 1) { /* I  :55-58  (slice never raises) */
                int64_t q0 = q < sl ? q : sl, q1 = q + len < sl ? q + len : sl;
                if (!pyidx(r, L + 1, &ix)) FAIL(KO_E_RANGE);
                int rc = ins_add(a, ix, seq, (uint64_t)q0, (uint64_t)q1);
                if (rc) FAIL(rc);
                q += len;
                a->n_events_walked += (uint64_t)len;
            } else if (op == 2) { /* D  :59-62 */
                for (int64_t d = 0; d < len; d++) {
                    if (!pyidx(r + d, L + 1, &ix)) FAIL(KO_E_RANGE);
                    a->deletions[ix]++;
                }
                r += len;
                a->n_events_walked += (uint64_t)len;
            } else if (op == 4) { /* S */
                if (k == 0) { /* :64-73 */
                    if (!pyidx(r, L + 1, &ix)) FAIL(KO_E_RANGE);
                    a->clip_ends[ix]++;
                    for (int64_t g = 0; g < len; g++) {
                        if (g >= sl) FAIL(KO_E_RANGE);
                        int ch = chan_of_nib(nib_at(seq, (uint64_t)g));
                        int64_t rel = r - len + g;
                        if (rel >= 0) {
                            if (rel >= L) FAIL(KO_E_RANGE);
                            if (ch < 0) FAIL(KO_E_BASE);
                            a->clip_end_weights[rel * 5 + ch]++;
                        }
                    }
                    q += len;
                } else { /* :74-81 */
                    if (!pyidx(r - 1, L + 1, &ix)) FAIL(KO_E_RANGE);
                    a->clip_starts[ix]++;
                    for (int64_t j = 0; j < len; j++) {
                        if (q >= sl) FAIL(KO_E_RANGE);
                        int ch = chan_of_nib(nib_at(seq, (uint64_t)q));
                        if (r < L) {
                            if (!pyidx(r, L, &ix)) FAIL(KO_E_RANGE);
                            if (ch < 0) FAIL(KO_E_BASE);
                            a->clip_start_weights[ix * 5 + ch]++;
                            r++; q++;
                        }
                    }
                }
                a->n_events_walked += (uint64_t)len;
            }
            /* H, N, P and anything else: ignored entirely (:49-81 has no branch) */
        }
    }
#undef FAIL
    return a;
}

uint32_t ko_len(const ko_aln *a) { return a->L; }
const uint32_t *ko_weights(const ko_aln *a) { return a->weights; }
const uint32_t *ko_clip_start_weights(const ko_aln *a) { return a->clip_start_weights; }
const uint32_t *ko_clip_end_weights(const ko_aln *a) { return a->clip_end_weights; }
const uint32_t *ko_clip_starts(const ko_aln *a) { return a->clip_starts; }
const uint32_t *ko_clip_ends(const ko_aln *a) { return a->clip_ends; }
const uint32_t *ko_deletions(const ko_aln *a) { return a->deletions; }
uint64_t ko_ins_n(const ko_aln *a) { return a->n_ins_keys; }
uint64_t ko_ins_bytes(const ko_aln *a) { return a->n_ins_bytes; }
void ko_stats(const ko_aln *a, uint64_t *out3) {
    out3[0] = a->n_reads_used; out3[1] = a->n_events_aligned; out3[2] = a->n_events_walked;
}

/* enumerate the insertion dicts: site ascending, then dict insertion order */
void ko_ins_enumerate(const ko_aln *a, uint32_t *sites, uint32_t *counts, uint32_t *lens,
                      uint64_t *offs, uint8_t *bytes) {
    uint64_t k = 0, o = 0;
    for (uint64_t s = 0; s <= a->L; s++)
        for (ins_node *n = a->ins_head[s]; n; n = n->next) {
            sites[k] = (uint32_t)s; counts[k] = n->count; lens[k] = n->len; offs[k] = o;
            memcpy(bytes + o, n->s, n->len);
            o += n->len; k++;
        }
}

/* sum(insertions[pos].values())  -- kindel.py:402, :581 */
void ko_ins_totals(const ko_aln *a, uint32_t *out /* [L+1] */) {
    for (uint64_t s = 0; s <= a->L; s++) {
        uint32_t t = 0;
        for (ins_node *n = a->ins_head[s]; n; n = n->next) t += n->count;
        out[s] = t;
    }
}

/* consensus(weight) for a 5-channel site -- kindel.py:369-381 (proportion omitted: unused) */
static void site_consensus(const uint32_t *w, int *base, uint32_t *freq, int *tie) {
    uint32_t sum = w[0] + w[1] + w[2] + w[3] + w[4];
    if (!sum) { *base = CH_N; *freq = 0; *tie = 0; return; }
    int b = 0;
    for (int c = 1; c < 5; c++) if (w[c] > w[b]) b = c; /* first max in A,T,G,C,N order */
    int t = 0;
    for (int c = 0; c < 5; c++) if (c != b && w[c] == w[b]) t = 1;
    *base = b; *freq = w[b]; *tie = (w[b] != 0) && t;
}

/* derived arrays of parse_records -- kindel.py:83-96.  Each out pointer may be NULL. */
void ko_derived(const ko_aln *a, uint32_t *aligned_depth, uint32_t *consensus_depth,
                uint32_t *clip_start_depth, uint32_t *clip_end_depth, uint32_t *clip_depth) {
    for (uint64_t p = 0; p < a->L; p++) {
        const uint32_t *w = a->weights + p * 5, *s = a->clip_start_weights + p * 5,
                       *e = a->clip_end_weights + p * 5;
        int b, t; uint32_t f;
        site_consensus(w, &b, &f, &t);
        uint32_t ad = w[0] + w[1] + w[2] + w[3] + w[4];          /* :83 */
        uint32_t discordant = ad - w[b];                           /* :85-88 */
        uint32_t csd = s[CH_A] + s[CH_C] + s[CH_G] + s[CH_T];     /* :90-92 no N */
        uint32_t ced = e[CH_A] + e[CH_C] + e[CH_G] + e[CH_T];     /* :93-95 */
        if (aligned_depth) aligned_depth[p] = ad;
        if (consensus_depth) consensus_depth[p] = ad - discordant; /* :89 */
        if (clip_start_depth) clip_start_depth[p] = csd;
        if (clip_end_depth) clip_end_depth[p] = ced;
        if (clip_depth) clip_depth[p] = csd + ced;                 /* :96 */
    }
}

/* min/max of ACGT depth -- build_report kindel.py:450,477-479 */
void ko_depth_minmax(const ko_aln *a, uint32_t *out2) {
    uint32_t mn = 0xffffffffu, mx = 0;
    for (uint64_t p = 0; p < a->L; p++) {
        const uint32_t *w = a->weights + p * 5;
        uint32_t d = w[CH_A] + w[CH_C] + w[CH_G] + w[CH_T];
        if (d < mn) mn = d;
        if (d > mx) mx = d;
    }
    out2[0] = mn; out2[1] = mx;
}

/*
 * consensus_sequence(weights, insertions, deletions, cdr_patches, trim_ends, min_depth,
 *                    uppercase)  -- kindel.py:384-430.
 * Patches: arrays p_start/p_end (+ p_seq, NULL entry == Python None), in list order.
 * The float compares of :411-419 are restated as exact integer compares
 * (x > 0.5*y  <=>  2x > y for non-negative integers; doubles are exact at these sizes).
 * out must hold at least L + sum(insertion bytes) + sum(patch lengths) bytes.
 */
int ko_consensus_sequence(const ko_aln *a, uint32_t min_depth, int n_patches,
                          const int64_t *p_start, const int64_t *p_end, const char *const *p_seq,
                          int trim_ends, int uppercase, char *out, uint64_t cap,
                          uint64_t *out_len, uint8_t *changes) {
    const int64_t L = a->L;
    uint64_t n = 0;
    int64_t skip = 0;
    memset(changes, 0, (size_t)L);
#define PUT(c) do { if (n >= cap) return KO_E_NOMEM; out[n++] = (char)(c); } while (0)
    for (int64_t pos = 0; pos < L; pos++) {
        if (skip) { skip -= 1; continue; } /* :393-395 (negative stays truthy) */
        if (n_patches) {                   /* :396-401 */
            int any = 0;
            for (int k = 0; k < n_patches; k++)
                if (p_start[k] == pos && p_seq[k] && p_seq[k][0]) { any = 1; break; }
            if (any) {
                int k = 0;
                while (p_start[k] != pos) k++;
                if (!p_seq[k]) return KO_E_PATCH;
                for (const char *c = p_seq[k]; *c; c++)
                    PUT((*c >= 'A' && *c <= 'Z') ? *c + 32 : *c);
                skip += (p_end[k] - p_start[k]) - 1;
                continue;
            }
        }
        const uint32_t *w = a->weights + pos * 5;
        uint64_t ins_freq = 0; /* :402 */
        for (ins_node *nd = a->ins_head[pos]; nd; nd = nd->next) ins_freq += nd->count;
        uint64_t del_freq = a->deletions[pos];                                     /* :403 */
        uint64_t ad = (uint64_t)w[CH_A] + w[CH_C] + w[CH_G] + w[CH_T];             /* :404 */
        uint64_t ad_next = 0;                                                      /* :405-410 */
        if (pos + 1 < L) {
            const uint32_t *wn = w + 5;
            ad_next = (uint64_t)wn[CH_A] + wn[CH_C] + wn[CH_G] + wn[CH_T];
        }
        uint64_t thr2 = ad;                                   /* 2*threshold_freq  :411 */
        uint64_t ind2 = ad < ad_next ? ad : ad_next;          /* 2*indel_threshold :412 */
        if (2 * del_freq > thr2) {                            /* :413-414 */
            changes[pos] = 'D';
        } else if (ad < min_depth) {                          /* :415-417 */
            PUT('N');
            changes[pos] = 'N';
        } else {
            if (2 * ins_freq > ind2) {                        /* :419-422 */
                ins_node *best = a->ins_head[pos];
                int tie = 0;
                for (ins_node *nd = best ? best->next : NULL; nd; nd = nd->next)
                    if (nd->count > best->count) best = nd;
                for (ins_node *nd = a->ins_head[pos]; nd; nd = nd->next)
                    if (nd != best && nd->count == best->count) tie = 1;
                if (!tie) {
                    for (uint32_t j = 0; j < best->len; j++) {
                        char c = best->s[j];
                        PUT((c >= 'A' && c <= 'Z') ? c + 32 : c);
                    }
                } else {
                    PUT('N');
                }
                changes[pos] = 'I';
            }
            int b, t; uint32_t f;                             /* :423-424 */
            site_consensus(w, &b, &f, &t);
            PUT(t ? 'N' : CH2CHR[b]);
        }
    }
#undef PUT
    uint64_t lo = 0, hi = n;
    if (trim_ends) { /* :425-426 str.strip("N") */
        while (lo < hi && out[lo] == 'N') lo++;
        while (hi > lo && out[hi - 1] == 'N') hi--;
        memmove(out, out + lo, hi - lo);
    }
    n = hi - lo;
    if (uppercase) /* :427-428 */
        for (uint64_t j = 0; j < n; j++)
            if (out[j] >= 'a' && out[j] <= 'z') out[j] -= 32;
    *out_len = n;
    return KO_OK;
}
