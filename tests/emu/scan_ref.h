// scan_ref.h -- TEST INFRASTRUCTURE ONLY: the 64-bit, branch-per-op-kind statement of k_prep's CIGAR scan (rounds 1 - 2 of
// kd_prep.h), kept as the checker of the product's 32-bit branch-free kd_scan_cigar: tests/test_emu_kernels.py runs both on
// random and adversarial CIGARs through kd_emu_scan_compare (emu_lib.cpp).  Follows kindel.py:40-81 seen from the cursors.
#pragma once
#include <stdint.h>
struct KdScanRef { uint32_t cls, cold, lead; uint64_t span, n_ins, ins_bases, aligned, walked; };
static inline KdScanRef kd_scan_cigar_ref(const uint32_t *cg, uint32_t nc, int64_t pos0, int64_t sl, int64_t L) {
    KdScanRef s;
    s.cls = 1; s.cold = 0; s.lead = 0; s.span = 0; s.n_ins = 0; s.ins_bases = 0; s.aligned = 0; s.walked = 0;
    bool regular = pos0 >= 0;
    bool seen_nfs = false;  // a non-first S was seen: r is no longer plain prefix arithmetic
    int64_t r = pos0, q = 0, hot_hi = pos0;
    for (uint32_t k = 0; k < nc; k++) {
        const uint32_t c = cg[k];
        const int64_t len = c >> 4;
        const uint32_t op = c & 15u;
        if (op == 0 || op == 7 || op == 8) {  // M = X
            if (seen_nfs || r + len > L || q + len > sl) regular = false;
            r += len; q += len; hot_hi = r;
            s.aligned += (uint64_t)len; s.walked += (uint64_t)len;
        } else if (op == 1) {  // I
            s.cold = 4;
            if (seen_nfs || r > L) regular = false;
            int64_t q0 = q < sl ? q : sl, q1 = q + len < sl ? q + len : sl;
            s.n_ins += 1; s.ins_bases += (uint64_t)(q1 - q0);
            q += len; s.walked += (uint64_t)len;
        } else if (op == 2) {  // D
            if (seen_nfs || r + len > L + 1) regular = false;
            r += len; hot_hi = r;
            s.walked += (uint64_t)len;
        } else if (op == 4) {  // S
            s.cold = 4;
            s.walked += (uint64_t)len;
            if (k == 0) {
                if (r > L || len > sl) regular = false;
                s.lead = (uint32_t)(len < r ? len : (r > 0 ? r : 0));
                q += len;
            } else {
                if (seen_nfs || r - 1 > L) regular = false;   // clip_starts[r - 1] must exist (kindel.py:75)
                seen_nfs = true;
                int64_t n_adv = r < L ? (len < L - r ? len : L - r) : 0;
                if (n_adv > sl - q || (len > n_adv && q + n_adv >= sl)) regular = false;
                r += n_adv; q += n_adv;
                hot_hi = r;  // the clip_start_weights writes extend the read's footprint
            }
        }
    }
    if (!regular) s.cls = 2;
    s.span = hot_hi > pos0 ? (uint64_t)(hot_hi - pos0) : 0;
    return s;
}
