"""TEST INFRASTRUCTURE ONLY -- hand-written SAM inputs that exercise every branch and
quirk of the reference's CIGAR walk (/root/reference/kindel/kindel.py:40-81) and of
consensus_sequence (:384-430); see SURVEY.md section 8a "#1 semantics".

oracle/make_golden.py feeds each case to the *unmodified* reference and stores what it
returned (tables, consensus, or the exception type) in tests/golden/quirks.json.
"""

_HDR1 = "@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:c1\tLN:30\n"
_HDR2 = "@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:c1\tLN:30\n@SQ\tSN:c2\tLN:20\n"


def _r(name, flag, rname, pos, cigar, seq):
    return "\t".join([name, str(flag), rname, str(pos), "60", cigar, "*", "0", "0", seq,
                      "*" if seq == "*" else "I" * len(seq)]) + "\n"


def _mk(hdr, *reads):
    return hdr + "".join(_r(*r) for r in reads)


A10 = "ACGTACGTAC"
CASES = {
    "plain_M": _mk(_HDR1, ("a", 0, "c1", 1, "10M", A10), ("b", 16, "c1", 5, "10M", "TTTTTGGGGG")),
    "pos0_wraps_to_last_site": _mk(_HDR1, ("a", 0, "c1", 0, "5M", "ACGTN")),
    "pos0_deletion_first": _mk(_HDR1, ("a", 0, "c1", 0, "2D5M", "ACGTA")),
    "pos0_leading_softclip": _mk(_HDR1, ("a", 0, "c1", 0, "3S4M", "ACGTACG")),
    "pos0_insertion_first": _mk(_HDR1, ("a", 0, "c1", 0, "2I4M", "ACGTAC")),
    "pos0_nonfirst_softclip": _mk(_HDR1, ("a", 0, "c1", 0, "2H3S4M", "ACGTACG")),
    "negative_pos": _mk(_HDR1, ("a", 0, "c1", -4, "8M", "ACGTACGT")),
    "leading_S_at_pos1": _mk(_HDR1, ("a", 0, "c1", 1, "4S6M", A10)),
    "leading_S_partial_overlap": _mk(_HDR1, ("a", 0, "c1", 3, "4S6M", A10)),
    "leading_S_inside": _mk(_HDR1, ("a", 0, "c1", 11, "4S6M", A10), ("b", 0, "c1", 11, "2S8M", A10)),
    "H_then_S_is_nonfirst": _mk(_HDR1, ("a", 0, "c1", 5, "3H4S6M", A10)),
    "trailing_S": _mk(_HDR1, ("a", 0, "c1", 5, "6M4S", A10)),
    "trailing_S_clamped_at_L": _mk(_HDR1, ("a", 0, "c1", 24, "6M4S", A10)),
    "trailing_S_exactly_to_L": _mk(_HDR1, ("a", 0, "c1", 21, "6M4S", A10)),
    "trailing_S_starts_at_L": _mk(_HDR1, ("a", 0, "c1", 25, "6M4S", A10)),
    "trailing_S_then_H": _mk(_HDR1, ("a", 0, "c1", 5, "6M4S5H", A10)),
    "middle_S_advances_ref": _mk(_HDR1, ("a", 0, "c1", 5, "3M2S5M", A10)),
    "two_leading_S": _mk(_HDR1, ("a", 0, "c1", 8, "2S2S6M", A10)),
    "only_softclip": _mk(_HDR1, ("a", 0, "c1", 8, "10S", A10)),
    "N_op_ignored_no_advance": _mk(_HDR1, ("a", 0, "c1", 2, "4M100N6M", A10)),
    "P_eq_X_ops": _mk(_HDR1, ("a", 0, "c1", 2, "3=2X1P5M", "ACGTACGTACG")),
    "lowercase_bases": _mk(_HDR1, ("a", 0, "c1", 2, "10M", "acgtnACGTN")),
    "insertion_simple": _mk(_HDR1, ("a", 0, "c1", 2, "4M2I4M", A10), ("b", 0, "c1", 2, "4M2I4M", A10),
                            ("c", 0, "c1", 2, "10M", A10)),
    "insertion_at_site0": _mk(_HDR1, ("a", 0, "c1", 1, "3I7M", A10), ("b", 0, "c1", 1, "3I7M", A10)),
    "insertion_at_L": _mk(_HDR1, ("a", 0, "c1", 25, "6M4I", A10)),
    "insertion_iupac_allowed": _mk(_HDR1, ("a", 0, "c1", 2, "4M2I4M", "ACGTRYACGT"),
                                   ("b", 0, "c1", 2, "4M2I4M", "ACGTRYACGT")),
    "insertion_zero_length": _mk(_HDR1, ("a", 0, "c1", 2, "4M0I6M", A10), ("b", 0, "c1", 2, "4M0I6M", A10)),
    "insertion_slice_truncated": _mk(_HDR1, ("a", 0, "c1", 2, "6M9I", A10), ("b", 0, "c1", 2, "6M9I", A10)),
    "insertion_slice_empty": _mk(_HDR1, ("a", 0, "c1", 2, "10M3I", A10)),
    "insertion_tie": _mk(_HDR1, ("a", 0, "c1", 2, "4M2I4M", "ACGTAAACGT"), ("b", 0, "c1", 2, "4M2I4M", "ACGTCCACGT")),
    "insertion_majority_unique": _mk(_HDR1, ("a", 0, "c1", 2, "4M2I4M", "ACGTAAACGT"),
                                     ("b", 0, "c1", 2, "4M2I4M", "ACGTAAACGT"),
                                     ("c", 0, "c1", 2, "4M2I4M", "ACGTCCACGT")),
    "insertion_minority": _mk(_HDR1, ("a", 0, "c1", 2, "4M2I4M", "ACGTAAACGT"), ("b", 0, "c1", 2, "8M", "ACGTACGT"),
                              ("c", 0, "c1", 2, "8M", "ACGTACGT"), ("d", 0, "c1", 2, "8M", "ACGTACGT")),
    "insertion_next_site_low_depth": _mk(_HDR1, ("a", 0, "c1", 2, "4M2I", "ACGTAA"), ("b", 0, "c1", 2, "4M", "ACGT"),
                                         ("c", 0, "c1", 2, "4M", "ACGT")),
    "deletion_simple": _mk(_HDR1, ("a", 0, "c1", 2, "4M3D6M", A10), ("b", 0, "c1", 2, "4M3D6M", A10),
                           ("c", 0, "c1", 2, "10M", A10)),
    "deletion_half_is_not_majority": _mk(_HDR1, ("a", 0, "c1", 2, "4M3D6M", A10), ("b", 0, "c1", 2, "10M", A10),
                                         ("c", 0, "c1", 2, "10M", A10)),
    "deletion_to_slot_L": _mk(_HDR1, ("a", 0, "c1", 25, "4M2D", "ACGT")),
    "base_tie": _mk(_HDR1, ("a", 0, "c1", 2, "6M", "ACGTAC"), ("b", 0, "c1", 2, "6M", "ACCTAG")),
    "base_tie_three_way": _mk(_HDR1, ("a", 0, "c1", 2, "4M", "ACGT"), ("b", 0, "c1", 2, "4M", "CCGT"),
                              ("c", 0, "c1", 2, "4M", "GCGT")),
    "N_majority": _mk(_HDR1, ("a", 0, "c1", 2, "4M", "NNGT"), ("b", 0, "c1", 2, "4M", "NAGT"),
                      ("c", 0, "c1", 2, "4M", "AAGT")),
    "unmapped_and_short_reads_skipped": _mk(_HDR1, ("a", 4, "c1", 2, "4M", "ACGT"), ("b", 0, "c1", 2, "1M", "A"),
                                            ("c", 0, "c1", 2, "*", "*"), ("d", 0, "c1", 3, "4M", "ACGT")),
    "unmapped_placed_only_contig_all_N": _mk(_HDR2, ("a", 4, "c2", 2, "4M", "ACGT"), ("b", 0, "c1", 3, "4M", "ACGT")),
    "star_rname_dropped": _mk(_HDR2, ("a", 4, "*", 0, "*", "ACGT"), ("b", 0, "c1", 3, "4M", "ACGT")),
    "contig_order_first_appearance": _mk(_HDR2, ("a", 0, "c2", 3, "4M", "ACGT"), ("b", 0, "c1", 3, "4M", "ACGT"),
                                         ("c", 0, "c2", 5, "4M", "ACGT")),
    "secondary_supplementary_dup_counted": _mk(_HDR1, ("a", 256, "c1", 2, "4M", "ACGT"), ("b", 2048, "c1", 2, "4M", "ACGT"),
                                               ("c", 1024, "c1", 2, "4M", "ACGT"), ("d", 512, "c1", 2, "4M", "ACGT")),
    "many_ops_long_cigar": _mk(_HDR1, ("a", 0, "c1", 1, "1M1I1M1D1M1I1M1D1M1I1M1D1M1I1M1D2M", "ACGTACGTACGTAC"),
                               ("b", 0, "c1", 1, "1M1I1M1D1M1I1M1D1M1I1M1D1M1I1M1D2M", "ACGTACGTACGTAC"),
                               ("c", 0, "c1", 1, "14M", "ACGTACGTACGTAC")),
    "clip_both_ends_pair": _mk(_HDR1, ("a", 0, "c1", 10, "3S5M2S", A10), ("b", 16, "c1", 12, "2S6M2S", A10),
                               ("c", 0, "c1", 10, "3S5M2S", "TTTACGTAGG")),
    # ---- cases where the reference raises ----
    "ERR_M_overhang_past_L": _mk(_HDR1, ("a", 0, "c1", 25, "10M", A10)),
    "ERR_iupac_in_M": _mk(_HDR1, ("a", 0, "c1", 2, "10M", "ACGTRCGTAC")),
    "ERR_iupac_in_leading_S_inside": _mk(_HDR1, ("a", 0, "c1", 11, "4S6M", "ACRTACGTAC")),
    "OK_iupac_in_leading_S_offref": _mk(_HDR1, ("a", 0, "c1", 1, "4S6M", "ARYTACGTAC")),
    "ERR_iupac_in_trailing_S": _mk(_HDR1, ("a", 0, "c1", 5, "6M4S", "ACGTACGRAC")),
    "OK_iupac_in_trailing_S_past_L": _mk(_HDR1, ("a", 0, "c1", 25, "6M4S", "ACGTACGRAC")),
    "ERR_cigar_star_mapped": _mk(_HDR1, ("a", 0, "c1", 2, "*", "ACGT")),
    "ERR_seq_shorter_than_M": _mk(_HDR1, ("a", 0, "c1", 2, "10M", "ACGT")),
    "ERR_seq_shorter_than_leading_S": _mk(_HDR1, ("a", 0, "c1", 12, "10S", "ACGT")),
    "ERR_deletion_past_slot_L": _mk(_HDR1, ("a", 0, "c1", 26, "4M3D", "ACGT")),
    "deletion_exactly_to_slot_L": _mk(_HDR1, ("a", 0, "c1", 25, "4M3D", "ACGT")),
    "ERR_pos_far_past_L": _mk(_HDR1, ("a", 0, "c1", 40, "4M", "ACGT")),
    "ERR_insertion_past_slot_L": _mk(_HDR1, ("a", 0, "c1", 25, "4M3D2I", "ACGTAA")),
    "ERR_leading_S_past_slot_L": _mk(_HDR1, ("a", 0, "c1", 32, "4S", "ACGT")),
    "ERR_nonfirst_S_past_slot_L": _mk(_HDR1, ("a", 0, "c1", 33, "4P0H6S", "ACGTA")),
    "nonfirst_S_at_slot_L": _mk(_HDR1, ("a", 0, "c1", 32, "4P0H6S", "ACGTA")),
    "ERR_eq_base_in_M": _mk(_HDR1, ("a", 0, "c1", 2, "4M", "AC=T")),
    "ERR_second_contig_only": _mk(_HDR2, ("a", 0, "c1", 3, "4M", "ACGT"), ("b", 0, "c2", 18, "6M", "ACGTAC")),
    # several invalid reads: the reference walks contig by contig in order of first appearance (kindel.py:143-151), so
    # it raises for the first failing read of the earliest-appearing contig, not for the first failing read of the file
    "ERR_order_first_contig_wins_IndexError": _mk(_HDR2, ("a", 0, "c1", 3, "4M", "ACGT"), ("b", 0, "c2", 3, "4M", "ACRT"),
                                                  ("c", 0, "c1", 28, "6M", "ACGTAC")),
    "ERR_order_first_contig_wins_KeyError": _mk(_HDR2, ("a", 0, "c2", 3, "4M", "ACGT"), ("b", 0, "c1", 28, "6M", "ACGTAC"),
                                                ("c", 0, "c2", 3, "4M", "ACRT")),
    "ERR_order_within_contig_file_order": _mk(_HDR2, ("a", 0, "c1", 28, "6M", "ACGTAC"), ("b", 0, "c1", 3, "4M", "ACRT")),
    # round 5 (found by a fuzz campaign): a non-first S of length 0 looks nothing up (kindel.py:77 loops over range(0)) -- also when an
    # insertion has carried the query cursor past the read's end (an I only slices); with length >= 1 the same cursor is an IndexError
    "zero_length_S_behind_an_overshooting_I": _mk(_HDR1, ("a", 0, "c1", 2, "6M9I0S", A10), ("b", 0, "c1", 2, "10M", A10)),
    "zero_length_S_then_more_ops": _mk(_HDR1, ("a", 0, "c1", 2, "6M9I0S2D0S1P", A10), ("b", 0, "c1", 4, "8M", "ACGTACGT")),
    "ERR_one_base_S_behind_an_overshooting_I": _mk(_HDR1, ("a", 0, "c1", 2, "6M9I1S", A10)),
}

#: option sets every non-error case is run with: (min_depth, trim_ends, uppercase)
OPTION_SETS = [(1, False, False), (0, False, False), (2, False, False), (3, True, False), (1, True, True)]


def long_cases():
    """Hand-built reads with more than 16 CIGAR words (the long-read path, kd_long.h: tiles of 256 ops, rows, "+ins" symbols):
    name -> (SAM text, expected exception or None).  Every case also holds a few ordinary reads over the same sites."""
    import random
    rng = random.Random(99)
    L = 4000

    def filler(n_ops, rr):
        """n_ops ops alternating short M runs with I / D / N / P ops, ending on an M"""
        ops = []
        while len(ops) < n_ops - 1:
            ops.append("%dM" % rr.randint(1, 9))
            ops.append(rr.choice(["1I", "2I", "1D", "3D", "2N", "1P", "1I", "1D"]))
        ops = ops[: n_ops - 1] + ["%dM" % rr.randint(1, 9)]
        return ops

    def qlen(ops):
        return sum(int(o[:-1]) for o in ops if o[-1] in "MIS=X")

    def sam_of(reads, extra_len=L):
        txt = "@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:c\tLN:%d\n" % extra_len
        for n, (pos, ops, seq) in enumerate(reads):
            txt += "r%d\t0\tc\t%d\t60\t%s\t*\t0\t0\t%s\t*\n" % (n, pos + 1, "".join(ops), seq)
        return txt

    def rseq(n, alphabet="ACGTN"):
        return "".join(rng.choice(alphabet) for _ in range(n))

    shorts = [(p, ["60M"], rseq(60)) for p in range(0, 900, 37)]
    cases = {}

    def add(name, pos, ops, exc=None, seq=None, shorts_=shorts, Lc=L):
        cases[name] = (sam_of([(pos, ops, seq if seq is not None else rseq(qlen(ops)))] + list(shorts_), Lc), exc)

    rr = random.Random(5)
    add("ins_first_after_clip", 40, ["5S", "3I"] + filler(40, rr))
    add("ins_first", 0, ["2I"] + filler(30, rr))
    add("ends_on_ins", 100, filler(33, rr) + ["2I"])
    add("ends_on_ins_then_clip", 100, filler(33, rr) + ["2I", "6S"])
    add("two_ins_one_site_N", 50, filler(21, rr) + ["2I", "3N", "1I"] + filler(5, rr))
    add("three_ins_one_site", 50, filler(21, rr) + ["2I", "1P", "1I", "0M", "3I"] + filler(5, rr))
    add("two_ins_at_the_end", 50, filler(25, rr) + ["1I", "2P", "2I"])
    add("two_ins_at_the_start", 7, ["1I", "1P", "2I"] + filler(25, rr))
    add("long_runs", 10, filler(9, rr) + ["300M", "100D", "17M", "64D", "8M", "1I", "8M", "1D", "16M"] + filler(9, rr))
    for n_ops in (17, 255, 256, 257, 511, 512, 513, 1025):
        add("ops_%d" % n_ops, 3, filler(n_ops, random.Random(n_ops)))
    # one tile (64 ops) consumes more query bases than k_long_expand copies into LDS (8192): the fetches go to memory
    add("runs_longer_than_the_copy", 5, ["700M", "1I", "650M", "2D"] * 7 + ["5M"] + filler(9, rr), Lc=12000)
    add("tile_of_insertions", 20, ["5M"] + ["1I", "1P"] * 300 + ["9M"] + filler(7, rr))
    add("clips_both_ends", 300, ["200S"] + filler(41, rr) + ["150S"])
    add("leading_clip_over_the_contig_start", 30, ["100S"] + filler(41, rr))
    ops = filler(35, rr)
    foot = sum(int(o[:-1]) for o in ops if o[-1] in "MD")
    add("ends_on_the_last_site_then_ins", L - foot, ops + ["3I"])
    add("trailing_clip_over_the_contig_end", L - foot, ops + ["40S"])
    # bad bases: the reference raises KeyError (a base outside A,C,G,T,N inside an aligned or clipped segment)
    ops = filler(41, rr)
    s = rseq(qlen(ops), "ACGT")
    add("bad_base_aligned", 60, ops, KeyError, s[:70] + "R" + s[71:])
    ops = ["30S"] + filler(41, rr)
    s = rseq(qlen(ops), "ACGT")
    add("bad_base_in_leading_clip", 200, ops, KeyError, s[:12] + "Y" + s[13:])
    ops = filler(41, rr) + ["30S"]
    s = rseq(qlen(ops), "ACGT")
    add("bad_base_in_trailing_clip", 200, ops, KeyError, s[:-5] + "K" + s[-4:])
    ops = filler(41, rr)
    s = rseq(qlen(ops), "ACGT")
    qi = 0
    for o in ops:                       # a bad base inside an INSERTION is kept verbatim by the reference: no error
        if o[-1] == "I":
            break
        if o[-1] in "M":
            qi += int(o[:-1])
    add("odd_base_inside_an_insertion_is_fine", 60, ops, None, s[:qi] + "N" + s[qi + 1:])
    # CIGARs of more than 512 words with something awkward around word 512 (round 5 expanded long reads by 512-word segments; that
    # design was measured slower and dropped, the cases stay as long-CIGAR coverage): insertions on one site on both sides of the
    # mark, an insertion in front of the run behind it, deletions across it, the trailing clip right behind it
    add("ins_pair_across_a_segment", 3, filler(511, random.Random(21)) + ["2I", "1P", "1I"] + filler(41, random.Random(22)))
    add("segment_ends_on_ins", 3, filler(511, random.Random(23)) + ["3I"] + filler(40, random.Random(24)))
    add("deletions_across_a_segment", 3, filler(510, random.Random(25)) + ["5D", "4D", "3M"] + filler(41, random.Random(26)))
    add("trailing_clip_opens_a_segment", 3, filler(512, random.Random(27)) + ["25S"])
    add("ops_1536", 3, filler(1536, random.Random(1536)), Lc=9000)
    add("ops_2049", 3, filler(2049, random.Random(2049)), Lc=12000)
    return cases
