"""Known-answer tests that pin the oracle (oracle/kmr_oracle.cpp) to the reference.

Vectors are re-typed from the reference's own unit tests and self-test driver:
  test/TwoBitSequenceTest.cpp:119-170 (shiftLeft), :172-252 (reverseComplement),
  :285-350 (markups), test/KmerTest.cpp:253-306 (window extraction / last-byte mask),
  src/lookup3.h:981-1000 (driver5), plus the KATs SURVEY.md 8(c) generated from the
  reference's lookup3.h + TwoBitSequence.cpp.  When oracle/_ref/libref_lookup3.so (the
  reference's own lookup3.h compiled from /root/reference) is present the hash is also
  cross-checked on random inputs.
"""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import (REF_SO, default_config, oracle_lib, oracle_weighted_kmers, _ptr)


def compress(s):
    lib = oracle_lib()
    out = np.zeros((len(s) + 3) // 4 + 1, dtype=np.uint8)
    mpos = np.zeros(max(1, len(s)), dtype=np.uint32)
    mch = C.create_string_buffer(max(1, len(s)))
    n = lib.orc_compress_sequence(s.encode(), len(s), _ptr(out, C.c_uint8), _ptr(mpos, C.c_uint32), mch, len(s))
    return out[:(len(s) + 3) // 4], [(mch.raw[i:i + 1].decode(), int(mpos[i])) for i in range(n)]


def uncompress(b, n):
    return "".join("ACGT"[(int(b[i >> 2]) >> (6 - 2 * (i & 3))) & 3] for i in range(n))


def revcomp_str(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


LEFT_SHIFT = [
    ("ACGT", "ACGT", 0), ("ACGT", "CGTA", 1), ("ACGT", "GTAA", 2), ("ACGT", "TAAA", 3),
    ("AAAAAAAA", "AAAAAAAA", 0), ("AAAAAAAA", "AAAAAAAA", 1), ("AAAAAAAA", "AAAAAAAA", 2), ("AAAAAAAA", "AAAAAAAA", 3),
    ("CCCCCCCC", "CCCCCCCA", 1), ("CCCCCCCC", "CCCCCCAA", 2), ("CCCCCCCC", "CCCCCAAA", 3),
    ("CCCCCCC", "CCCCCCA", 1), ("CCCCCCC", "CCCCCAA", 2), ("CCCCCCC", "CCCCAAA", 3),
    ("CCCCCC", "CCCCCA", 1), ("CCCCCC", "CCCCAA", 2), ("CCCCCC", "CCCAAA", 3),
    ("CCCCC", "CCCCA", 1), ("CCCCC", "CCCAA", 2), ("CCCCC", "CCAAA", 3),
    ("ACGTACGT", "CGTACGTA", 1), ("ACGTACGT", "GTACGTAA", 2), ("ACGTACGT", "TACGTAAA", 3),
    ("TACGTACGT", "ACGTACGTA", 1), ("TACGTACGT", "CGTACGTAA", 2), ("TACGTACGT", "GTACGTAAA", 3),
    ("GTACGTACGT", "TACGTACGTA", 1), ("GTACGTACGT", "ACGTACGTAA", 2), ("GTACGTACGT", "CGTACGTAAA", 3),
    ("CGTACGTACGT", "GTACGTACGTA", 1), ("CGTACGTACGT", "TACGTACGTAA", 2), ("CGTACGTACGT", "ACGTACGTAAA", 3),
]


@pytest.mark.parametrize("src,dst,shift", LEFT_SHIFT)
def test_shift_left(src, dst, shift):
    lib = oracle_lib()
    b, _ = compress(src)
    out = np.zeros_like(b)
    lib.orc_shift_left(_ptr(b, C.c_uint8), _ptr(out, C.c_uint8), len(b), shift, 0)
    assert uncompress(out, len(src)) == dst


REV_COMP = [("A", "T"), ("C", "G"), ("AA", "TT"), ("CC", "GG"), ("AAA", "TTT"), ("CCC", "GGG"),
            ("AAAA", "TTTT"), ("CCCCC", "GGGGG"), ("AAAAAA", "TTTTTT"), ("CCCCCCC", "GGGGGGG"),
            ("AAAAAAAAA", "TTTTTTTTT"), ("CCCCCCCCCC", "GGGGGGGGGG"), ("AAAAAAAAAAA", "TTTTTTTTTTT"),
            ("CCCCCCCCCCCC", "GGGGGGGGGGGG"), ("ACGT", "ACGT"), ("AGCT", "AGCT"), ("GCTA", "TAGC"),
            ("TTAA", "TTAA"), ("ATAT", "ATAT"), ("TACC", "GGTA"), ("AC", "GT"), ("AG", "CT"), ("AT", "AT"),
            ("CA", "TG"), ("CG", "CG"), ("CT", "AG"), ("GA", "TC"), ("GC", "GC"), ("GT", "AC"), ("TA", "TA"),
            ("TC", "GA"), ("TG", "CA"), ("TT", "AA"),
            ("ACGTCGTAGTACTACGA", "TCGTAGTACTACGACGT")]


@pytest.mark.parametrize("fwd,rev", REV_COMP)
def test_reverse_complement(fwd, rev):
    lib = oracle_lib()
    b, _ = compress(fwd)
    out = np.zeros_like(b)
    lib.orc_reverse_complement(_ptr(b, C.c_uint8), _ptr(out, C.c_uint8), len(fwd))
    assert uncompress(out, len(fwd)) == rev
    back = np.zeros_like(b)
    lib.orc_reverse_complement(_ptr(out, C.c_uint8), _ptr(back, C.c_uint8), len(fwd))
    assert uncompress(back, len(fwd)) == fwd
    # pad bits of the reverse complement are zero (re-left-justified)
    rb, _ = compress(rev)
    assert out.tobytes() == rb.tobytes()


def test_reverse_complement_random():
    rng = np.random.default_rng(3)
    lib = oracle_lib()
    for n in list(range(1, 70)) + [95, 128]:
        s = "".join("ACGT"[i] for i in rng.integers(0, 4, n))
        b, _ = compress(s)
        out = np.zeros_like(b)
        lib.orc_reverse_complement(_ptr(b, C.c_uint8), _ptr(out, C.c_uint8), n)
        assert out.tobytes() == compress(revcomp_str(s))[0].tobytes()


def test_compress_and_markups():
    assert compress("ACGTCGTAGTACTACGA")[0].tobytes().hex() == "1b6cb1c600"
    assert compress("acgt")[0].tobytes().hex() == "1b"
    # test/TwoBitSequenceTest.cpp:285-350: N at 0,5,10,15,23 -> markups, packed as A
    n5 = "NACGTNACGTNACGTNACGTACGNAC"
    b, m = compress(n5)
    assert m == [("N", 0), ("N", 5), ("N", 10), ("N", 15), ("N", 23)]
    assert uncompress(b, len(n5)) == n5.replace("N", "A")
    # '.' is recorded as N (src/TwoBitSequence.cpp:255-257)
    assert compress("AC.T")[1] == [("N", 2)]
    assert compress("ACXT")[1] == [("X", 2)]


def test_lookup3_driver5_vectors():
    """src/lookup3.h:981-1000"""
    lib = oracle_lib()

    def h2(key, c, b):
        pc, pb = C.c_uint32(c), C.c_uint32(b)
        lib.orc_hashlittle2(key, len(key), C.byref(pc), C.byref(pb))
        return pc.value, pb.value
    assert h2(b"", 0, 0) == (0xdeadbeef, 0xdeadbeef)
    assert h2(b"", 0, 0xdeadbeef) == (0xbd5b7dde, 0xdeadbeef)
    assert h2(b"", 0xdeadbeef, 0xdeadbeef) == (0x9c093ccd, 0xbd5b7dde)
    s = b"Four score and seven years ago"
    assert h2(s, 0, 0) == (0x17770551, 0xce7226e6)
    assert h2(s, 0, 1) == (0xe3607cae, 0xbd371de4)
    assert h2(s, 1, 0) == (0xcd628161, 0x6cbea4b3)


def test_get_hash_kats():
    """generated from the reference's own lookup3.h + TwoBitSequence.cpp."""
    lib = oracle_lib()
    assert lib.orc_hash(bytes([0x1b] * 8), 8) == 0x840005f7669b0c06
    assert lib.orc_hash(bytes(range(6)), 6) == 0xc0a3c819da6148c3
    assert lib.orc_hash(bytes(range(13)), 13) == 0xd7f4f68530825494
    assert lib.orc_hash(bytes([0x1b]), 1) == 0x6641f64a79d036e7


KMER_KATS = [
    # k, fasta, fwd hex, rc hex, fwd_is_least, hash, bucket(mask 0xfff), owner of 8
    (21, "AAAAAAAGTTTGAATTATGGC", "0002fe0f3a40", "94c3d01fffc0", True, 0x549138987bbcf1f3, 499, 3),
    (31, "AGCATCAGTGACGACATTAGAAATATCCTTT", "24d2e184f20335fc", "028cfdc3b6d1e39c", False, 0xa511b942241edb49, 2889, 4),
    (51, "TCAACAGAAGGAGTCTACTGCTCGCGTTGCGTCTATTATGGAAAACACCAA", "d04828b71e766f9b73ce801140",
     "faeff530c86419892c875f7be0", True, 0x84841c2fb546924d, 589, 5),
]


@pytest.mark.parametrize("k,fasta,fwd,rc,least,h,bucket,owner", KMER_KATS)
def test_kmer_kats(k, fasta, fwd, rc, least, h, bucket, owner):
    lib = oracle_lib()
    b, _ = compress(fasta)
    assert b.tobytes().hex() == fwd
    out = np.zeros_like(b)
    lib.orc_reverse_complement(_ptr(b, C.c_uint8), _ptr(out, C.c_uint8), k)
    assert out.tobytes().hex() == rc
    can = np.zeros_like(b)
    assert lib.orc_least_complement(_ptr(b, C.c_uint8), k, _ptr(can, C.c_uint8)) == (1 if least else 0)
    assert can.tobytes().hex() == (fwd if least else rc)
    hv = lib.orc_hash(can.tobytes(), len(can))
    assert hv == h
    assert lib.orc_bucket_idx(hv, 4096) == bucket
    assert lib.orc_distributed_thread_id(hv, 8) == owner
    # the reverse-complement string gives the same canonical k-mer and hash
    b2, _ = compress(revcomp_str(fasta))
    can2 = np.zeros_like(b2)
    lib.orc_least_complement(_ptr(b2, C.c_uint8), k, _ptr(can2, C.c_uint8))
    assert can2.tobytes() == can.tobytes()


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (no reference checkout)")
def test_hash_against_reference_lookup3():
    ref = C.CDLL(REF_SO)
    ref.ref_get_hash.restype = C.c_uint64
    ref.ref_get_hash.argtypes = [C.c_char_p, C.c_uint64]
    lib = oracle_lib()
    rng = np.random.default_rng(11)
    for n in range(1, 41):
        for _ in range(50):
            key = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
            buf = C.create_string_buffer(key, n)   # 8-byte aligned copy -> the reference's aligned path
            assert lib.orc_hash(key, n) == ref.ref_get_hash(buf, n)
            # misaligned start exercises the reference's byte-wise path
            buf2 = C.create_string_buffer(b"x" + key, n + 1)
            assert lib.orc_hash(key, n) == ref.ref_get_hash(C.cast(C.addressof(buf2) + 1, C.c_char_p), n)


def test_window_extraction_and_mask():
    """KmerArrayPair::build for k=1..12 (test/KmerTest.cpp:253-306): window i is the
    k bases starting at i, left-justified, pad bits zero."""
    seq = "ACGTCGTAGTACTACGATTTACGGGCAT"
    for k in range(1, 13):
        cfg = default_config(k, min_quality_score=0)
        keys, w, ext = oracle_weighted_kmers(cfg, seq.encode(), b"I" * len(seq))
        assert keys.shape[0] == len(seq) - k + 1
        for i in range(keys.shape[0]):
            sub = seq[i:i + k]
            f, r = compress(sub)[0].tobytes(), compress(revcomp_str(sub))[0].tobytes()
            assert keys[i].tobytes() == min(f, r)
            assert (w[i] > 0) == (f <= r)


def test_quality_table():
    lib = oracle_lib()
    P = np.zeros(256)
    lib.orc_quality_table(3, 33, _ptr(P, C.c_double))
    assert P[33 + 2] == 0.0 and P[33 + 3] == 1.0 - 10.0 ** (-0.3)
    assert P[33 + 40] == 1.0 - 10.0 ** (-4.0)
    assert P[102] == 1.0 - 10.0 ** ((33 - 102) / 10.0) and P[103] == 1.0 and P[255] == 1.0
    P64 = np.zeros(256)
    lib.orc_quality_table(3, 64, _ptr(P64, C.c_double))
    assert np.array_equal(P64[31:], P[:256 - 31])   # same function of the Phred value


def test_weight_recurrence_and_markups():
    """KmerReadUtils.h:201-219: fp64 running product, recompute after a zero, N -> weight 0."""
    k = 5
    cfg = default_config(k)
    seq = b"ACGTACGTNACGTACGTACG"
    qual = b"IIIIIIII#IIII5IIIII!"
    keys, w, ext = oracle_weighted_kmers(cfg, seq, qual)
    P = np.zeros(256)
    oracle_lib().orc_quality_table(3, 33, _ptr(P, C.c_double))
    for i in range(len(seq) - k + 1):
        window_has_n = any(seq[i + j] == ord("N") for j in range(k))
        prod = 1.0
        for j in range(k):
            prod *= P[qual[i + j]]
        if window_has_n:
            assert w[i] == 0.0
        else:
            assert abs(abs(w[i]) - prod) <= 1e-6 * max(prod, 1e-30)
    assert w[-1] == 0.0   # '!' is below min-quality-score


def test_bucket_sizing():
    """KmerSpectrum ctor sizing (src/KmerSpectrum.h:414-416, src/Kmer.h:2837,2224-2229):
    config 2 (1.2e9 raw k-mers) -> weak 2^21, singleton 2^24 buckets."""
    lib = oracle_lib()
    cfg = default_config(31, estimated_raw_kmers=1200000000)
    w, s = C.c_uint64(), C.c_uint64()
    lib.orc_derive_buckets(C.byref(cfg), C.byref(w), C.byref(s))
    assert (w.value, s.value) == (1 << 21, 1 << 24)
    assert lib.orc_min_power_of_2(0) == 1 and lib.orc_min_power_of_2(5) == 8 and lib.orc_min_power_of_2(64) == 64
    cfg = default_config(31, estimated_raw_kmers=1 << 40)
    lib.orc_derive_buckets(C.byref(cfg), C.byref(w), C.byref(s))
    assert s.value == 1 << 26   # MAX_KMER_MAP_BUCKETS
