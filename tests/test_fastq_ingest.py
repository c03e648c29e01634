"""FASTQ ingest.

CPU: the oracle's restatement of FastqStreamParser / nextRead / validateFastqStart against the reference's own
fixtures -- test/1000.fastq (Phred-64) and its Phred-33 twin test/1000.std.fastq must give the same reads once the
quality base has been detected.  GPU: kmr_ingest_fastq against the oracle, byte for byte, and end to end into the
spectrum build."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, KMR_MAP_WEAK, OracleSpectrum, default_config, oracle_parse_fastq, read_fastq


def _text(name):
    return open(os.path.join(GOLDEN, name), "rb").read()


def synth_fastq(seed=3, n=300, casava=True, lower=True, blank=True, trailing_newline=True):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        L = int(rng.integers(1, 260))
        seq = rng.choice(list(b"ACGTN"), size=L, p=[.24, .24, .24, .24, .04]).astype(np.uint8).tobytes()
        if lower and i % 7 == 0:
            seq = seq.lower()
        q = (rng.integers(2, 41, size=L) + 33).astype(np.uint8).tobytes()
        name = b"r%d" % i
        if casava:
            kind = i % 5
            if kind == 0:
                name += b" 1:N:0:ACGT"
            elif kind == 1:
                name += b" 2:Y:0:ACGT"        # failed filter: dropped when comments are stored
            elif kind == 2:
                name += b"/1 1:Y:0:ACGT"      # already has a /1 suffix: the filter rule does not apply
            elif kind == 3:
                name += b"\tsome comment"
        out.append(b"@" + name + b"\n" + seq + b"\n+" + (name if i % 3 == 0 else b"") + b"\n" + q + b"\n")
        if blank and i % 11 == 0:
            out.append(b"\n")
    text = b"".join(out)
    if not trailing_newline:
        text = text.rstrip(b"\n")
    return text


# ------------------------------------------------------------------ oracle (CPU)
def test_oracle_parser_reads_the_reference_fixture():
    rb, base = oracle_parse_fastq(_text("1000.std.fastq"), 33, 33)
    ref = read_fastq(os.path.join(GOLDEN, "1000.std.fastq"))
    assert rb.n == ref.n == 1000 and base == 33
    assert np.array_equal(rb.bases, ref.bases) and np.array_equal(rb.quals, ref.quals) and np.array_equal(rb.offsets, ref.offsets)
    assert [n.split()[0] for n in rb.names] == [n.split()[0] for n in ref.names]


def test_oracle_detects_phred64_and_matches_the_std_twin():
    """validateFastqStart flips the input base to 64 and rescales everything: 1000.fastq read with the default
    --fastq-base-quality 33 equals its Phred-33 twin (test/1000.std.fastq is how the reference's tests get base 33)"""
    a, base_a = oracle_parse_fastq(_text("1000.fastq"), 33, 33)
    b, base_b = oracle_parse_fastq(_text("1000.std.fastq"), 33, 33)
    c, base_c = oracle_parse_fastq(_text("1000.fastq"), 33, 64)        # --fastq-base-quality 64, as test/runFilterTests.sh does
    assert (base_a, base_b, base_c) == (64, 33, 64)
    assert np.array_equal(a.quals, b.quals) and np.array_equal(a.bases, b.bases)
    assert np.array_equal(c.quals, b.quals)


def test_oracle_parser_rules():
    t = synth_fastq()
    rb, _ = oracle_parse_fastq(t, 33, 33, store_comment=True)
    dropped = sum(1 for i in range(300) if i % 5 == 1)
    assert rb.n == 300 - dropped
    assert not any(n.startswith(b"r1 ") or n.startswith(b"r6 ") for n in rb.names)
    assert rb.bases.tobytes() == rb.bases.tobytes().upper()
    # comments not stored: the reference's index moves with the rewritten name and looks two chars further on
    rb2, _ = oracle_parse_fastq(t, 33, 33, store_comment=False)
    assert rb2.n == 300
    # malformed input: the reference throws
    assert oracle_parse_fastq(b"@r\nACGT\nIIII\n+\n", 33, 33) is None             # '+' missing
    assert oracle_parse_fastq(b"@r\nACGT\n+\nIII\n", 33, 33) is None              # lengths differ
    assert oracle_parse_fastq(b"@r\n\n+\n\n", 33, 33) is None                     # no bases
    # no trailing newline, blank lines between records
    rb3, _ = oracle_parse_fastq(b"@a\nAC\n+\nII\n\n\n@b\nGT\n+\nII", 33, 33)
    assert rb3.n == 2 and rb3.names == [b"a", b"b"]


# ------------------------------------------------------------------ device (GPU)
def _spectrum(k=21, start=33, **kw):
    import kmernator_amd as ka
    return ka.KmerSpectrum(ka.default_config(k, estimated_raw_kmers=200000, device=0, fastq_start_char=start, **kw))


def _same(rs, rb, base):
    b, q, o, names = rs.arrays()
    assert rs.n == rb.n
    assert np.array_equal(o, rb.offsets) and np.array_equal(b, rb.bases) and np.array_equal(q, rb.quals)
    assert names == rb.names
    assert rs.input_quality_base == base


@pytest.mark.gpu
@pytest.mark.parametrize("fq,start,inb", [("1000.fastq", 33, 33), ("1000.fastq", 33, 64), ("1000.std.fastq", 33, 33), ("1000.fastq", 64, 64),
                                           ("1000.std.fastq", 64, 64), ("10.fastq", 33, 0)])
def test_ingest_reference_fixtures(fq, start, inb):
    import kmernator_amd as ka
    sp = _spectrum(start=start)
    text = _text(fq)
    rb, base = oracle_parse_fastq(text, start, inb or start)
    _same(ka.ReadSet(sp, text, inb), rb, base)


@pytest.mark.gpu
@pytest.mark.parametrize("store_comment", [True, False])
@pytest.mark.parametrize("trailing", [True, False])
def test_ingest_synthetic_text(store_comment, trailing):
    import kmernator_amd as ka
    sp = _spectrum()
    text = synth_fastq(seed=11, n=5000, trailing_newline=trailing)
    rb, base = oracle_parse_fastq(text, 33, 33, store_comment)
    rs = ka.ReadSet(sp, text, 33, store_comment)
    _same(rs, rb, base)
    assert rs.filtered == 5000 - rb.n


@pytest.mark.gpu
def test_ingest_empty_and_malformed():
    import kmernator_amd as ka
    sp = _spectrum()
    assert ka.ReadSet(sp, b"").n == 0
    assert ka.ReadSet(sp, b"\n\n\n").n == 0
    for bad in (b"@r\nACGT\nIIII\n+\n", b"@r\nACGT\n+\nIII\n", b"@r\nACGT\n+\n", b"r\nACGT\n+\nIIII\n", b"@r\nACGT\n\n+\nIIII\n",
                b"@ x\nACGT\n+\nIIII\n"):
        with pytest.raises(ka.KmerSpectrumError):
            ka.ReadSet(sp, bad)
        assert oracle_parse_fastq(bad, 33, 33) is None or oracle_parse_fastq(bad, 33, 33)[0].n == 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2])
def test_ingest_then_build_equals_host_parsed_build(mode):
    """FASTQ text -> device ReadSet -> spectrum, against the oracle fed by the oracle's own parser (k=31, FilterReads settings)"""
    import kmernator_amd as ka
    text = _text("1000.fastq")
    rb, _ = oracle_parse_fastq(text, 33, 33)
    cfg = default_config(31, estimated_raw_kmers=1000 * 46)
    o = OracleSpectrum(cfg)
    o.add_reads(rb)
    o.finalize(2)
    sp = ka.KmerSpectrum(ka.default_config(31, estimated_raw_kmers=1000 * 46, device=0, build_mode=mode))
    rs = ka.ReadSet(sp, text)
    sp.buildKmerSpectrumFromReadSet(rs)
    rs.close()
    sp.finalize(2)
    assert sp.stats() == o.stats()
    ko, cnt, _, _, _ = o.entries()
    assert np.array_equal(sp.getCount(ko), cnt)
    nb = o.num_buckets(KMR_MAP_WEAK)
    assert np.array_equal(sp.image(KMR_MAP_WEAK)[:16 + 8 * nb], o.image(KMR_MAP_WEAK)[:16 + 8 * nb])


@pytest.mark.gpu
def test_filterreads_from_fastq_text_on_the_device():
    """FASTQ text -> device ReadSet (quality base detected: 1000.fastq is Phred-64) -> spectrum -> scoreAndTrimReads,
    the reads never staged by the host: the MedianScore / Trim labels of the 949 reads without AFTrim in the reference's
    FilterReads golden test/1000-Filtered.fastq."""
    import kmernator_amd as ka
    gold = read_fastq(os.path.join(GOLDEN, "1000-Filtered.fastq"))
    sp = ka.KmerSpectrum(ka.default_config(31, estimated_raw_kmers=46000, device=0))
    rs = ka.ReadSet(sp, _text("1000.fastq"))
    assert rs.input_quality_base == 64 and rs.n == 1000
    sp.buildKmerSpectrumFromReadSet(rs)
    sp.finalize(2)
    to, tl, sc, wt = sp.scoreAndTrimReadSet(rs, 2, "MEDIAN")
    checked = 0
    for i in range(rs.n):
        if b"AFTrim" in gold.names[i]:
            continue
        label = b""
        if wt[i]:
            label += b"Trim:%d+%d " % (to[i], tl[i])
        label += b"MedianScore:%d" % int(sc[i] + 0.5)
        assert label == gold.names[i].split(b" ", 1)[1], (i, label, gold.names[i])
        checked += 1
    assert checked == 949


@pytest.mark.gpu
def test_twobit_pack_of_a_batch_matches_compress_sequence():
    """f2: TwoBitSequence::compressSequence over a device-resident batch (packed bytes + markup list per read) against the
    oracle restatement (test/TwoBitSequenceTest.cpp's cases are in tests/test_oracle_kat.py), incl. '.', 'X', IUPAC codes, lower
    case, reads of 0-3 bases and lengths that are not multiples of 4"""
    import ctypes as C
    import kmernator_amd as ka
    from helpers import oracle_lib, ReadBatch
    rng = np.random.default_rng(23)
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTacgtN.XRY", dtype=np.uint8)
    seqs = [bytes(rng.choice(alphabet, size=int(L)).tobytes()) for L in list(rng.integers(1, 300, 4000)) + [1, 2, 3, 4, 5, 8]]
    rb = ReadBatch(seqs, [b"I" * len(s) for s in seqs])
    sp = ka.KmerSpectrum(ka.default_config(21, estimated_raw_kmers=100000, device=0))
    rs = ka.ReadSet.from_arrays(sp, rb.bases, rb.quals, rb.offsets)
    tw, to, mp, mc, mo = rs.twobit()
    lib = oracle_lib()
    assert int(to[-1]) == tw.size == sum((len(s) + 3) // 4 for s in seqs)
    for i, s in enumerate(seqs):
        out = np.zeros((len(s) + 3) // 4 + 1, dtype=np.uint8)
        pos = np.zeros(len(s) + 1, dtype=np.uint32)
        ch = C.create_string_buffer(len(s) + 1)
        nm = lib.orc_compress_sequence(s, len(s), out.ctypes.data_as(C.POINTER(C.c_uint8)), pos.ctypes.data_as(C.POINTER(C.c_uint32)), ch, len(s) + 1)
        assert np.array_equal(tw[int(to[i]):int(to[i + 1])], out[:(len(s) + 3) // 4]), i
        assert int(mo[i + 1] - mo[i]) == nm, i
        assert np.array_equal(mp[int(mo[i]):int(mo[i + 1])], pos[:nm]), i
        assert mc[int(mo[i]):int(mo[i + 1])].tobytes() == ch.raw[:nm], i
    assert int(mo[-1]) > 1000
