"""f4: the artifact filter (FilterKnownOddities, src/FilterKnownOddities.h) -- oracle against the reference's golden
FilterReads output (CPU), product against the oracle (GPU, through the C-ABI)."""
import os
import re

import numpy as np
import pytest

from helpers import (GOLDEN, OracleArtifactFilter, OracleSpectrum, ReadBatch, apply_artifact_result, artifact_config, default_config,
                     oracle_weighted_kmers, read_fastq, synth_reads)
from refsemantics import median_trim_label


def fasta(name):
    return open(os.path.join(GOLDEN, name), "rb").read()


def golden_cfg(start):
    # test/runFilterTests.sh:24,63: --artifact-edit-distance 1 --mask-simple-repeats 0 --min-read-length 25 (min-quality-score 3)
    return artifact_config(edit_distance=1, fastq_start_char=start, min_read_length=25.0)


def test_filter_set_of_the_reference_table():
    """prepareMaps: every 24-mer of the circularised sequences, then one round of substitutions built in"""
    f0 = OracleArtifactFilter(artifact_config(edit_distance=0), fasta("artifact_sequences.fa"))
    nseq, n0, left = f0.info()
    assert nseq == 25 and left == 0          # 24 sequences + the empty read 0
    keys0, vals0 = f0.entries()
    assert vals0.min() >= 1 and vals0.max() <= 24
    # Homopolymer-A (sequence 3): a single canonical key, all A = 0
    assert keys0[0] == 0 and vals0[0] == 3
    f1 = OracleArtifactFilter(artifact_config(edit_distance=1), fasta("artifact_sequences.fa"))
    _, n1, left1 = f1.info()
    assert left1 == 0 and n0 < n1 <= n0 * 73
    keys1, vals1 = f1.entries()
    pos = np.searchsorted(keys1, keys0)
    assert np.array_equal(keys1[pos], keys0) and np.array_equal(vals1[pos], vals0)      # exact k-mers keep their sequence
    # default settings (distance 2, build while < 750 000 keys): both rounds are built in; a third edit stays for query time
    f2 = OracleArtifactFilter(artifact_config(), fasta("artifact_sequences.fa"))
    _, n2, left2 = f2.info()
    assert left2 == 0 and n2 > 750000
    assert OracleArtifactFilter(artifact_config(edit_distance=3), fasta("artifact_sequences.fa")).info() == (25, n2, 1)
    # build_edits = 0: nothing built in
    f3 = OracleArtifactFilter(artifact_config(build_edits=0), fasta("artifact_sequences.fa"))
    assert f3.info() == (25, n0, 2)


@pytest.mark.parametrize("fq,start", [("1000.fastq", 64), ("1000.std.fastq", 33)])
def test_golden_aftrim_labels(fq, start):
    """test/1000-Filtered.fastq: the 51 reads the reference's artifact filter trimmed, and only those"""
    rb = read_fastq(os.path.join(GOLDEN, fq))
    gold = read_fastq(os.path.join(GOLDEN, "1000-Filtered.fastq"))
    f = OracleArtifactFilter(golden_cfg(start), fasta("artifact_sequences.fa"))
    res = f.apply(rb)
    n_trim = 0
    for i in range(rb.n):
        m = re.search(rb"AFTrim:(\d+)\+(\d+)", gold.names[i])
        if m:
            n_trim += 1
            assert res["action"][i] == 1, (i, gold.names[i])
            assert (int(res["min_pass"][i]), int(res["max_pass"][i] - res["min_pass"][i])) == (int(m.group(1)), int(m.group(2))), (i, gold.names[i])
            assert res["value"][i] == 25           # sequences.getSize(): quality trim only
        else:
            assert res["action"][i] == 0 and res["value"][i] == 0, (i, gold.names[i])
    assert n_trim == 51
    assert gold.n == rb.n and not res["remnant_len"].any()


def test_golden_filterreads_end_to_end():
    """artifact filter -> spectrum of the filtered reads -> scoreAndTrimReads: all 1000 labels and sequences of
    test/1000-Filtered.fastq (runFilterTests.sh:63)"""
    k = 31
    rb = read_fastq(os.path.join(GOLDEN, "1000.fastq"))
    gold = read_fastq(os.path.join(GOLDEN, "1000-Filtered.fastq"))
    f = OracleArtifactFilter(golden_cfg(64), fasta("artifact_sequences.fa"))
    res = f.apply(rb)
    fr = apply_artifact_result(rb, res)
    cfg = default_config(k, fastq_start_char=64, estimated_raw_kmers=(76 - k + 1) * 1000)
    s = OracleSpectrum(cfg)
    s.add_reads(fr)
    s.finalize(2)
    for i in range(rb.n):
        keys, w, ext = oracle_weighted_kmers(cfg, fr.seq(i), fr.qual(i))
        label = median_trim_label(s.lookup(keys), k)
        if res["action"][i] == 1:
            label = b"AFTrim:%d+%d " % (res["min_pass"][i], res["max_pass"][i] - res["min_pass"][i]) + label
        assert label == gold.names[i].split(b" ", 1)[1].replace(b"\t", b" "), (i, gold.names[i], label)
        m = re.search(rb"Trim:(\d+)\+(\d+) Median", gold.names[i])
        seq = fr.seq(i)
        if m:
            seq = seq[int(m.group(1)):int(m.group(1)) + int(m.group(2))]
        assert seq == gold.seq(i), (i, gold.names[i])


# ---------------------------------------------------------------- product (GPU) against the oracle, through the C-ABI

def _fastq_text(rb, names=None):
    out = []
    for i in range(rb.n):
        out.append(b"@" + (names[i] if names else b"r%d" % i) + b"\n" + rb.seq(i) + b"\n+\n" + rb.qual(i) + b"\n")
    return b"".join(out)


def _device_filter(cfg_kw, fasta_text, k=31):
    import kmernator_amd as ka
    sp = ka.KmerSpectrum(ka.default_config(k, estimated_raw_kmers=100000, device=0))
    return sp, ka.FilterKnownOddities(sp, fasta_text, **cfg_kw)


def _all_tables():
    """artifact table + simple repeats + PhiX in the order the reference appends them (:213-229), with the class ranges"""
    a, s, p = fasta("artifact_sequences.fa"), fasta("simple_repeats.fa"), fasta("phix.fa")
    na, ns = a.count(b">"), s.count(b">")
    return a + s + p, dict(simple_repeat_begin=1 + na, simple_repeat_end=1 + na + ns, phix_idx=1 + na + ns)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(edit_distance=0), dict(edit_distance=1), dict(), dict(edit_distance=3), dict(build_edits=0), dict(match_length=20, edit_distance=1)])
def test_filter_set_matches_oracle(kw):
    """prepareMaps on the device: same keys, same sequence index per key (first writer in the map's iteration order)"""
    o = OracleArtifactFilter(artifact_config(**kw), fasta("artifact_sequences.fa"))
    sp, f = _device_filter(kw, fasta("artifact_sequences.fa"))
    assert (f.n_sequences, f.n_filter_kmers, f.remaining_edits) == o.info()
    ko, vo = o.entries()
    kd, vd = f.entries()
    assert np.array_equal(kd, ko) and np.array_equal(vd, vo)


@pytest.mark.gpu
def test_filter_set_with_repeats_and_phix():
    text, cls = _all_tables()
    kw = dict(edit_distance=2, **cls)
    o = OracleArtifactFilter(artifact_config(**kw), text)
    sp, f = _device_filter(kw, text)
    assert (f.n_sequences, f.n_filter_kmers, f.remaining_edits) == o.info()
    assert f.n_filter_kmers > 100000
    ko, vo = o.entries()
    kd, vd = f.entries()
    assert np.array_equal(kd, ko) and np.array_equal(vd, vo)


def _check_apply(f, o, rs, rb, mate=None):
    want = o.apply(rb, mate)
    got, frs = f.applyFilter(rs, mate)
    for key in ("value", "min_pass", "max_pass", "action", "remnant_off", "remnant_len"):
        bad = np.nonzero(got[key] != want[key])[0]
        assert bad.size == 0, (key, bad[:5], got[key][bad[:5]], want[key][bad[:5]])
    fr = apply_artifact_result(rb, want)
    b, q, off, names = frs.arrays()
    assert frs.n == fr.n and np.array_equal(off, fr.offsets)
    assert np.array_equal(b, fr.bases) and np.array_equal(q, fr.quals)
    return want, frs, names


@pytest.mark.gpu
def test_golden_filterreads_end_to_end_on_the_device():
    """FASTQ text -> device reads -> artifact filter -> spectrum of the filtered reads -> scoreAndTrimReads, everything on
    the device: every label of all 1000 reads of test/1000-Filtered.fastq incl. the 51 AFTrim ones"""
    import kmernator_amd as ka
    gold = read_fastq(os.path.join(GOLDEN, "1000-Filtered.fastq"))
    sp = ka.KmerSpectrum(ka.default_config(31, estimated_raw_kmers=46000, device=0))
    rs = ka.ReadSet(sp, fasta("1000.fastq"))
    assert rs.input_quality_base == 64
    f = ka.FilterKnownOddities(sp, fasta("artifact_sequences.fa"), edit_distance=1, min_read_length=25.0)
    res, frs = f.applyFilter(rs)
    assert frs.n == 1000 and int((res["action"] == 1).sum()) == 51 and not (res["action"] == 2).any()
    sp.buildKmerSpectrumFromReadSet(frs)
    sp.finalize(2)
    to, tl, sc, wt = sp.scoreAndTrimReadSet(frs, 2, "MEDIAN")
    b, q, off, names = frs.arrays()
    for i in range(1000):
        label = b""
        if res["action"][i] == 1:
            label += b"AFTrim:%d+%d " % (res["min_pass"][i], res["max_pass"][i] - res["min_pass"][i])
        if wt[i]:
            label += b"Trim:%d+%d " % (to[i], tl[i])
        label += b"MedianScore:%d" % int(sc[i] + 0.5)
        assert names[i].split(b" ")[0] == gold.names[i].split(b" ")[0]
        assert label == gold.names[i].split(b" ", 1)[1].replace(b"\t", b" "), (i, label, gold.names[i])
        seq = bytes(b[int(off[i]):int(off[i + 1])])
        if wt[i]:
            seq = seq[int(to[i]):int(to[i]) + int(tl[i])]
        assert seq == gold.seq(i), (i, gold.names[i])


@pytest.mark.gpu
@pytest.mark.parametrize("gold_name,mrl,both,out_base", [
    ("1000-Filtered-0.85.fastq", 0.85, False, 64), ("1000-Filtered-0.85.std.fastq", 0.85, False, 33),
    ("1000-Filtered-readlength.fastq", 1.0, False, 64), ("1000-Filtered-readlength-both.fastq", 1.0, True, 64), ("1000-Filtered.fastq", 25.0, False, 64)])
@pytest.mark.parametrize("fq", ["1000.fastq", "1000.std.fastq"])
def test_filterreads_selection_goldens_on_the_device(fq, gold_name, mrl, both, out_base):
    """test/runFilterTests.sh:43-63, every golden the script checks, WHOLE FILE byte for byte: FASTQ text -> device reads (quality
    base detected) -> artifact filter with the run's --min-read-length -> spectrum of the filtered reads -> scoreAndTrimReads, all on
    the device through the C-ABI; what remains is FilterReads' host-side selection and printing (isPassingRead / isPassingPair /
    writePicks, src/ReadSelector.h:547-596,1242-1262) over kmr_score_read_batch's output (tests/refsemantics.py)."""
    import kmernator_amd as ka
    from refsemantics import filterreads_output
    gold = fasta(gold_name)
    sp = ka.KmerSpectrum(ka.default_config(31, estimated_raw_kmers=46000, device=0))
    rs = ka.ReadSet(sp, fasta(fq))
    in_base = rs.input_quality_base
    assert in_base == (33 if "std" in fq else 64)
    f = ka.FilterKnownOddities(sp, fasta("artifact_sequences.fa"), edit_distance=1, min_read_length=mrl)
    res, frs = f.applyFilter(rs)
    assert frs.n == 1000
    sp.buildKmerSpectrumFromReadSet(frs)
    sp.finalize(2)
    to, tl, sc, wt = sp.scoreAndTrimReadSet(frs, 2, "MEDIAN")
    b, q, off, names = frs.arrays()
    labels, seqs, quals, disc = [], [], [], []
    for i in range(1000):
        d = res["action"][i] == 2
        disc.append(d)
        seqs.append(bytes(b[int(off[i]):int(off[i + 1])]))
        quals.append(bytes(q[int(off[i]):int(off[i + 1])]))
        label = b""
        if not d:
            if res["action"][i] == 1:
                label += b"AFTrim:%d+%d " % (res["min_pass"][i], res["max_pass"][i] - res["min_pass"][i])
            if wt[i]:
                label += b"Trim:%d+%d " % (to[i], tl[i])
            label += b"MedianScore:%d" % int(sc[i] + 0.5)
        labels.append(label)
    # the device read set keeps qualities at Phred-33 (Read::FASTQ_START_CHAR, the reference's internal form)
    text = filterreads_output([nm.split(b" ")[0] for nm in names], seqs, quals, labels, disc, to, tl, sc, 2, mrl, both, qual_shift=out_base - 33, out_base=out_base)
    assert text.replace(b"\t", b" ") == gold.replace(b"\t", b" ")
    assert sum(disc) == {0.85: 5, 1.0: 51, 25.0: 0}[mrl]


def _spiked_reads(n, seed, tables, read_len=(30, 160)):
    """synthetic reads with what the filter reacts to: adapter / repeat / PhiX pieces (exact and with one or two substitutions)
    at random places, runs of low quality, N's, reads shorter than the match length"""
    rng = np.random.default_rng(seed)
    seqs = []
    cur = None
    for line in tables.split(b"\n"):
        if line.startswith(b">"):
            cur = []
            seqs.append(cur)
        elif cur is not None and line:
            cur.append(line)
    seqs = [b"".join(s) for s in seqs]
    out_s, out_q = [], []
    for i in range(n):
        L = int(rng.integers(read_len[0], read_len[1])) if rng.random() > 0.03 else int(rng.integers(1, 30))
        s = bytearray(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L).tobytes())
        q = bytearray(rng.choice(np.frombuffer(b"5:?DI", dtype=np.uint8), size=L).tobytes())
        r = rng.random()
        if r < 0.5 and L >= 30:
            a = seqs[int(rng.integers(0, len(seqs)))]
            a = a + a
            piece = bytearray(a[int(rng.integers(0, max(1, len(a) // 2))):][:int(rng.integers(20, 60))])
            for _ in range(int(rng.integers(0, 3))):
                if len(piece):
                    piece[int(rng.integers(0, len(piece)))] = b"ACGT"[int(rng.integers(0, 4))]
            at = int(rng.integers(0, L))
            s[at:at + len(piece)] = piece
            s = s[:L]
        if rng.random() < 0.4 and L:
            for _ in range(int(rng.integers(1, 4))):
                at, ln = int(rng.integers(0, L)), int(rng.integers(1, 12))
                q[at:at + ln] = b"#" * len(q[at:at + ln])
        if rng.random() < 0.1 and L:
            s[int(rng.integers(0, L))] = ord("N")
        out_s.append(bytes(s))
        out_q.append(bytes(q))
    return ReadBatch(out_s, out_q)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(edit_distance=1, min_read_length=25.0), dict(edit_distance=0, min_read_length=0.5),
                                dict(edit_distance=3), dict(build_edits=0, edit_distance=1), dict(build_edits=0, edit_distance=2, min_read_length=0.0)])
def test_screen_matches_oracle_on_spiked_reads(kw):
    """applyFilterToRead / recordAffectedRead on reads with adapters (0-2 substitutions), low-quality runs, N's and short reads;
    query-time edits 0, 1 and 2"""
    import kmernator_amd as ka
    table = fasta("artifact_sequences.fa")
    rb = _spiked_reads(3000 if kw.get("build_edits", 2) else 600, 7, table)
    o = OracleArtifactFilter(artifact_config(**kw), table)
    sp, f = _device_filter(kw, table)
    rs = ka.ReadSet(sp, _fastq_text(rb), input_quality_base=33)
    assert rs.n == rb.n
    want, frs, _ = _check_apply(f, o, rs, rb)
    assert (want["action"] == 1).sum() > 50 and ((want["action"] == 2).sum() > 10 or kw.get("min_read_length") == 0.0)
    assert kw.get("min_read_length") == 0.5 or (want["remnant_len"] > 0).sum() > 5      # two runs of half a read cannot both exist
    assert ((want["value"] > 0) & (want["value"] < o.info()[0])).sum() > (100 if rb.n >= 3000 else 20)        # real artifact hits, not only quality trims


@pytest.mark.gpu
def test_pairs_repeats_phix_and_reference_sequences():
    """applyFilterToPair: PhiX in one read discards the pair, a hit to an --artifact-reference-file sequence discards instead of
    trimming, simple repeats in the middle of a read with good margins are let through"""
    import kmernator_amd as ka
    text, cls = _all_tables()
    rng = np.random.default_rng(3)
    ref = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=400).tobytes())
    nseq = text.count(b">")
    text = text + b">reference_1\n" + ref + b"\n"
    kw = dict(edit_distance=1, min_read_length=0.3, reference_begin=1 + nseq, **cls)
    spikes = fasta("phix.fa") + b">reference_1\n" + ref + b"\n"            # pieces come from PhiX, the reference sequence,
    spikes = spikes * 8 + fasta("artifact_sequences.fa") + fasta("simple_repeats.fa")[:4000]      # the adapters and a few repeats
    rb = _spiked_reads(4000, 11, spikes, read_len=(60, 200))
    mate = np.arange(rb.n, dtype=np.int64) ^ 1
    mate[-100:] = -1                                          # the last 100 reads are unpaired
    o = OracleArtifactFilter(artifact_config(**kw), text)
    sp, f = _device_filter(kw, text)
    rs = ka.ReadSet(sp, _fastq_text(rb), input_quality_base=33)
    want, frs, _ = _check_apply(f, o, rs, rb, mate)
    n_all = o.info()[0]
    assert (want["value"] == cls["phix_idx"]).sum() > 20
    assert ((want["value"] >= cls["simple_repeat_begin"]) & (want["value"] < cls["simple_repeat_end"])).sum() > 20
    assert ((want["value"] >= kw["reference_begin"]) & (want["value"] < n_all)).sum() > 5
    # a clean read whose mate hit PhiX is discarded too
    clean_discarded = (want["value"] == 0) & (want["action"] == 2)
    assert clean_discarded.sum() > 5
    # and without the mates it is not
    want1, _, _ = _check_apply(f, o, rs, rb, None)
    assert not ((want1["value"] == 0) & (want1["action"] != 0)).any()


@pytest.mark.gpu
def test_filter_errors_and_empty_batches():
    import kmernator_amd as ka
    sp = ka.KmerSpectrum(ka.default_config(31, estimated_raw_kmers=1000, device=0))
    with pytest.raises(ka.KmerSpectrumError):
        ka.FilterKnownOddities(sp, fasta("artifact_sequences.fa"), match_length=30)
    with pytest.raises(ka.KmerSpectrumError):
        ka.FilterKnownOddities(sp, fasta("artifact_sequences.fa"), match_length=22)
    f = ka.FilterKnownOddities(sp, b"", edit_distance=1)                   # no sequences: only the quality screen acts
    assert (f.n_sequences, f.n_filter_kmers) == (1, 0)
    rs = ka.ReadSet(sp, b"@a\nACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIII#IIIIIIIIIIIIIII\n", input_quality_base=33)
    res, frs = f.applyFilter(rs)
    assert (res["value"][0], res["action"][0], res["min_pass"][0], res["max_pass"][0]) == (1, 1, 0, 16)
    assert frs.n == 2 and (res["remnant_off"][0], res["remnant_len"][0]) == (17, 15)      # 15 of 32 >= 0.40: rescued
    empty = ka.ReadSet(sp, b"")
    res, frs = f.applyFilter(empty)
    assert frs.n == 0 and res["action"].size == 0


@pytest.mark.gpu
def test_filter_on_reads_handed_over_as_host_arrays():
    """kmr_reads_from_host: the route the reference-side shim takes (ReadSet flattened by the host, include/kmernator_amd_shim.hpp)"""
    import kmernator_amd as ka
    table = fasta("artifact_sequences.fa")
    rb = _spiked_reads(2000, 21, table)
    kw = dict(edit_distance=1, min_read_length=25.0)
    o = OracleArtifactFilter(artifact_config(**kw), table)
    sp, f = _device_filter(kw, table)
    rs = ka.ReadSet.from_arrays(sp, rb.bases, rb.quals, rb.offsets)
    assert rs.n == rb.n and rs.total_bases == rb.bases.size
    _check_apply(f, o, rs, rb)


@pytest.mark.gpu
def test_screen_matches_oracle_at_scale():
    """200 000 spiked reads of 30-300 bases through the reference's default settings (both substitution rounds built in)"""
    import kmernator_amd as ka
    table = fasta("artifact_sequences.fa")
    rb = _spiked_reads(200000, 31, table, read_len=(30, 300))
    o = OracleArtifactFilter(artifact_config(), table)
    sp, f = _device_filter({}, table)
    rs = ka.ReadSet.from_arrays(sp, rb.bases, rb.quals, rb.offsets)
    want, frs, _ = _check_apply(f, o, rs, rb)
    assert (want["action"] == 1).sum() > 50000 and (want["remnant_len"] > 0).sum() > 5000
