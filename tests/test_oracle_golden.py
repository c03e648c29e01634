"""Pins the oracle against the reference's own end-to-end golden fixtures
(copied as data files into tests/golden/):
  phix.mercount.m21, phix.mergraph.m21.D2  <- MeraculousCounter --min-kmer-quality=0
      --min-quality-score=2 --kmer-size 21 --fastq-base-quality 64 1000.fastq
      (test/runMeraculousTests.sh:40-74), compared after sort
  1000-Filtered.fastq <- FilterReads --kmer-scoring-type MEDIAN ... 31 1000.fastq
      (test/runFilterTests.sh:24-41): MedianScore/Trim labels of the 949 reads the
      artifact filter did not touch.
"""
import os
import re

import numpy as np
import pytest

from helpers import (GOLDEN, KMR_MAP_SINGLETON, KMR_MAP_WEAK, KMR_VALUE_EXT, OracleSpectrum, ReadBatch, default_config,
                     oracle_weighted_kmers, parse_image, read_fastq, synth_reads)
from refsemantics import median_trim_label


def meraculous_cfg(**kw):
    return default_config(21, value_kind=KMR_VALUE_EXT, min_weight=0.0, min_quality_score=2, fastq_start_char=64,
                          estimated_raw_kmers=56000, **kw)


def sorted_lines(path):
    with open(path) as f:
        return sorted(f.read().splitlines())


@pytest.mark.parametrize("threads", [1, 3])
def test_phix_mercount_and_mergraph(tmp_path, threads):
    rb = read_fastq(os.path.join(GOLDEN, "1000.fastq"))
    s = OracleSpectrum(meraculous_cfg())
    s.add_reads(rb, threads=threads)
    st = s.stats()
    assert (st["raw_kmers"], st["unique_kmers"], st["singleton_kmers"], st["weak_entries"]) == (56000, 5407, 6, 5401)
    s.finalize(2)
    s.dump(str(tmp_path / "c"), 2, False)
    s.dump(str(tmp_path / "g"), 2, True)
    assert sorted_lines(tmp_path / "c") == sorted_lines(os.path.join(GOLDEN, "phix.mercount.m21"))
    assert sorted_lines(tmp_path / "g") == sorted_lines(os.path.join(GOLDEN, "phix.mergraph.m21.D2"))


def test_phix_owner_partitions_union(tmp_path):
    """_buildKmerSpectrumMPI owner partition (src/DistributedFunctions.h:433): the union of
    the per-owner spectra equals the single-partition spectrum (np sweep of runMeraculousTests.sh)."""
    rb = read_fastq(os.path.join(GOLDEN, "1000.fastq"))
    for world in (2, 3):
        for r in range(world):
            s = OracleSpectrum(meraculous_cfg(rank=r, world_size=world))
            s.add_reads(rb)
            s.finalize(2)
            s.dump(str(tmp_path / ("c%d" % world)), 2, False)
        assert sorted_lines(tmp_path / ("c%d" % world)) == sorted_lines(os.path.join(GOLDEN, "phix.mercount.m21"))


@pytest.mark.parametrize("fq,start", [("1000.fastq", 64), ("1000.std.fastq", 33)])
def test_filterreads_k31_labels(fq, start):
    k = 31
    rb = read_fastq(os.path.join(GOLDEN, fq))
    gold = read_fastq(os.path.join(GOLDEN, "1000-Filtered.fastq"))
    cfg = default_config(k, fastq_start_char=start, estimated_raw_kmers=(76 - k + 1) * 1000)
    s = OracleSpectrum(cfg)
    s.add_reads(rb)
    s.finalize(2)
    st = s.stats()
    assert (st["raw_kmers"], st["raw_good_kmers"], st["unique_kmers"], st["weak_entries"]) == (46000, 45846, 5409, 5380)
    checked = 0
    for i in range(rb.n):
        if b"AFTrim" in gold.names[i]:
            continue
        keys, w, ext = oracle_weighted_kmers(cfg, rb.seq(i), rb.qual(i))
        label = median_trim_label(s.lookup(keys), k)
        assert label == gold.names[i].split(b" ", 1)[1], (i, gold.names[i])
        checked += 1
    assert checked == 949


def test_image_round_trip_and_layout():
    """store() -> restore() (test/KmerTest.cpp:545-594 testStore; runFilterTests.sh:72-74)."""
    rb = read_fastq(os.path.join(GOLDEN, "1000.fastq"))
    cfg = default_config(21, fastq_start_char=64, num_buckets_weak=64, num_buckets_singleton=256)
    s = OracleSpectrum(cfg)
    s.add_reads(rb)
    s.finalize(1)                      # keep singletons
    for which, vsize in ((KMR_MAP_WEAK, 12), (KMR_MAP_SINGLETON, 1)):
        img = s.image(which)
        nb, mask, buckets = parse_image(img, s.kb, vsize)
        assert nb == s.num_buckets(which) and mask == nb - 1
        n_entries = sum(len(k) for k, _ in buckets)
        assert img.size == 8 * (2 + nb) + 4 * nb + n_entries * (s.kb + vsize)
        lib_hash = __import__("helpers").oracle_lib().orc_hash
        for b, (keys, vals) in enumerate(buckets):
            prev = None
            for kk in keys:
                kb_ = kk.tobytes()
                assert lib_hash(kb_, len(kb_)) & mask == b
                assert prev is None or prev < kb_
                prev = kb_
        s2 = OracleSpectrum(cfg)
        s2.load_image(which, img)
        assert np.array_equal(s2.image(which), img)
    # lookups on a restored spectrum equal the original
    s3 = OracleSpectrum(cfg)
    s3.load_image(KMR_MAP_WEAK, s.image(KMR_MAP_WEAK))
    s3.load_image(KMR_MAP_SINGLETON, s.image(KMR_MAP_SINGLETON))
    keys, w, ext = oracle_weighted_kmers(cfg, rb.seq(5), rb.qual(5))
    assert np.array_equal(s.lookup(keys), s3.lookup(keys))


@pytest.mark.parametrize("k", [21, 31, 51])
def test_threads_and_batches_do_not_change_counts(k):
    rb = synth_reads(3000, read_len=100, seed=5, quality="noisy", n_rate=0.002)
    cfg = default_config(k, estimated_raw_kmers=3000 * (100 - k + 1))
    a = OracleSpectrum(cfg)
    a.add_reads(rb, threads=1)
    a.finalize(2)
    b = OracleSpectrum(cfg)
    b.add_reads(rb.slice(0, 1100), threads=4)
    b.add_reads(rb.slice(1100, 3000), first_idx=1100, threads=2)
    b.finalize(2)
    ka, ca, da, wa, _ = a.entries()
    kb_, cb, db, wb, _ = b.entries()
    assert np.array_equal(ka, kb_) and np.array_equal(ca, cb)
    assert a.stats() == b.stats()
    # weightedCount is order dependent: whichever sighting comes first is quantised to
    # 1/254 steps in the singleton map (src/KmerTrackingData.h:646,658)
    assert np.all(np.abs(wa - wb) <= 1.0 / 254 + 1e-4 * np.maximum(wa, 1))


def test_count_saturation_and_direction():
    """TrackingData::track stops at 65535 (src/KmerTrackingData.h:434); the first sighting's
    direction is lost through the singleton (TrackingDataSingleton::getDirectionBias, :654)."""
    k = 9
    seq = b"ACGTTGCAAGGCTA"          # 6 9-mers
    n = 66000
    rb = ReadBatch([seq] * n, [b"I" * len(seq)] * n)
    cfg = default_config(k, num_buckets_weak=16, num_buckets_singleton=16)
    s = OracleSpectrum(cfg)
    s.add_reads(rb)
    s.finalize(2)
    keys, cnt, dirb, w, _ = s.entries()
    assert len(cnt) == 6 and np.all(cnt == 65535)
    fwd = s.stats()
    assert fwd["raw_good_kmers"] == 6 * n and fwd["unique_kmers"] == 6
    # every occurrence has the same orientation: bias is 0 or count-1 (first sighting lost)
    assert set(int(d) for d in dirb) <= {0, 65534}


@pytest.mark.parametrize("gold_name,mrl,both,out_base", [
    ("1000-Filtered-0.85.fastq", 0.85, False, 64), ("1000-Filtered-0.85.std.fastq", 0.85, False, 33),
    ("1000-Filtered-readlength.fastq", 1.0, False, 64), ("1000-Filtered-readlength-both.fastq", 1.0, True, 64), ("1000-Filtered.fastq", 25.0, False, 64)])
@pytest.mark.parametrize("fq,start", [("1000.fastq", 64), ("1000.std.fastq", 33)])
def test_filterreads_selection_goldens(fq, start, gold_name, mrl, both, out_base):
    """test/runFilterTests.sh:43-63: the reference's other FilterReads outputs -- `--min-read-length 0.85` (the artifact filter
    trims where 85 % of the read survive and discards otherwise; the spectrum, hence the scores, are those of the reads so filtered),
    `--min-read-length 1` (only untrimmed reads pass; a pair is kept when either read passes) and `--min-passing-in-pair 2` (both
    must) -- reproduced WHOLE FILE, byte for byte (names, labels, trimmed sequences and qualities, the one-base placeholder of an
    emptied read, the order), from either quality encoding of the input to either encoding of the output: artifact filter ->
    spectrum of the filtered reads -> scoring / trimming (all the oracle's) -> isPassingRead / isPassingPair / writePicks
    (tests/refsemantics.py).  (test/ also holds `-readlength.std`, `-readlength-both.std` and `1000-Filtered.std.fastq`: the script
    uses none of them and they carry no labels at all -- files of an older version, not fixtures.)"""
    import re
    from helpers import OracleArtifactFilter, apply_artifact_result, artifact_config, oracle_weighted_kmers
    from refsemantics import filterreads_output, score_and_trim
    k = 31
    rb = read_fastq(os.path.join(GOLDEN, fq))
    gold = open(os.path.join(GOLDEN, gold_name), "rb").read()
    f = OracleArtifactFilter(artifact_config(edit_distance=1, fastq_start_char=start, min_read_length=mrl),
                             open(os.path.join(GOLDEN, "artifact_sequences.fa"), "rb").read())
    res = f.apply(rb)
    assert not res["remnant_len"].any()
    fr = apply_artifact_result(rb, res)
    cfg = default_config(k, fastq_start_char=start, estimated_raw_kmers=(76 - k + 1) * 1000)
    s = OracleSpectrum(cfg)
    s.add_reads(fr)
    s.finalize(2)
    labels, to, tl, sc, disc = [], [], [], [], []
    for i in range(rb.n):
        d = res["action"][i] == 2
        disc.append(d)
        if d:
            labels.append(b""); to.append(0); tl.append(0); sc.append(0.0)
            continue
        seq = fr.seq(i)
        keys, w, ext = oracle_weighted_kmers(cfg, seq, fr.qual(i))
        o, l, score, trimmed = score_and_trim(s.lookup(keys) if len(keys) else [], seq, k, 2, "MEDIAN")
        label = b""
        if res["action"][i] == 1:
            label += b"AFTrim:%d+%d " % (res["min_pass"][i], res["max_pass"][i] - res["min_pass"][i])
        if trimmed:
            label += b"Trim:%d+%d " % (o, l)
        label += b"MedianScore:%d" % int(score + 0.5)
        labels.append(label); to.append(o); tl.append(l); sc.append(score)
    names = [nm.split(b" ")[0] for nm in rb.names]
    text = filterreads_output(names, [fr.seq(i) for i in range(rb.n)], [fr.qual(i) for i in range(rb.n)], labels, disc, to, tl, sc,
                              2, mrl, both, qual_shift=out_base - start, out_base=out_base)
    text, gold = text.replace(b"\t", b" "), gold.replace(b"\t", b" ")          # the script compares with diff -w
    if text != gold:
        a, b = text.split(b"\n"), gold.split(b"\n")
        first = next((j for j in range(min(len(a), len(b))) if a[j] != b[j]), None)
        raise AssertionError((len(a), len(b), first, a[first - 1:first + 3] if first is not None else None, b[first - 1:first + 3] if first is not None else None))
