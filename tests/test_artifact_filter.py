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
