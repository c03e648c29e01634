"""The plumbing the full-size parity tests stand on, checked on the CPU: SURVEY.md 8(d)'s generator is pinned to committed
values, the map digest of include/kmernator_amd.h (kmr_map_digest / orc_map_digest) is held to a numpy restatement over the
bytes of the stored map, and the oracle digests under tests/golden/full_size_digests.json (built part by part by
tests/golden/make_full_size_digests.py) are reproduced by ONE serial oracle build of the whole input at the sizes that takes
seconds -- which is what lets part digests stand for a whole spectrum at C2 / C4 size."""
import hashlib

import numpy as np
import pytest

from helpers import (KMR_MAP_SINGLETON, KMR_MAP_WEAK, KMR_VALUE_EXT, OracleSpectrum, default_config, digest_of_image, digests_agree,
                     full_size_golden, synth_reads_8d)


def _h(a):
    return hashlib.blake2b(a.tobytes(), digest_size=8).hexdigest()


def test_generator_known_answers():
    """the first reads of job seed 1 (C2's), flat and noisy, and a slice far into a job (global read indices, not positions in the
    call): committed digests, so that neither statement of the generator can drift unnoticed"""
    rb = synth_reads_8d(1, 0, 1_000_000, 150, 5_000_000, True)
    assert rb.seq(0) == b"CGACCCGTTTTCAGTAGGTGCGAAACAAATATTACCGTCCCCGGAGGGGACTTGCGAATGGGTAGACTTGGGCGCGGTCGGCTATGGGATCCATTGTGATACGTACCTACGCTGACAGGCGCCCTCTAAGGGAAGGCGGGGTCGGTTTTG"
    assert (_h(rb.bases), _h(rb.quals)) == ("9f4046b99b531874", "bb8f8202ba5937d0")
    flat = synth_reads_8d(1, 0, 1000, 150, 5_000_000, False)
    assert np.array_equal(flat.bases, rb.bases[:150_000]) and np.all(flat.quals == ord("I"))
    # a slice is a slice of the job: reads 700..900 generated alone equal reads 700..900 of the first thousand
    part = synth_reads_8d(1, 700, 200, 150, 5_000_000, True, threads=3)
    assert np.array_equal(part.bases, rb.bases[700 * 150:900 * 150]) and np.array_equal(part.quals, rb.quals[700 * 150:900 * 150])
    # error rate 1 %, quality mix .80/.10/.05/.04/.01 with errors at Q10
    q = np.bincount(rb.quals[:3_000_000], minlength=128) / 3e6
    assert abs(q[73] - 0.80 * 0.99) < 2e-3 and abs(q[63] - 0.10 * 0.99) < 1e-3 and abs(q[53] - 0.05 * 0.99) < 1e-3
    assert abs(q[43] - (0.04 * 0.99 + 0.01)) < 1e-3 and abs(q[35] - 0.01 * 0.99) < 5e-4


@pytest.mark.parametrize("k,ext,min_depth", [(31, False, 2), (51, False, 1), (21, True, 1), (21, True, 2)])
def test_map_digest_is_what_the_header_says(k, ext, min_depth):
    """orc_map_digest against the numpy restatement over the stored map's bytes (weak and singleton maps, 12 / 60 / 1 / 5-byte values)"""
    rb = synth_reads_8d(21, 0, 3000, 150, 20_000, True)
    kw = dict(value_kind=KMR_VALUE_EXT, min_weight=0.0, min_quality_score=2) if ext else {}
    o = OracleSpectrum(default_config(k, estimated_raw_kmers=3000 * (150 - k + 1), **kw))
    o.add_reads(rb)
    o.finalize(min_depth)
    kb = (k + 3) // 4
    d = o.digest(KMR_MAP_WEAK)
    assert d["entries"] == o.stats()["weak_entries"] > 1000
    assert digests_agree(d, digest_of_image(o.image(KMR_MAP_WEAK), kb, ext=ext), rel=1e-12)
    ds = o.digest(KMR_MAP_SINGLETON)
    if min_depth == 1:
        assert ds["entries"] == o.stats()["singleton_kmers"] > 100
        assert digests_agree(ds, digest_of_image(o.image(KMR_MAP_SINGLETON), kb, ext=ext, singleton=True), rel=1e-12)
    else:
        assert ds["entries"] == 0


@pytest.mark.parametrize("name", ["small_k31_noisy", "small_k51_flat"])
def test_part_digests_add_up_to_the_whole_build(name):
    """the committed digest (sum of the parts' digests) == the digest of one serial oracle build of the whole input, statistics
    included: exact, weightedCount too (each k-mer meets the same weights in the same order in its part as in the whole)"""
    g = full_size_golden(name)
    c = g["config"]
    rb = synth_reads_8d(c["seed"], 0, c["reads"], c["read_len"], c["genome"], c["noisy"])
    o = OracleSpectrum(default_config(c["k"], estimated_raw_kmers=c["reads"] * (c["read_len"] - c["k"] + 1)))
    o.add_reads(rb)
    o.finalize(c["min_depth"])
    assert o.stats() == g["stats"]
    assert digests_agree(o.digest(KMR_MAP_WEAK), g["weak_digest"], rel=1e-12)


def test_oracle_merge_add_against_a_joint_build():
    """orc_merge_add (KmerMapByKmerArrayPair::mergeAdd, src/Kmer.h:3209-3261) of the weak maps of two spectra built WITHOUT a singleton
    map (so that no first sighting is set aside) == the weak map of one build over both read sets: keys, counts and direction
    biases exactly, weightedCount up to the float additions' order -- the self-consistency the GPU test of kmr_merge_image leans on"""
    from helpers import synth_reads
    a = synth_reads(1500, read_len=120, genome_len=12000, seed=5, quality="noisy", n_rate=0.002)
    b = synth_reads(1500, read_len=120, genome_len=12000, seed=5, quality="noisy", n_rate=0.002)
    b.bases[:] = np.roll(b.bases.reshape(1500, 120), 7, axis=0).reshape(-1)          # the same genome, the reads in another order ...
    b.quals[:] = np.roll(b.quals.reshape(1500, 120), 311, axis=0).reshape(-1)        # ... under other qualities
    cfg = default_config(27, num_buckets_weak=256, num_buckets_singleton=256, separate_singletons=0)
    oa, ob, oj = OracleSpectrum(cfg), OracleSpectrum(cfg), OracleSpectrum(cfg)
    oa.add_reads(a); ob.add_reads(b)
    oj.add_reads(a); oj.add_reads(b, first_idx=1500)
    for o in (oa, ob, oj):
        o.finalize(1)
    oa.merge_add(ob)
    ka_, ca, da, wa, _ = oa.entries()
    kj, cj, dj, wj, _ = oj.entries()
    assert np.array_equal(ka_, kj) and np.array_equal(ca, cj) and np.array_equal(da, dj)
    assert np.all(np.abs(wa.astype(np.float64) - wj) <= 1e-5 * cj)
    assert ob.stats()["weak_entries"] == 0 or ob.entries()[0].shape[0] == 0          # the source map is emptied (src.clear(), :3259)
