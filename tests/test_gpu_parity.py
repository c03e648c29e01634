"""Parity of the HIP path (through the C-ABI) with the oracle on the same inputs:
bit-exact k-mer sets, counts, direction bias, extension tallies, statistics, singleton
bytes and on-disk images; weightedCount within the documented tolerance (it is order
dependent in the reference itself: the first sighting is quantised to 1/254 steps,
src/KmerTrackingData.h:646,658)."""
import os

import numpy as np
import pytest

import kmernator_amd as ka
from helpers import (GOLDEN, KMR_MAP_SINGLETON, KMR_MAP_WEAK, KMR_VALUE_EXT, OracleSpectrum, ReadBatch, default_config,
                     oracle_weighted_kmers, parse_image, read_fastq, synth_reads)
from refsemantics import median_trim_label, score_and_trim

pytestmark = pytest.mark.gpu


MODES = [1, 2, 3]   # kmr_config.build_mode: 1 = open-addressed device table, 2 = streaming partition + LDS counting, 3 = super-k-mer lists


def product(cfg, mode=0, **tune):
    c = ka.default_config(cfg.k)
    for name, _ in cfg._fields_:
        setattr(c, name, getattr(cfg, name))
    c.build_mode = mode
    if mode == 3 and cfg.k < 13:
        pytest.skip("build_mode 3 (super-k-mer lists) needs k >= 13")
    return ka.KmerSpectrum(c).tune(**tune)


def add(sp, rb, first=0):
    sp.buildKmerSpectrum(rb.bases, rb.quals, rb.offsets, first, rb.discarded)


def compare_weak_images(img_o, img_p, kb, ext, dir_tol=0, saturated_dir_free=False, first_tol=0.0, exact_from=None):
    """first_tol: 1/254 where the two sides may disagree about which sighting was the first (arrival order through an
    exchange, as in the reference's own MPI build), 0 against the serial oracle"""
    vsize = 60 if ext else 12
    nb, mask, bo = parse_image(img_o, kb, vsize)
    nb2, mask2, bp = parse_image(img_p, kb, vsize)
    assert (nb, mask) == (nb2, mask2)
    assert img_o.size == img_p.size
    assert np.array_equal(img_o[:16 + 8 * nb], img_p[:16 + 8 * nb])   # header + offsets
    n = 0
    for (ko, vo), (kp, vp) in zip(bo, bp):
        assert np.array_equal(ko, kp)
        if len(ko) == 0:
            continue
        vo32 = np.ascontiguousarray(vo).view(np.uint32).reshape(len(ko), vsize // 4)
        vp32 = np.ascontiguousarray(vp).view(np.uint32).reshape(len(kp), vsize // 4)
        assert np.array_equal(vo32[:, 0] & 0xffff, vp32[:, 0] & 0xffff)            # count
        do, dp = (vo32[:, 2] & 0xffff).astype(np.int64), (vp32[:, 2] & 0xffff).astype(np.int64)
        # directionBias; for a k-mer seen more than 65 535 times the reference stops counting directions (and weights) at
        # its 65 535th sighting in arrival order (src/KmerTrackingData.h:435-447,517-529), which no order-free count reproduces
        free = ((vo32[:, 0] & 0xffff) == 65535) if saturated_dir_free else np.zeros(len(ko), dtype=bool)
        assert np.all((np.abs(do - dp) <= dir_tol) | free)
        wo, wp = vo32[:, 1].view(np.float32), vp32[:, 1].view(np.float32)
        cnt = (vo32[:, 0] & 0xffff).astype(np.float64)
        # weightedCount: the serial reference adds f32 weights in read order onto the first sighting's weight as the singleton
        # map kept it ((unsigned char)(w * 254) / 254); the product takes that quantisation from its first-sighting word and
        # rounds an f64 sum once: SURVEY section 7's contract, |d| <= 1e-5 * count
        assert np.all((np.abs(wo.astype(np.float64) - wp) <= first_tol + 1e-5 * cnt) | free)
        if exact_from is not None:      # k-mers seen that often are accumulated in the reference's order and precision: bit for bit
            hot = cnt >= exact_from
            assert np.array_equal(vo32[hot, 1], vp32[hot, 1])
        if ext:
            assert np.array_equal(vo32[:, 3:], vp32[:, 3:])
        n += len(ko)
    return n


def run_both(cfg, rb, min_depth=2, batches=None, mode=0, **tune):
    o = OracleSpectrum(cfg)
    p = product(cfg, mode, **tune)
    if batches is None:
        o.add_reads(rb)
        add(p, rb)
    else:
        lo = 0
        for hi in batches + [rb.n]:
            o.add_reads(rb.slice(lo, hi), first_idx=lo)
            add(p, rb.slice(lo, hi), first=lo)
            lo = hi
    so, sp_ = o.stats(), p.stats()
    for key in ("raw_kmers", "raw_good_kmers", "discarded", "reads"):
        assert so[key] == sp_[key], (key, so, sp_)
    o.finalize(min_depth)
    p.finalize(min_depth)
    so, sp_ = o.stats(), p.stats()
    assert so == sp_, (so, sp_)
    return o, p


@pytest.mark.parametrize("mode", MODES)
def test_phix_meraculous_goldens(tmp_path, mode):
    """Config 5: MeraculousCounter k=21 on 1000.fastq, text equal to the reference's goldens."""
    rb = read_fastq(os.path.join(GOLDEN, "1000.fastq"))
    cfg = default_config(21, value_kind=KMR_VALUE_EXT, min_weight=0.0, min_quality_score=2, fastq_start_char=64, estimated_raw_kmers=56000)
    o, p = run_both(cfg, rb, mode=mode)
    p.dumpCounts(str(tmp_path / "c"), 2)
    p.dumpGraphs(str(tmp_path / "g"), 2)
    for got, exp in (("c", "phix.mercount.m21"), ("g", "phix.mergraph.m21.D2")):
        a = sorted(open(tmp_path / got).read().splitlines())
        b = sorted(open(os.path.join(GOLDEN, exp)).read().splitlines())
        assert a == b
    assert compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, True) == 5401


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("fq,start", [("1000.fastq", 64), ("1000.std.fastq", 33)])
@pytest.mark.parametrize("k", [21, 31])
def test_filterreads_fixture(fq, start, k, mode):
    """Config 1 (k=21) and the reference's own FilterReads golden (k=31 labels)."""
    rb = read_fastq(os.path.join(GOLDEN, fq))
    cfg = default_config(k, fastq_start_char=start, estimated_raw_kmers=(76 - k + 1) * 1000)
    o, p = run_both(cfg, rb, mode=mode)
    compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False)
    if k == 31:
        gold = read_fastq(os.path.join(GOLDEN, "1000-Filtered.fastq"))
        counts, off = p.getCountsForReads(rb.bases, rb.offsets)
        checked = 0
        for i in range(rb.n):
            if b"AFTrim" in gold.names[i]:
                continue
            label = median_trim_label(counts[int(off[i]):int(off[i + 1])], k)
            assert label == gold.names[i].split(b" ", 1)[1]
            checked += 1
        assert checked == 949


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("k", [5, 21, 31, 32, 33, 51, 64, 65, 95, 127])
def test_synthetic_noisy_reads(k, mode):
    """multi-word keys (k>32), N bases, sub-threshold qualities, ragged lengths"""
    rl = max(100, k + 40)
    rb = synth_reads(4000, read_len=rl, seed=k, quality="noisy", n_rate=0.003)
    # ragged: chop some reads, including shorter than k and empty
    rng = np.random.default_rng(k)
    seqs, quals = [], []
    for i in range(rb.n):
        L = rl
        r = rng.random()
        if r < 0.05:
            L = int(rng.integers(0, k + 3))
        elif r < 0.3:
            L = int(rng.integers(k, rl + 1))
        seqs.append(rb.seq(i)[:L])
        quals.append(rb.qual(i)[:L])
    rb2 = ReadBatch(seqs, quals)
    cfg = default_config(k, estimated_raw_kmers=4000 * (rl - k + 1))
    o, p = run_both(cfg, rb2, batches=[1500, 1501, 3000], mode=mode)
    n = compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False)
    assert n == o.stats()["weak_entries"] and n > 0
    # lookups: every k-mer of a few reads, plus absent keys
    for i in (0, 7, 100):
        if len(seqs[i]) >= k:
            keys, w, ext = oracle_weighted_kmers(cfg, seqs[i], quals[i])
            assert np.array_equal(o.lookup(keys), p.getCount(keys))
    absent = np.zeros((4, p.kb), dtype=np.uint8)
    absent[:, -1] = 0
    assert np.array_equal(o.lookup(absent), p.getCount(absent))


@pytest.mark.parametrize("mode", MODES)
def test_singleton_map_and_min_depth_variants(mode):
    rb = synth_reads(2000, read_len=100, seed=9, quality="noisy")
    for min_depth in (1, 2, 3):
        for sep in (1, 0):
            cfg = default_config(25, separate_singletons=sep, num_buckets_weak=512, num_buckets_singleton=2048)
            o, p = run_both(cfg, rb, min_depth=min_depth, mode=mode)
            compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False)
            so = o.image(KMR_MAP_SINGLETON)
            sp_ = p.image(KMR_MAP_SINGLETON)
            assert np.array_equal(so, sp_)      # 1-byte values: bit-exact, including the quantised weight


@pytest.mark.parametrize("mode,k", [(1, 19), (2, 19), (2, 41), (2, 95)])
def test_ext_singletons_and_image_reload(mode, k):
    rb = synth_reads(1500, read_len=110 if k > 32 else 90, seed=4, quality="noisy", n_rate=0.002)
    cfg = default_config(k, value_kind=KMR_VALUE_EXT, min_weight=0.0, min_quality_score=2, num_buckets_weak=256, num_buckets_singleton=256)
    o, p = run_both(cfg, rb, min_depth=1, mode=mode)
    assert np.array_equal(o.image(KMR_MAP_SINGLETON), p.image(KMR_MAP_SINGLETON))
    wimg = p.image(KMR_MAP_WEAK)
    compare_weak_images(o.image(KMR_MAP_WEAK), wimg, p.kb, True)
    # store -> restore (test/KmerTest.cpp:545-594; runFilterTests.sh:72-74): product image into a fresh
    # product handle and into the oracle, oracle image into the product
    q = product(cfg)
    q.load_image(KMR_MAP_WEAK, wimg)
    q.load_image(KMR_MAP_SINGLETON, p.image(KMR_MAP_SINGLETON))
    assert np.array_equal(q.image(KMR_MAP_WEAK), wimg)
    assert np.array_equal(q.image(KMR_MAP_SINGLETON), p.image(KMR_MAP_SINGLETON))
    o2 = OracleSpectrum(cfg)
    o2.load_image(KMR_MAP_WEAK, wimg)
    assert np.array_equal(o2.image(KMR_MAP_WEAK), wimg)
    r = product(cfg)
    r.load_image(KMR_MAP_WEAK, o.image(KMR_MAP_WEAK))
    keys, w, ext = oracle_weighted_kmers(cfg, rb.seq(3), rb.qual(3))
    o3 = OracleSpectrum(cfg)
    o3.load_image(KMR_MAP_WEAK, o.image(KMR_MAP_WEAK))
    assert np.array_equal(r.getCount(keys), o3.lookup(keys))
    r.load_image(KMR_MAP_SINGLETON, o.image(KMR_MAP_SINGLETON))
    assert np.array_equal(r.getCount(keys), o.lookup(keys))


@pytest.mark.parametrize("mode", MODES)
def test_count_saturation(mode):
    k = 9
    seq = b"ACGTTGCAAGGCTA"
    n = 66000
    rb = ReadBatch([seq] * n, [b"I" * len(seq)] * n)
    cfg = default_config(k, num_buckets_weak=16, num_buckets_singleton=16)
    o, p = run_both(cfg, rb, mode=mode)
    keys, w, ext = oracle_weighted_kmers(cfg, seq, b"I" * len(seq))
    assert np.all(p.getCount(keys) == 65535)
    assert np.array_equal(o.lookup(keys), p.getCount(keys))


def test_table_growth():
    rb = synth_reads(20000, read_len=100, seed=12, err=0.05)
    cfg = default_config(31, max_table_entries=1000, num_buckets_weak=1024, num_buckets_singleton=4096)
    o, p = run_both(cfg, rb, batches=[100, 5000], mode=1)
    compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False)


@pytest.mark.parametrize("mode", MODES)
def test_reference_reads_without_quals_and_discarded(mode):
    rb = synth_reads(500, read_len=120, seed=3, n_rate=0.01)
    disc = np.zeros(rb.n, dtype=np.uint8)
    disc[::7] = 1
    rbn = ReadBatch([rb.seq(i) for i in range(rb.n)], None, disc)
    cfg = default_config(21, num_buckets_weak=64, num_buckets_singleton=64)
    o, p = run_both(cfg, rbn, mode=mode)
    compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False)


@pytest.mark.parametrize("k", [13, 16, 17, 32, 33, 47, 64, 65, 96, 127])
def test_ext_values_on_the_lists_many_k(k):
    """extension values on the super-k-mer lists across key widths and the window geometries of small k: a k-mer's right neighbour comes
    out of the register window it was cut from -- except at k = 32 W, where it is the record's next dword -- its left one from the k-mer
    made before it; reads of 260 bases, N's, with and without a singleton map: tallies and singleton packets are the oracle's"""
    rb = synth_reads(1200, read_len=260, genome_len=9000, seed=300 + k, quality="noisy", n_rate=0.004)
    for sep in (1, 0):
        cfg = default_config(k, value_kind=KMR_VALUE_EXT, min_weight=0.0, min_quality_score=2, num_buckets_weak=128, num_buckets_singleton=256, separate_singletons=sep)
        o, p = run_both(cfg, rb, min_depth=1, mode=3)
        assert compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, True) == o.stats()["weak_entries"]
        if sep:
            assert np.array_equal(o.image(KMR_MAP_SINGLETON), p.image(KMR_MAP_SINGLETON))


@pytest.mark.parametrize("k", [21, 51])
def test_ext_values_with_filters_on_the_lists(k):
    """extension values through the FILTERING extraction of build_mode 3 (a hash partition of the k-mers, a sub-sample): the part's weak
    map with all its tallies and the singletons' packets are the oracle's"""
    rb = synth_reads(3000, read_len=110, seed=21, quality="noisy", n_rate=0.003)
    for kw in (dict(num_parts=3, part_idx=1), dict(kmer_subsample=3)):
        cfg = default_config(k, value_kind=KMR_VALUE_EXT, min_weight=0.0, min_quality_score=2, num_buckets_weak=256, num_buckets_singleton=512, **kw)
        o, p = run_both(cfg, rb, min_depth=1, mode=3)
        assert compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, True) == o.stats()["weak_entries"]
        assert np.array_equal(o.image(KMR_MAP_SINGLETON), p.image(KMR_MAP_SINGLETON))


@pytest.mark.parametrize("mode", MODES)
def test_owner_and_part_filters(mode):
    """getDistributedThreadId owner filter (src/Kmer.h:2284-2295) and --build-partitions (:1680)"""
    rb = synth_reads(3000, read_len=100, seed=8, quality="noisy")
    for kw in (dict(rank=1, world_size=3), dict(num_parts=4, part_idx=2), dict(kmer_subsample=3)):
        cfg = default_config(27, num_buckets_weak=256, num_buckets_singleton=1024, **kw)
        o, p = run_both(cfg, rb, mode=mode)
        compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False)


def test_histogram():
    rb = synth_reads(3000, read_len=100, seed=21)
    cfg = default_config(31, num_buckets_weak=256, num_buckets_singleton=1024)
    o, p = run_both(cfg, rb)
    co, wo = o.histogram(64)
    cp, wp = p.histogram(64)
    assert np.array_equal(co, cp)
    assert np.allclose(wo, wp, rtol=1e-3, atol=1.0)


@pytest.mark.parametrize("k,recycle", [(31, "1"), (31, "0"), (51, "1")])
def test_three_partition_levels_and_chunk_recycling(k, recycle):
    """More records than two partition passes can cut into countable lists (forced here by a tiny target list size)
    take a third pass; with recycle_chunks = 1 every pass after the first writes into the chunks it has just read instead
    of fresh ones.  Both must give the maps of the device-table build byte for byte."""
    rb = synth_reads(300000, read_len=150, seed=31, quality="noisy", n_rate=0.001)
    cfg = default_config(k, estimated_raw_kmers=300000 * (150 - k + 1))
    a = product(cfg, 2, target_list_records=6, recycle_chunks=int(recycle))
    add(a, rb)
    a.finalize(1)
    b = product(cfg, 1)
    add(b, rb)
    b.finalize(1)
    assert a.stats() == b.stats()
    assert np.array_equal(a.image(KMR_MAP_WEAK), b.image(KMR_MAP_WEAK))
    assert np.array_equal(a.image(KMR_MAP_SINGLETON), b.image(KMR_MAP_SINGLETON))
    # a second build on the same handle starts from a clean pool
    a.reset()
    add(a, rb.slice(0, 1000))
    a.finalize(1)
    c = product(cfg, 1)
    add(c, rb.slice(0, 1000))
    c.finalize(1)
    assert np.array_equal(a.image(KMR_MAP_WEAK), c.image(KMR_MAP_WEAK))


@pytest.mark.parametrize("min_depth", [1, 2])
def test_entry_buffers_grow_when_the_estimate_was_too_small(min_depth):
    """The count pass writes kept entries into buffers sized from a sampled share of repeated keys; if they overflow the
    pass is run again with larger ones (here the estimate is forced to almost nothing)."""
    rb = synth_reads(40000, read_len=100, seed=23, quality="noisy")
    cfg = default_config(31, estimated_raw_kmers=40000 * 70)
    o, p = run_both(cfg, rb, min_depth=min_depth, mode=2, entry_share=0.00001)
    compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False)
    assert np.array_equal(o.image(KMR_MAP_SINGLETON), p.image(KMR_MAP_SINGLETON))


@pytest.mark.parametrize("k", [31, 51])
def test_many_sub_batches_keep_level1_state(k):
    """A build cut into ~60 sub-batches (level-1 partition state carried from launch to launch, flushed once at finalize)
    and fed through several kmr_add_reads calls equals the oracle; a reset in between starts from an empty state."""
    rb = synth_reads(6000, read_len=120, seed=17, quality="noisy", n_rate=0.002)
    cfg = default_config(k, num_buckets_weak=512, num_buckets_singleton=2048)
    o, p = run_both(cfg, rb, mode=2, batches=[1000, 1001, 4000], sub_batch_bases=12000)
    compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False)
    p.reset()
    add(p, rb.slice(0, 500))          # left unfinished on purpose: its kept-back records must not leak into the next build
    p.reset()
    add(p, rb)
    p.finalize(2)
    assert p.stats() == o.stats()
    assert np.array_equal(p.image(KMR_MAP_WEAK)[:16 + 8 * 512], o.image(KMR_MAP_WEAK)[:16 + 8 * 512])


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("k,parts", [(31, 3), (51, 2)])
def test_build_in_parts_and_merge(k, parts, mode):
    """KmerSpectrum::buildKmerSpectrumInParts (src/KmerSpectrum.h:1818-1902): every part keeps the k-mers with
    getDMPThread(kmer, numParts) == partIdx, is purged and stored; the parts are restored and merged.  The merged maps
    equal the one-pass build bit for bit (the reference's own merge, mergeStripedBuckets, expects bucket-striped parts)."""
    rb = synth_reads(4000, read_len=120, seed=9, quality="noisy", n_rate=0.002)
    cfg = default_config(k, num_buckets_weak=512, num_buckets_singleton=1024)
    whole = product(cfg, mode)
    add(whole, rb)
    whole.finalize(1)
    acc = None
    for part in range(parts):
        c = default_config(k, num_buckets_weak=512, num_buckets_singleton=1024, num_parts=parts, part_idx=part)
        o = OracleSpectrum(c)
        o.add_reads(rb)
        o.finalize(1)
        p = product(c, mode)
        add(p, rb)
        p.finalize(1)
        assert p.stats() == o.stats()
        wi, si = p.image(KMR_MAP_WEAK), p.image(KMR_MAP_SINGLETON)
        assert np.array_equal(si, o.image(KMR_MAP_SINGLETON))
        if acc is None:
            acc = p
        else:
            acc.merge_image(KMR_MAP_WEAK, wi)
            acc.merge_image(KMR_MAP_SINGLETON, si)
    assert np.array_equal(acc.image(KMR_MAP_WEAK), whole.image(KMR_MAP_WEAK))
    assert np.array_equal(acc.image(KMR_MAP_SINGLETON), whole.image(KMR_MAP_SINGLETON))
    assert acc.stats()["weak_entries"] == whole.stats()["weak_entries"]
    with pytest.raises(ka.KmerSpectrumError, match="share"):
        acc.merge_image(KMR_MAP_SINGLETON, whole.image(KMR_MAP_SINGLETON))
    wrong = product(default_config(k, num_buckets_weak=256, num_buckets_singleton=1024), mode)
    add(wrong, rb.slice(0, 10))
    wrong.finalize(1)
    with pytest.raises(ka.KmerSpectrumError, match="differing"):
        acc.merge_image(KMR_MAP_WEAK, wrong.image(KMR_MAP_WEAK))


@pytest.mark.parametrize("k,ext", [(31, False), (51, False), (21, True), (95, True)])
def test_merge_add_of_spectra_that_share_kmers(k, ext):
    """KmerMapByKmerArrayPair::mergeAdd (src/Kmer.h:3209-3261) as KmerSpectrum::mergeVector uses it (src/KmerSpectrum.h:2572-2584):
    two spectra of different reads of ONE genome (most k-mers in both) -- the merged weak map equals the oracle's mergeAdd of its
    own two spectra byte for byte: a shared key's count, direction bias and extension tallies are the sums, its weightedCount the
    float sum; keys of one side only are taken over; a third spectrum is merged on top."""
    kw = dict(value_kind=KMR_VALUE_EXT, min_weight=0.0, min_quality_score=2) if ext else {}
    cfg = default_config(k, num_buckets_weak=2048, num_buckets_singleton=4096, **kw)
    rbs = [synth_reads(3000, read_len=150, genome_len=30000, seed=70 + i, quality="noisy", n_rate=0.001) for i in range(3)]
    for rb in rbs[1:]:                       # one genome for all three read sets
        rb.bases[:] = synth_reads(3000, read_len=150, genome_len=30000, seed=70, quality="noisy").bases
    rng = np.random.default_rng(3)
    for i, rb in enumerate(rbs):             # different reads all the same: a rotation of the read order plus fresh errors
        rb.bases[:] = np.roll(rb.bases.reshape(3000, 150), 500 * i, axis=0).reshape(-1)
        hit = rng.random(rb.bases.size) < 0.004 * i
        rb.bases[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(hit.sum()))]
    os_, ps = [], []
    for rb in rbs:
        o = OracleSpectrum(cfg)
        o.add_reads(rb)
        o.finalize(2)
        p = product(cfg, 0)
        add(p, rb)
        p.finalize(2)
        os_.append(o)
        ps.append(p)
    n0 = ps[0].stats()["weak_entries"]
    for j in (1, 2):
        img = ps[j].image(KMR_MAP_WEAK)
        ps[0].merge_image(KMR_MAP_WEAK, img)
        os_[0].merge_add(os_[j])
        io, ip = os_[0].image(KMR_MAP_WEAK), ps[0].image(KMR_MAP_WEAK)
        n = compare_weak_images(io, ip, ps[0].kb, ext)
        assert n == ps[0].stats()["weak_entries"]
    assert n0 < n < n0 + ps[1].stats()["weak_entries"]          # shared keys were added, not appended
    # lookups see the summed counts
    keys, cnt, _, _, _ = os_[0].entries()
    assert np.array_equal(ps[0].getCount(keys[:5000]), cnt[:5000])


def test_merge_add_wraps_like_the_reference():
    """TrackingData::add is `count += other.getCount()` on an unsigned short (src/KmerTrackingData.h:489-493): two maps that hold a
    k-mer 40 000 times each merge to 80 000 - 65 536 = 14 464, and so does the direction bias -- restated as it is, not repaired"""
    k = 25
    cfg = default_config(k, num_buckets_weak=64, num_buckets_singleton=64)
    seq = b"ACGTTGCAAGGCTTAACCGATCGGATTACAGGCATTCGA"          # 39 bases: 15 k-mers per read
    n = 40000
    rb = ReadBatch([seq] * n, [b"I" * len(seq)] * n)
    o1, o2, p1, p2 = OracleSpectrum(cfg), OracleSpectrum(cfg), product(cfg, 0), product(cfg, 0)
    for s in (o1, o2):
        s.add_reads(rb)
        s.finalize(2)
    for s in (p1, p2):
        add(s, rb)
        s.finalize(2)
    p1.merge_image(KMR_MAP_WEAK, p2.image(KMR_MAP_WEAK))
    o1.merge_add(o2)
    keys, cnt, dirb, w, _ = o1.entries()
    assert set(cnt.tolist()) == {2 * n - 65536}
    io, ip = o1.image(KMR_MAP_WEAK), p1.image(KMR_MAP_WEAK)
    assert compare_weak_images(io, ip, p1.kb, False) == 15
    assert np.array_equal(p1.getCount(keys), cnt)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("k", [21, 51])
def test_subtract_reference(k, mode):
    """KmerSpectrum::subtractReference (apps/FilterReads-P.cpp:117): k-mers of the reference spectrum are skipped before
    they count as raw k-mers (append(), src/KmerSpectrum.h:1582-1588); the link ends with optimize()."""
    lib = __import__("helpers").oracle_lib()
    rb = synth_reads(3000, read_len=100, seed=4, quality="noisy")
    ref_reads = rb.slice(0, 400)
    cfg = default_config(k, num_buckets_weak=256, num_buckets_singleton=512)
    o_ref, p_ref = run_both(cfg, ref_reads, min_depth=1, mode=mode)
    o = OracleSpectrum(cfg)
    lib.orc_subtract_reference(o.h, o_ref.h)
    o.add_reads(rb)
    p = product(cfg, mode)
    p.subtractReference(p_ref)
    add(p, rb)
    assert p.getSubtracted() == lib.orc_subtracted(o.h) > 0
    o.finalize(2)
    p.finalize(2)
    assert o.stats() == p.stats()
    compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False)
    ko, cnt, _, _, _ = o_ref.entries()
    assert not p.getCount(ko).any()                 # nothing of the reference is left
    # lookups ignore the link; a finalized spectrum refuses a new one
    with pytest.raises(ka.KmerSpectrumError):
        p.subtractReference(p_ref)


@pytest.mark.parametrize("min_depth,ext", [(1, False), (2, False), (1, True)])
def test_reference_histogram(min_depth, ext):
    """KmerSpectrum::Histogram(256).set(): visits and visitedCount bit-exact per bucket (weak + singleton maps, buckets
    above zoomMax through the reference's log formula), visitedWeight within the weightedCount tolerance; a count of
    600 exercises a log bucket"""
    rb = synth_reads(3000, read_len=100, seed=21, quality="noisy")
    hot = ReadBatch([b"ACGTTGCAAGGCTTAACCGGTATGCATCGAT" + b"G" * 5] * 600, [b"I" * 36] * 600)
    rb = ReadBatch([rb.seq(i) for i in range(rb.n)] + [hot.seq(i) for i in range(hot.n)], [rb.qual(i) for i in range(rb.n)] + [hot.qual(i) for i in range(hot.n)])
    cfg = default_config(31, num_buckets_weak=256, num_buckets_singleton=1024, value_kind=KMR_VALUE_EXT if ext else 0)
    o, p = run_both(cfg, rb, min_depth=min_depth)
    vo, co, wo = o.ref_histogram(256, 2.0)
    h = p.getHistogram(256, 2.0)
    assert np.array_equal(vo, h.visits) and np.array_equal(co, h.visitedCount)
    assert vo[257:].sum() > 0 and (min_depth > 1 or vo[1] > 0)
    assert np.all(np.abs(wo - h.visitedWeight) <= vo * (1.0 / 254) + 1e-5 * co + 1e-9)
    text = h.toString()
    assert text.startswith("Counts, Weights and Directions\nCounts:\t%d\t" % int(vo.sum()))
    assert "\n512\t" in text                                  # getBucketValue of the log bucket that holds count 600
    # other zoom / base
    vo2, co2, _ = o.ref_histogram(15, 1.5)
    h2 = p.getHistogram(15, 1.5)
    assert np.array_equal(vo2, h2.visits) and np.array_equal(co2, h2.visitedCount)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("k", [31, 51])
def test_reads_longer_than_a_tile_are_segmented(k, mode):
    """Reads beyond the 9 952-base LDS tile are cut into segments that start at k-mer indices which are
    multiples of 1024 -- where buildWeightedKmers restarts its weight product (src/KmerReadUtils.h:204) -- so
    weights, and with them the counted k-mers, stay bit-exact."""
    rng = np.random.default_rng(k)
    lens = [30000, 9953, 16000, 200, 9952, 25000, 150, 12345]
    seqs, quals = [], []
    genome = rng.integers(0, 4, 40000)
    qv = np.array([40, 30, 20, 10, 2]) + 33
    for L in lens * 3:
        st = int(rng.integers(0, 40000 - L)) if L < 40000 else 0
        codes = genome[st:st + L].copy()
        errs = rng.random(L) < 0.01
        codes[errs] = (codes[errs] + rng.integers(1, 4, errs.sum())) & 3
        b = np.frombuffer(b"ACGT", dtype=np.uint8)[codes].copy()
        b[rng.random(L) < 0.0005] = ord("N")
        q = qv[rng.choice(5, size=L, p=[0.80, 0.10, 0.05, 0.04, 0.01])].astype(np.uint8)
        seqs.append(b.tobytes())
        quals.append(q.tobytes())
    rb = ReadBatch(seqs, quals)
    cfg = default_config(k, num_buckets_weak=512, num_buckets_singleton=2048)
    o, p = run_both(cfg, rb, mode=mode, batches=[5])
    compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False)
    counts, off = p.getCountsForReads(rb.bases, rb.offsets)
    for i in (0, 2, 5):
        keys, w, ext = oracle_weighted_kmers(cfg, seqs[i], quals[i])
        assert np.array_equal(o.lookup(keys), counts[int(off[i]):int(off[i + 1])])


@pytest.mark.parametrize("mode", MODES)
def test_long_reads_with_extension_values(mode):
    """Segments of a long read take the extension base outside their own span from the read (left neighbour of a
    later segment's first k-mer, right neighbour of an earlier segment's last k-mer): tallies stay bit-exact."""
    rng = np.random.default_rng(5)
    genome = rng.integers(0, 4, 30000)
    seqs, quals = [], []
    qv = np.array([40, 30, 20, 10, 2]) + 33
    for L in (30000, 9953, 18432 + 20, 200, 18432 + 21, 12000) * 2:
        st = int(rng.integers(0, 30000 - L + 1))
        codes = genome[st:st + L].copy()
        errs = rng.random(L) < 0.01
        codes[errs] = (codes[errs] + rng.integers(1, 4, errs.sum())) & 3
        b = np.frombuffer(b"ACGT", dtype=np.uint8)[codes].copy()
        b[rng.random(L) < 0.0005] = ord("N")
        seqs.append(b.tobytes())
        quals.append(qv[rng.choice(5, size=L, p=[0.70, 0.10, 0.10, 0.09, 0.01])].astype(np.uint8).tobytes())
    rb = ReadBatch(seqs, quals)
    cfg = default_config(21, value_kind=KMR_VALUE_EXT, num_buckets_weak=512, num_buckets_singleton=2048, min_weight=0.0, min_quality_score=2)
    o, p = run_both(cfg, rb, min_depth=1, mode=mode)
    compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, True)
    assert np.array_equal(o.image(KMR_MAP_SINGLETON), p.image(KMR_MAP_SINGLETON))


@pytest.mark.parametrize("k,ext,mode", [(31, False, 1), (31, False, 2), (21, True, 1), (21, True, 2), (51, False, 1), (51, False, 2)])
def test_extract_by_owner_and_insert_records(k, ext, mode):
    """The two device halves of the owner exchange (kmr_extract_by_owner_dev ->
    [all-to-all] -> kmr_insert_records_dev) with both 'ranks' on one GPU: each rank's
    spectrum must equal the oracle's spectrum of the k-mers that rank owns."""
    import torch
    world = 2
    rb = synth_reads(3000, read_len=110, seed=33, quality="noisy", n_rate=0.002)
    kw = dict(value_kind=KMR_VALUE_EXT, min_weight=0.0, min_quality_score=2) if ext else {}
    cfgs = [default_config(k, num_buckets_weak=256, num_buckets_singleton=1024, rank=r, world_size=world, **kw) for r in range(world)]
    handles = [product(c, mode) for c in cfgs]
    recb = ka.record_bytes(k, cfgs[0].value_kind)
    dev = torch.device("cuda", 0)
    half = rb.n // 2
    seg_cap = 3000 * 110
    incoming = [[] for _ in range(world)]
    for r, (lo, hi) in enumerate(((0, half), (half, rb.n))):
        part = rb.slice(lo, hi)
        tb = torch.from_numpy(np.concatenate([part.bases, np.zeros(64, np.uint8)])).to(dev)
        tq = torch.from_numpy(np.concatenate([part.quals, np.zeros(64, np.uint8)])).to(dev)
        to = torch.from_numpy(part.offsets.astype(np.int64)).to(dev)
        recs = torch.zeros(world * seg_cap * recb, dtype=torch.uint8, device=dev)
        counts = torch.zeros(world, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        handles[r].extractByOwnerDevice(tb.data_ptr(), tq.data_ptr(), to.data_ptr(), part.n, int(part.offsets[-1]), lo,
                                        recs.data_ptr(), seg_cap, counts.data_ptr())
        handles[r].sync()
        c = counts.cpu().tolist()
        for o in range(world):
            incoming[o].append(recs.view(world, seg_cap * recb)[o, :c[o] * recb].clone())
    for o in range(world):
        allr = torch.cat(incoming[o]).contiguous()
        torch.cuda.synchronize()
        handles[o].insertRecordsDevice(allr.data_ptr(), allr.numel() // recb)
        handles[o].sync()
        handles[o].finalize(2)
        orc = OracleSpectrum(cfgs[o])
        orc.add_reads(rb)
        orc.finalize(2)
        so, sp_ = orc.stats(), handles[o].stats()
        for key in ("raw_good_kmers", "unique_kmers", "singleton_kmers", "weak_entries"):
            assert so[key] == sp_[key], (key, so, sp_)
        # which sighting is 'first' (and loses its direction in the singleton map) depends on arrival order
        # once records travel through the exchange, exactly as in the reference's MPI build
        compare_weak_images(orc.image(KMR_MAP_WEAK), handles[o].image(KMR_MAP_WEAK), handles[o].kb, ext, dir_tol=1, first_tol=1.0 / 254)


def test_many_distinct_keys_force_subpass_split():
    """low coverage, high error: nearly every k-mer is distinct, so final lists overflow the LDS
    table and count_kernel has to split them by further hash bits"""
    rb = synth_reads(30000, read_len=150, genome_len=40_000_000, seed=77, err=0.02)
    cfg = default_config(31, estimated_raw_kmers=1000, num_buckets_weak=4096, num_buckets_singleton=16384)   # estimate far too low on purpose
    o, p = run_both(cfg, rb, min_depth=1, mode=2)
    compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False)
    assert np.array_equal(o.image(KMR_MAP_SINGLETON), p.image(KMR_MAP_SINGLETON))


@pytest.mark.parametrize("k", [31, 51])
def test_one_weight_count_pass_cold_paths(k):
    """The one-weight form of the default build's count pass (one quality character; over two-word keys its LDS table keeps neither
    weight sums nor state words) off its usual path: lists whose distinct k-mers overflow the table and are redone in sub-passes by
    further hash bits (low coverage, 2 % errors, an estimate a third of the truth), lists so long that they are counted in pieces
    through the merge table (an estimate of next to nothing; long_list_chunks turned down on an ordinary input)."""
    sparse = synth_reads(30000, read_len=150, genome_len=40_000_000, seed=78 + k, err=0.02)
    n_kmers = 30000 * (150 - k + 1)
    for est in (n_kmers // 3, 1000):
        cfg = default_config(k, estimated_raw_kmers=est, num_buckets_weak=4096, num_buckets_singleton=16384)
        o, p = run_both(cfg, sparse, min_depth=1, mode=3)
        assert p.build_info("uniform_count") == 1.0
        assert compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False) == o.stats()["weak_entries"]
        assert np.array_equal(o.image(KMR_MAP_SINGLETON), p.image(KMR_MAP_SINGLETON))
    covered = synth_reads(20000, read_len=150, genome_len=120000, seed=k)
    cfg = default_config(k, estimated_raw_kmers=20000 * 40)
    for min_depth in (2, 1):
        o, p = run_both(cfg, covered, min_depth=min_depth, mode=3, long_list_chunks=3)
        assert p.build_info("uniform_count") == 1.0
        assert compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False) == o.stats()["weak_entries"]
        if min_depth == 1:
            assert np.array_equal(o.image(KMR_MAP_SINGLETON), p.image(KMR_MAP_SINGLETON))


@pytest.mark.parametrize("k,nbw,reads", [(31, 1 << 10, 4000), (31, 1 << 16, 30000), (31, 1 << 20, 30000), (51, 1 << 19, 20000), (127, 1 << 12, 6000)])
def test_buckets_by_radix_partition(k, nbw, reads):
    """The weak map of the default build is bucketed by an MSD radix partition over the bucket index (kmr_buckets.hpp) instead of a
    scatter of single entries + per-bucket sort.  Forced on small inputs (kmr_tune binned_buckets_min = 0) in geometries that take one
    level (few buckets), two levels (2^19 / 2^20 buckets: 11 - 12 bits to resolve) and multi-word keys: bucket offsets, key order and
    values must be the serial oracle's; with the partition switched off (-1) the image must be the same bytes."""
    rb = synth_reads(reads, read_len=150, genome_len=reads * 12, seed=321 + k, err=0.01)
    cfg = default_config(k, estimated_raw_kmers=reads * (150 - k + 1), num_buckets_weak=nbw, num_buckets_singleton=nbw)
    o, p = run_both(cfg, rb, min_depth=2, mode=3, binned_buckets_min=0)
    n = compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False)
    assert n > reads
    q = product(cfg, 3, binned_buckets_min=-1)
    add(q, rb)
    q.finalize(2)
    assert np.array_equal(p.image(KMR_MAP_WEAK), q.image(KMR_MAP_WEAK))
    # and with singletons kept (min_depth 1: the singleton map takes the per-bucket path beside it)
    o1, p1 = run_both(cfg, rb, min_depth=1, mode=3, binned_buckets_min=0)
    compare_weak_images(o1.image(KMR_MAP_WEAK), p1.image(KMR_MAP_WEAK), p1.kb, False)
    assert np.array_equal(o1.image(KMR_MAP_SINGLETON), p1.image(KMR_MAP_SINGLETON))


def _uniform_quality_reads(seed, k, quals):
    """reads whose k-mers all weigh the same unless they hold an N: one quality character everywhere ('flat'), REF_QUAL ('ref'), or
    no qualities at all ('none'); N's at 0.3 % of the positions, a few reads longer than an LDS tile, a few shorter than k"""
    rng = np.random.default_rng(seed)
    parts = [synth_reads(6000, read_len=150, genome_len=90000, seed=seed), synth_reads(6, read_len=11000, genome_len=90000, seed=seed + 1),
             synth_reads(40, read_len=k - 1, genome_len=90000, seed=seed + 2), synth_reads(500, read_len=97, genome_len=90000, seed=seed + 3)]
    out = []
    for rb in parts:
        b = rb.bases.copy()
        b[rng.random(b.size) < 0.003] = ord("N")
        q = None if quals == "none" else np.full(b.size, 127 if quals == "ref" else ord("I"), dtype=np.uint8)
        out.append(ReadBatch.from_arrays(b, q, rb.offsets))
    return out


@pytest.mark.parametrize("quals", ["flat", "ref", "none"])
@pytest.mark.parametrize("k", [13, 21, 31, 32, 51, 64, 127])
def test_uniform_weight_extraction(k, quals):
    """sk_extract_lean_kernel takes launches whose k-mers all weigh the same (no qualities, or one quality character): the spectrum
    must be the serial oracle's, and byte for byte what the general kernel makes of the same reads (kmr_tune lean_extract = 0) -- with
    N's in the reads, reads cut into units, reads shorter than k, several calls."""
    batches = _uniform_quality_reads(900 + k, k, quals)
    cfg = default_config(k, estimated_raw_kmers=1_100_000)
    o = OracleSpectrum(cfg)
    p, g, u = product(cfg, 3), product(cfg, 3, lean_extract=0), product(cfg, 3, uniform_count=0)
    first = 0
    for rb in batches:
        o.add_reads(rb, first_idx=first)
        add(p, rb, first=first)
        add(g, rb, first=first)
        add(u, rb, first=first)
        first += rb.n
    for x in (o, p, g, u):
        x.finalize(1)
    # one weight throughout: the count pass's one-weight form (no weight sums in the table) against the general form on the same lists
    # (multi-word keys without pad bits in the last word -- k = 64 -- claim slots through state words, which the one-weight table of
    # multi-word keys does not keep: the general form counts them)
    one_weight_form = 1.0 if (k <= 32 or k % 32 != 0) else 0.0
    assert (p.build_info("uniform_count"), g.build_info("uniform_count"), u.build_info("uniform_count")) == (one_weight_form, 0.0, 0.0)
    assert np.array_equal(p.image(KMR_MAP_WEAK), u.image(KMR_MAP_WEAK)) and np.array_equal(p.image(KMR_MAP_SINGLETON), u.image(KMR_MAP_SINGLETON))
    assert o.stats() == p.stats() == g.stats() == u.stats()
    compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False)
    assert np.array_equal(o.image(KMR_MAP_SINGLETON), p.image(KMR_MAP_SINGLETON))
    assert np.array_equal(p.image(KMR_MAP_WEAK), g.image(KMR_MAP_WEAK))
    assert np.array_equal(p.image(KMR_MAP_SINGLETON), g.image(KMR_MAP_SINGLETON))


@pytest.mark.parametrize("k,win,quality", [(45, 32, "noisy"), (51, 32, "flat"), (51, 16, "noisy"), (51, 8, "flat"), (127, 32, "noisy"), (127, 16, "flat"), (44, 16, "noisy"),
                                           (31, 16, "flat"), (31, 8, "noisy")])
def test_minimizer_windows_at_large_k(k, win, quality):
    """build_mode 3 can take a minimizer window of 32 offsets from k = 45 on (runs of ~16 k-mers: half the records of a window of 16;
    kmr_tune superkmer_window, not the default: DESIGN.md) beside the windows of 16 / 8 / 4: every one of them must give the oracle's
    spectrum -- reads with N's, both extraction kernels (one quality character: the bases-only one; qualities of their own: the
    general one)"""
    rb = synth_reads(5000, read_len=200, genome_len=80000, seed=300 + k + win, quality=quality, n_rate=0.002)
    cfg = default_config(k, estimated_raw_kmers=5000 * (200 - k + 1))
    o = OracleSpectrum(cfg)
    p = product(cfg, 3, superkmer_window=win)
    assert p.build_info("superkmer_window") == win
    o.add_reads(rb)
    add(p, rb)
    o.finalize(1)
    p.finalize(1)
    assert o.stats() == p.stats()
    compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False)
    assert np.array_equal(o.image(KMR_MAP_SINGLETON), p.image(KMR_MAP_SINGLETON))
    # lookups stream over lists of the same geometry
    keys, cnt, _, _, _ = o.entries()
    assert np.array_equal(p.getCount(keys[:2000]), cnt[:2000])


@pytest.mark.parametrize("k,quality,share,n_rate", [(31, "flat", 0.5, 0.001), (31, "noisy", 0.3, 0.001), (51, "noisy", 0.5, 0.001), (31, "flat", 1.0, 0.001), (31, "noisy", 0.0, 0.001),
                                                   (31, "flat", 0.5, 0.0), (51, "flat", 0.4, 0.0)])      # (no N and one quality character: the one-weight count pass, early and late)
def test_lists_counted_early_then_finalize(k, quality, share, n_rate):
    """kmr_count_lists_prefix: the lists below a bound are counted ahead of kmr_finalize (what an owner does with the part of the list
    space that has arrived while the rest is on the wire); kmr_finalize counts the others and takes the early entries over.  Statistics
    and weak image must be those of kmr_finalize alone, byte for byte -- for a bound in the middle, at the end (everything early) and at
    zero; a second early count replaces the first; a min-depth that keeps the singleton map leaves everything to kmr_finalize."""
    rb = synth_reads(40000, read_len=150, genome_len=300000, seed=640 + k, quality=quality, n_rate=n_rate)
    cfg = default_config(k, estimated_raw_kmers=40000 * (150 - k + 1))
    plain, early = product(cfg, 3), product(cfg, 3)
    add(plain, rb)
    add(early, rb)
    nl = int(early.build_info("lists"))
    assert nl > 64
    early.count_lists_prefix(2, nl // 7)                 # replaced by the next one
    early.count_lists_prefix(2, int(nl * share))
    plain.finalize(2)
    early.finalize(2)
    assert early.build_info("early_lists") == int(nl * share) and plain.build_info("early_lists") == 0
    if share > 0:
        assert 0 < early.build_info("early_entries") <= early.stats()["weak_entries"]
    if share == 1.0:
        assert early.build_info("early_entries") == early.stats()["weak_entries"]
    assert plain.stats() == early.stats()
    assert np.array_equal(plain.image(KMR_MAP_WEAK), early.image(KMR_MAP_WEAK))
    assert plain.build_info("uniform_count") == early.build_info("uniform_count") == (1.0 if (quality == "flat" and n_rate == 0.0) else 0.0)
    # with the singleton map kept nothing is counted early; the result is the same all the same
    for sp in (plain, early):
        sp.reset()
        add(sp, rb)
    early.count_lists_prefix(1, nl // 2)
    plain.finalize(1)
    early.finalize(1)
    assert early.build_info("early_lists") == 0
    assert plain.stats() == early.stats()
    assert np.array_equal(plain.image(KMR_MAP_WEAK), early.image(KMR_MAP_WEAK)) and np.array_equal(plain.image(KMR_MAP_SINGLETON), early.image(KMR_MAP_SINGLETON))
    # an early count made for another min-depth is void
    for sp in (plain, early):
        sp.reset()
        add(sp, rb)
    early.count_lists_prefix(3, nl // 2)
    plain.finalize(2)
    early.finalize(2)
    assert early.build_info("early_lists") == 0
    assert plain.stats() == early.stats() and np.array_equal(plain.image(KMR_MAP_WEAK), early.image(KMR_MAP_WEAK))


def test_host_batch_in_pieces_sizes_lists_from_the_whole_call():
    """kmr_add_reads / kmr_add_reads_twobit send a host batch to the device in pieces; without an estimate of the job's k-mers the first
    piece must size the lists (and the chunk pool) for the WHOLE call, as one device call does -- not for itself"""
    rb = synth_reads(60000, read_len=150, genome_len=300000, seed=5, quality="flat")
    cfg = default_config(31)                       # estimated_raw_kmers = 0
    one = product(cfg, 3)
    add(one, rb)
    pieces = product(cfg, 3, twobit_piece_bases=1 << 20)      # nine pieces
    add(pieces, rb)
    assert one.build_info("lists") == pieces.build_info("lists") > 64
    one.finalize(2)
    pieces.finalize(2)
    assert one.stats() == pieces.stats()
    assert np.array_equal(one.image(KMR_MAP_WEAK), pieces.image(KMR_MAP_WEAK))


def test_quality_mix_is_noticed_between_calls():
    """a build whose first call is uniform and whose second is not (and the other way round): every call picks its own kernel"""
    a = _uniform_quality_reads(77, 31, "flat")[0]
    b = synth_reads(5000, read_len=150, genome_len=90000, seed=78, quality="noisy", n_rate=0.002)
    for order in ((a, b), (b, a)):
        cfg = default_config(31, estimated_raw_kmers=1_400_000)
        o = OracleSpectrum(cfg)
        p = product(cfg, 3)
        first = 0
        for rb in order:
            o.add_reads(rb, first_idx=first)
            add(p, rb, first=first)
            first += rb.n
        o.finalize(2)
        p.finalize(2)
        assert p.build_info("uniform_count") == 0.0      # two different weights in the lists: the general count pass
        assert o.stats() == p.stats()
        compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False)


@pytest.mark.parametrize("mode", MODES)
def test_long_reads_within_tile(mode):
    """reads near the per-wavefront LDS tile limit (9 952 bases): one tile then holds few reads but
    hundreds of thousands of k-mers, i.e. many partition batches per extent"""
    rb = synth_reads(300, read_len=5000, genome_len=400000, seed=41, quality="noisy", n_rate=0.001)
    cfg = default_config(31, num_buckets_weak=1024, num_buckets_singleton=4096)
    o, p = run_both(cfg, rb, mode=mode)
    compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False)


@pytest.mark.parametrize("stream", [1, 0])
def test_score_and_trim_reads_golden_labels(stream):
    """f1: ReadSelector::scoreAndTrimReads on the device against the reference's FilterReads golden
    (test/1000-Filtered.fastq: MedianScore / Trim labels of the 949 reads without AFTrim); the k-mer counts come from the
    streaming pass over minimizer lists, or from the per-k-mer probes of the lookup table"""
    import re
    k = 31
    rb = read_fastq(os.path.join(GOLDEN, "1000.fastq"))
    gold = read_fastq(os.path.join(GOLDEN, "1000-Filtered.fastq"))
    p = product(default_config(k, fastq_start_char=64, estimated_raw_kmers=46000), stream_lookups=stream)
    add(p, rb)
    p.finalize(2)
    to, tl, sc, wt = p.scoreAndTrimReads(rb.bases, rb.offsets, 2, "MEDIAN")
    checked = 0
    for i in range(rb.n):
        if b"AFTrim" in gold.names[i]:
            continue
        label = b""
        if wt[i]:
            label += b"Trim:%d+%d " % (to[i], tl[i])
        label += b"MedianScore:%d" % int(sc[i] + 0.5)
        assert label == gold.names[i].split(b" ", 1)[1], (i, label, gold.names[i])
        checked += 1
    assert checked == 949


@pytest.mark.parametrize("scoring,stream", [("MEDIAN", 1), ("AVG", 1), ("MIN", 0), ("MAX", 1), ("SUM", 0), ("MEDIAN", 0)])
def test_score_and_trim_reads_all_types(scoring, stream):
    k = 25
    rb = synth_reads(1500, read_len=120, seed=6, quality="noisy", n_rate=0.004)
    cfg = default_config(k, num_buckets_weak=256, num_buckets_singleton=1024)
    o, p = run_both(cfg, rb, stream_lookups=stream)
    to, tl, sc, wt = p.scoreAndTrimReads(rb.bases, rb.offsets, 3, scoring)
    counts, off = p.getCountsForReads(rb.bases, rb.offsets)
    for i in range(rb.n):
        cnt = counts[int(off[i]):int(off[i + 1])]
        eo, el, es, et = score_and_trim(cnt, rb.seq(i), k, 3, scoring)
        assert (int(to[i]), int(tl[i]), bool(wt[i])) == (eo, el, et), (i, scoring)
        assert abs(float(sc[i]) - es) <= 1e-5 * max(1.0, abs(es)), (i, scoring, sc[i], es)


@pytest.mark.parametrize("pipeline", [False, True])
def test_build_partitioned_driver_single_rank(pipeline):
    """kmernator_amd.distributed.build_partitioned (the N>1 driver of bench.py) with a one-rank RCCL group:
    chunked extract-by-owner -> all-to-all -> insert must equal the direct build, with and without the
    comm/compute pipeline."""
    import torch
    import torch.distributed as dist
    from kmernator_amd.distributed import build_partitioned
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29600 + (os.getpid() % 300) + (1 if pipeline else 0))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n = 400000
        rb = synth_reads(n, read_len=150, genome_len=5 * n, seed=2, quality="noisy")
        tb = torch.from_numpy(np.concatenate([rb.bases, np.zeros(64, np.uint8)])).to(dev)
        tq = torch.from_numpy(np.concatenate([rb.quals, np.zeros(64, np.uint8)])).to(dev)
        to = torch.from_numpy(rb.offsets.astype(np.int64)).to(dev)
        torch.cuda.synchronize()
        a = ka.KmerSpectrum(ka.default_config(31, estimated_raw_kmers=n * 120, device=0))
        a.buildKmerSpectrumDevice(tb.data_ptr(), tq.data_ptr(), to.data_ptr(), n, n * 150, 0)
        a.finalize(2)
        b = ka.KmerSpectrum(ka.default_config(31, estimated_raw_kmers=n * 120, device=0))
        build_partitioned(b, tb, tq, to, chunk_reads=70000, pipeline=pipeline)
        b.finalize(2)
        sa, sb = a.stats(), b.stats()
        for key in ("raw_good_kmers", "unique_kmers", "singleton_kmers", "weak_entries"):
            assert sa[key] == sb[key], (key, sa, sb)
        compare_weak_images(a.image(KMR_MAP_WEAK), b.image(KMR_MAP_WEAK), a.kb, False, dir_tol=1, first_tol=1.0 / 254)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k", [31, 51])
def test_distributed_scoring_halves_three_owners_on_one_gpu(k):
    """f1, distributed form (DistributedReadSelector, src/DistributedFunctions.h:876-1045) without the collectives: three
    handles own a third of the k-mers each; requests binned by owner -> lookup at the owner -> answers scattered back ->
    trim + score must equal scoreAndTrimReads on one whole spectrum."""
    import torch
    world, n = 3, 60000
    rb = synth_reads(n, read_len=120, genome_len=4 * n, seed=9, quality="noisy", n_rate=0.002)
    dev = torch.device("cuda", 0)
    tb = torch.from_numpy(np.concatenate([rb.bases, np.zeros(64, np.uint8)])).to(dev)
    tq = torch.from_numpy(np.concatenate([rb.quals, np.zeros(64, np.uint8)])).to(dev)
    to = torch.from_numpy(rb.offsets.astype(np.int64)).to(dev)
    total = int(rb.offsets[-1])
    whole = ka.KmerSpectrum(ka.default_config(k, estimated_raw_kmers=n * 100, device=0))
    whole.buildKmerSpectrumDevice(tb.data_ptr(), tq.data_ptr(), to.data_ptr(), n, total, 0)
    whole.finalize(2)
    want = whole.scoreAndTrimReads(rb.bases, rb.offsets, 2, "MEDIAN")
    owners = []
    for r in range(world):
        s = ka.KmerSpectrum(ka.default_config(k, estimated_raw_kmers=n * 100, device=0, rank=r, world_size=world))
        s.buildKmerSpectrumDevice(tb.data_ptr(), tq.data_ptr(), to.data_ptr(), n, total, 0)
        s.finalize(2)
        owners.append(s)
    assert sum(s.stats()["weak_entries"] for s in owners) == whole.stats()["weak_entries"]
    words = (whole.kb + 7) // 8
    seg_cap = n * 120 // 2
    keys = torch.empty((world, seg_cap, words), dtype=torch.int64, device=dev)
    pos = torch.empty((world, seg_cap), dtype=torch.int32, device=dev)
    cnt = torch.zeros(world, dtype=torch.int64, device=dev)
    position_counts = torch.zeros(total, dtype=torch.int32, device=dev)
    req = owners[1]                                   # the requesting rank; two calls, as the chunks of score_partitioned
    half = n // 2 + 13
    n_req = 0
    for lo, hi in ((0, half), (half, n)):
        torch.cuda.synchronize()
        req.lookup_requests(tb, to, lo, hi, int(rb.offsets[hi] - rb.offsets[lo]), keys, pos, seg_cap, cnt)
        req.sync()
        sc = [int(x) for x in cnt.cpu().tolist()]
        n_req += sum(sc)
        assert min(sc) > 0.2 * sum(sc)
        lib = ka.load()
        hk = keys[2, :50].cpu().numpy().view(np.uint64).astype(">u8").view(np.uint8).reshape(50, 8 * words)[:, :whole.kb]
        for kk in hk:                                  # what sits in owner 2's segment is owned by rank 2
            assert lib.kmr_distributed_thread_id(lib.kmr_hash(kk.tobytes(), whole.kb), world) == 2
        for s in range(world):
            ans = torch.zeros(sc[s], dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            owners[s].lookup_keys(keys[s], sc[s], ans)
            owners[s].sync()
            req.scatter_counts(ans, pos[s], sc[s], position_counts)
            req.sync()
    no_n = sum(max(0, len(rb.seq(i)) - k + 1) for i in range(n) if b"N" not in rb.seq(i))
    assert no_n <= n_req <= n * (120 - k + 1)
    got = req.score_counts(tb, to, n, position_counts, 2, "MEDIAN")
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


def test_score_partitioned_driver_single_rank():
    """kmernator_amd.distributed.score_partitioned with a one-rank RCCL group (keys and answers through the all-to-alls, several
    chunks) equals scoreAndTrimReads"""
    import torch
    import torch.distributed as dist
    from kmernator_amd.distributed import score_partitioned
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29950 + (os.getpid() % 300))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n = 200000
        rb = synth_reads(n, read_len=150, genome_len=5 * n, seed=4, quality="noisy", n_rate=0.001)
        tb = torch.from_numpy(np.concatenate([rb.bases, np.zeros(64, np.uint8)])).to(dev)
        tq = torch.from_numpy(np.concatenate([rb.quals, np.zeros(64, np.uint8)])).to(dev)
        to = torch.from_numpy(rb.offsets.astype(np.int64)).to(dev)
        sp = ka.KmerSpectrum(ka.default_config(31, estimated_raw_kmers=n * 120, device=0))
        sp.buildKmerSpectrumDevice(tb.data_ptr(), tq.data_ptr(), to.data_ptr(), n, n * 150, 0)
        sp.finalize(2)
        want = sp.scoreAndTrimReads(rb.bases, rb.offsets, 2, "AVG")
        got = score_partitioned(sp, tb, to, 2, "AVG", chunk_reads=70000)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", [2, 3])
def test_ext_hot_kmer_takes_the_wide_tally_table(mode):
    """extension values: the count pass keeps 16-bit tallies for lists of up to 65 535 records and sends longer ones through
    the 32-bit table in a second launch -- a k-mer that occurs 130 000 times (homopolymer reads) must come out with its
    exact tallies (the reference's are u32, only `count` saturates at 65 535), next to ordinary lists.  build_mode 3: the long list is
    cut into pieces of at most SK_EXT_LONG_CHUNKS chunks whose tables (16-bit tallies) are added into a device table with 32-bit
    ones; its saturated keys keep the reference's weightedCount / directionBias (first 65 535 sightings)"""
    k = 21
    rng = np.random.default_rng(17)
    base = synth_reads(4000, read_len=100, seed=8, quality="noisy", n_rate=0.001)
    seqs = [base.seq(i) for i in range(base.n)]
    quals = [base.qual(i) for i in range(base.n)]
    for i in range(1700):                              # 1700 x 80 = 136 000 occurrences of A^21 (and of T^21's canonical form)
        s = (b"A" if i % 3 else b"T") * 100
        q = bytes(rng.choice(np.frombuffer(b"5:?DI#", dtype=np.uint8), size=100).tobytes())
        at = int(rng.integers(0, len(seqs)))
        seqs.insert(at, s)
        quals.insert(at, q)
    rb = ReadBatch(seqs, quals)
    cfg = default_config(k, value_kind=KMR_VALUE_EXT, min_weight=0.0, min_quality_score=2, num_buckets_weak=256, num_buckets_singleton=1024)
    o, p = run_both(cfg, rb, min_depth=2, mode=mode)
    n = compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, True, saturated_dir_free=(mode != 3))
    assert n > 1000
    hot = np.zeros((1, p.kb), dtype=np.uint8)          # A^21 packed
    assert p.getCount(hot)[0] == 65535 == o.lookup(hot)[0]


@pytest.mark.parametrize("k", [31, 51])
def test_million_noisy_reads_against_the_oracle(k):
    """1 M reads x 100 bp with noisy qualities (the discard path and the divide chain are live) against the SERIAL
    oracle: statistics, the weak image byte for byte in keys, counts, direction biases, weightedCount within 1e-5 * count --
    an order of magnitude above the other oracle comparisons, to catch offset arithmetic and seams that only show at size."""
    n, rl = 1_000_000, 100
    rb = synth_reads(n, read_len=rl, genome_len=3 * n, seed=100 + k, quality="noisy", n_rate=0.0005)
    cfg = default_config(k, estimated_raw_kmers=n * (rl - k + 1))
    o, p = run_both(cfg, rb, min_depth=2, mode=0)
    nkept = compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False)
    assert nkept == o.stats()["weak_entries"] and nkept > 1_000_000
    o.close()


@pytest.mark.parametrize("k", [31, 51])
def test_ragged_reads_at_scale_against_the_oracle(k):
    """150 000 reads of every length from 0 to 400 bases (empty reads, reads shorter than k, reads that fill several LDS tiles' worth of
    a wavefront unevenly), noisy qualities, 0.3 % N: the default build against the SERIAL oracle -- statistics, singleton image byte
    for byte, weak image in keys, counts and direction biases; the fixed-length synthetic reads of the other at-scale tests never
    put a short read next to a long one inside a tile"""
    n = 150_000
    full = synth_reads(n, read_len=400, genome_len=2_000_000, seed=500 + k, quality="noisy", n_rate=0.003)
    rng = np.random.default_rng(9)
    lens = rng.integers(0, 401, n)
    lens[::97] = 0
    lens[1::89] = k - 1
    lens[2::83] = k
    seqs = [full.bases[i * 400:i * 400 + int(lens[i])].tobytes() for i in range(n)]
    quals = [full.quals[i * 400:i * 400 + int(lens[i])].tobytes() for i in range(n)]
    rb = ReadBatch(seqs, quals)
    cfg = default_config(k, estimated_raw_kmers=int(np.maximum(lens - k + 1, 0).sum()))
    o, p = run_both(cfg, rb, min_depth=1, mode=0)
    assert compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False) == o.stats()["weak_entries"] > 100_000
    assert np.array_equal(o.image(KMR_MAP_SINGLETON), p.image(KMR_MAP_SINGLETON))
    o.close()


@pytest.mark.parametrize("k,mode,ext", [(21, 2, False), (31, 3, False), (51, 3, False), (95, 1, False), (127, 3, False), (21, 2, True), (31, 1, True)])
def test_streaming_lookups_equal_table_probes(k, mode, ext):
    """f1 as a streaming pass (reads -> super-k-mers -> minimizer lists, the weak map's entries grouped by the same lists, answers from
    LDS) against the per-k-mer probes: same trims and scores, for reads with N's, reads shorter than k, reads longer than an LDS tile,
    spectra built in any mode (a handle that never made lists sizes them for the batch) and a second batch against the same map."""
    rb = synth_reads(4000, read_len=180, genome_len=60000, seed=k, quality="noisy", n_rate=0.003)
    odd = synth_reads(12, read_len=12000, genome_len=60000, seed=k + 1, quality="noisy", n_rate=0.001)
    short = synth_reads(50, read_len=k - 1, genome_len=60000, seed=k + 2)
    cfg = default_config(k, estimated_raw_kmers=4000 * 180, **(dict(value_kind=KMR_VALUE_EXT) if ext else {}))      # extension values: the count heads the 60-byte value too
    a, b = product(cfg, mode, stream_lookups=1, long_list_chunks=(4 if k == 51 else 1024)), product(cfg, mode, stream_lookups=0)      # k = 51: most lists answered in pieces
    for p in (a, b):
        add(p, rb)
        add(p, odd, first=rb.n)
        p.finalize(2)
    for batch in (rb, odd, short, rb.slice(100, 900)):
        for scoring in ("MEDIAN", "SUM"):
            ra = a.scoreAndTrimReads(batch.bases, batch.offsets, 2, scoring)
            rb_ = b.scoreAndTrimReads(batch.bases, batch.offsets, 2, scoring)
            for x, y in zip(ra, rb_):
                assert np.array_equal(x, y)
    assert a.stats() == b.stats()          # the lookup pass leaves the build's counters alone


@pytest.mark.parametrize("mode", MODES)
def test_low_complexity_reads(mode):
    """homopolymers and dinucleotide repeats: a handful of k-mers seen 10^5-10^6 times (counts saturate at 65 535), one list / one
    table slot / one owner taking nearly everything -- the appends to ONE super-k-mer list from every lane of the chip at once must
    neither starve nor give up"""
    rb = synth_reads(24000, read_len=150, genome_len=20000, seed=77, quality="noisy")
    bases = rb.bases.copy().reshape(rb.n, 150)
    quals = rb.quals.copy().reshape(rb.n, 150)
    bases[0::3] = ord("A")
    bases[1::3, 0::2] = ord("A")
    bases[1::3, 1::2] = ord("C")
    quals[0::3] = ord("I")
    quals[1::3] = ord("5")
    rb = type(rb).from_arrays(bases.reshape(-1), quals.reshape(-1), rb.offsets)
    cfg = default_config(31, estimated_raw_kmers=24000 * 120)
    o, p = run_both(cfg, rb, mode=mode)
    # the default build redoes its saturated keys from their first 65 535 sightings in input order (sat_*_kernel): directionBias exact,
    # weightedCount within the usual tolerance.  The two other modes have no order to go by: those entries are exempt there.
    assert compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False, saturated_dir_free=(mode != 3)) == o.stats()["weak_entries"]
    if mode == 3:
        _, _, buckets = parse_image(p.image(KMR_MAP_WEAK), p.kb, 12)
        sat = sum(int(((np.ascontiguousarray(v).view(np.uint32).reshape(len(v), 3)[:, 0] & 0xffff) == 65535).sum()) for _, v in buckets if len(v))
        assert sat >= 3          # poly-A / poly-T and the two phases of the AC repeat
        # without a singleton map (no first sighting to set aside) and with the lists counted in pieces (merge table path)
        for kw, tune in ((dict(separate_singletons=0), dict()), (dict(), dict(long_list_chunks=8))):
            o2, p2 = run_both(default_config(31, estimated_raw_kmers=24000 * 120, **kw), rb, mode=3, **tune)
            assert compare_weak_images(o2.image(KMR_MAP_WEAK), p2.image(KMR_MAP_WEAK), p2.kb, False) == o2.stats()["weak_entries"]


@pytest.mark.parametrize("k,ext,kw,tune", [(31, False, {}, {}), (31, False, dict(separate_singletons=0), {}), (51, False, {}, dict(long_list_chunks=8)),
                                           (21, True, dict(value_kind=KMR_VALUE_EXT, min_weight=0.0, min_quality_score=2), {})])
def test_high_count_kmers_keep_the_reference_order(k, ext, kw, tune):
    """weightedCount of k-mers seen 10^3 ... 6 x 10^4 times (unsaturated), noisy weights: the reference adds the weights one after
    the other in arrival order into a float (weightedCount += weight, src/KmerTrackingData.h:427-448), which at these counts drifts
    from the exact sum by far more than 1e-5 * count; the default build redoes every k-mer seen SK_ORDERED_FROM = 256 times or more
    from its sightings in input order (sat_*_kernel) and must give the SERIAL oracle's value BIT FOR BIT -- with and without a
    singleton map (the first sighting comes back quantised, or not), with lists counted in pieces, with extension values.
    Three repeat families (400 bases under 4 000 and 40 000 reads, 150 bases under 60 000 reads of 100 bases) make ~900 keys at
    counts from several hundred to ~45 000."""
    rng = np.random.default_rng(1234 + k)
    L = 100
    seqs, quals = [], []
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    qv = np.frombuffer(b"I?5+#", dtype=np.uint8)
    for flen, cov_reads in ((400, 4000), (400, 40000), (150, 60000)):
        fam = lut[rng.integers(0, 4, flen)]
        for _ in range(cov_reads):
            st = int(rng.integers(0, flen - L + 1))
            r = fam[st:st + L]
            if rng.random() < 0.5:
                r = (lut[3 - np.searchsorted(lut, r[::-1])])
            seqs.append(r.tobytes())
            quals.append(qv[rng.choice(5, L, p=[0.80, 0.10, 0.05, 0.04, 0.01])].tobytes())
    order = rng.permutation(len(seqs))
    rb = ReadBatch([seqs[i] for i in order], [quals[i] for i in order])
    cfg = default_config(k, estimated_raw_kmers=rb.n * (L - k + 1), **kw)
    o, p = run_both(cfg, rb, min_depth=2, mode=0, **tune)
    keys, cnt, _, _, _ = o.entries()
    assert ((cnt >= 2000) & (cnt < 65535)).sum() >= 100 and (cnt >= 256).sum() >= 500
    assert cnt.max() > 20000
    n = compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, ext, exact_from=256)
    assert n == o.stats()["weak_entries"]


@pytest.mark.parametrize("k,chunks", [(31, 2), (51, 3), (27, 16)])
def test_long_lists_counted_in_pieces(k, chunks):
    """build_mode 3: a list of more than `long_list_chunks` chunks is counted by several blocks, each over a range of its chunks,
    whose tables are merged in a device hash table before entries are made.  With the threshold turned down to a few chunks nearly
    every list of an ordinary input goes that way: the maps must still be the oracle's (the weight sum is formed in another order:
    the usual tolerance), with and without a singleton map."""
    rb = synth_reads(20000, read_len=150, genome_len=120000, seed=k, quality="noisy", n_rate=0.002)
    for kw in (dict(), dict(separate_singletons=0)):
        cfg = default_config(k, estimated_raw_kmers=20000 * 40, **kw)          # few lists, so that they are long
        o, p = run_both(cfg, rb, mode=3, long_list_chunks=chunks)
        assert compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False) == o.stats()["weak_entries"]
    cfg = default_config(k, estimated_raw_kmers=20000 * 40)
    o, p = run_both(cfg, rb, min_depth=1, mode=3, long_list_chunks=chunks)      # singleton map kept
    assert np.array_equal(o.image(KMR_MAP_SINGLETON), p.image(KMR_MAP_SINGLETON))
    # extension values: the pieces' 16-bit tallies and packets go through the merge table's 32-bit ones
    cfg = default_config(k, estimated_raw_kmers=20000 * 40, value_kind=KMR_VALUE_EXT, min_weight=0.0, min_quality_score=2)
    o, p = run_both(cfg, rb, min_depth=1, mode=3, long_list_chunks=chunks)
    assert compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, True) == o.stats()["weak_entries"]
    assert np.array_equal(o.image(KMR_MAP_SINGLETON), p.image(KMR_MAP_SINGLETON))


def _pack_twobit(rb):
    """the reads of a ReadBatch as TwoBitSequence::compressSequence leaves them (src/TwoBitSequence.cpp:242-269): per read ceil(L / 4) bytes,
    first base in bits 7-6, anything but ACGT packed as A and written down as a markup ('.' as 'N')"""
    code = np.full(256, 4, dtype=np.uint8)
    for c, v in zip(b"ACGTacgt", (0, 1, 2, 3, 0, 1, 2, 3)):
        code[c] = v
    tw, to, mp, mc, mo = [], [0], [], [], [0]
    for i in range(rb.n):
        a, b = int(rb.offsets[i]), int(rb.offsets[i + 1])
        cs = code[rb.bases[a:b]]
        for j in np.nonzero(cs == 4)[0]:
            mp.append(int(j)); mc.append(ord("N") if rb.bases[a + j] == ord(".") else int(rb.bases[a + j]))
        cs = np.where(cs == 4, 0, cs).astype(np.uint8)
        pad = (-len(cs)) % 4
        cs = np.concatenate([cs, np.zeros(pad, np.uint8)]).reshape(-1, 4)
        tw.append((cs[:, 0] << 6 | cs[:, 1] << 4 | cs[:, 2] << 2 | cs[:, 3]).astype(np.uint8))
        to.append(to[-1] + len(tw[-1])); mo.append(len(mp))
    return (np.concatenate(tw) if tw else np.zeros(0, np.uint8), np.array(to, np.uint64), (np.array(mp, np.uint32), np.array(mc, np.uint8), np.array(mo, np.uint64)))


@pytest.mark.parametrize("mode", [3, 2])
@pytest.mark.parametrize("quals", ["array", "uniform", "none"])
def test_twobit_feed_equals_ascii_feed(mode, quals):
    """kmr_add_reads_twobit: reads handed over as the reference's Read keeps them (2-bit packed + markups, src/Sequence.h:166-171) must build
    the spectrum kmr_add_reads builds from their text -- ragged lengths (0, 1, k - 1, k, 4 m + 1 ...), N's and lower case, qualities as an
    array / one character for every base / none, fed in two calls; and that spectrum is the oracle's"""
    k = 31
    rng = np.random.default_rng(5)
    rb0 = synth_reads(3000, read_len=150, genome_len=30000, seed=44, quality="noisy" if quals == "array" else "flat", n_rate=0.003)
    seqs, qs = [], []
    for i in range(rb0.n):
        L = int(rng.choice([150, 149, 147, 121, 33, 31, 30, 5, 1, 0], p=[0.5, 0.1, 0.1, 0.1, 0.05, 0.05, 0.04, 0.03, 0.02, 0.01]))
        s = bytes(rb0.seq(i)[:L]); q = bytes(rb0.qual(i)[:L])
        if i % 17 == 0: s = s.lower()
        seqs.append(s); qs.append(q)
    rb = ReadBatch(seqs, qs if quals == "array" else ([b"I" * len(s) for s in seqs] if quals == "uniform" else None))
    cfg = default_config(k, estimated_raw_kmers=3000 * 120)
    o = OracleSpectrum(cfg); o.add_reads(rb); o.finalize(1)
    pa = product(cfg, mode); add(pa, rb); pa.finalize(1)
    pt = product(cfg, mode)
    half = rb.n // 2
    for part, first in ((rb.slice(0, half), 0), (rb.slice(half, rb.n), half)):
        tw, to, mk = _pack_twobit(part)
        pt.buildKmerSpectrumTwoBit(tw, to, part.offsets, quals=part.quals if quals == "array" else None, uniform_quality=ord("I") if quals == "uniform" else 0,
                                   markups=mk, first_read_idx=first)
    pt.finalize(1)
    assert pa.stats() == pt.stats() == o.stats()
    for which in (KMR_MAP_WEAK, KMR_MAP_SINGLETON):
        assert np.array_equal(pa.image(which), pt.image(which))
    assert compare_weak_images(o.image(KMR_MAP_WEAK), pt.image(KMR_MAP_WEAK), pt.kb, False) == o.stats()["weak_entries"]


def test_read_batch_from_packed_reads():
    """kmr_reads_from_twobit: the device batch made from packed reads holds the reads' text again (upper case, markups back in place) and
    the qualities given -- an array, one character, or Read::REF_QUAL -- and builds the spectrum of the text"""
    rb = synth_reads(1500, read_len=97, genome_len=12000, seed=3, quality="noisy", n_rate=0.01)
    tw, to, mk = _pack_twobit(rb)
    cfg = default_config(31, estimated_raw_kmers=1500 * 67)
    sp = product(cfg, 3)
    for quals, uq, want_q in ((rb.quals, 0, rb.quals), (None, ord("5"), np.full(rb.bases.size, ord("5"), np.uint8)), (None, 0, np.full(rb.bases.size, 127, np.uint8))):
        rs = ka.ReadSet.from_twobit(sp, tw, to, rb.offsets, quals=quals, uniform_quality=uq, markups=mk)
        b, q, o, _ = rs.arrays()
        assert np.array_equal(b, rb.bases) and np.array_equal(q, want_q) and np.array_equal(o, rb.offsets)
        rs.close()
    rs = ka.ReadSet.from_twobit(sp, tw, to, rb.offsets, quals=rb.quals, markups=mk)
    sp.buildKmerSpectrumFromReadSet(rs); sp.finalize(1)
    pa = product(cfg, 3); add(pa, rb); pa.finalize(1)
    assert pa.stats() == sp.stats() and np.array_equal(pa.image(KMR_MAP_WEAK), sp.image(KMR_MAP_WEAK))


def test_twobit_device_feed_without_byte_offsets():
    """kmr_add_reads_twobit_dev with device arrays: twobit offsets left out (every read starts on the byte behind the one before it: the
    library scans ceil(L / 4) itself), markups and a discarded read, two calls whose offsets do not start at zero -- the spectrum of
    kmr_add_reads on the same reads"""
    torch = pytest.importorskip("torch")
    k = 31
    rb = synth_reads(2500, read_len=101, genome_len=20000, seed=12, quality="noisy", n_rate=0.004)
    disc = np.zeros(rb.n, np.uint8); disc[7] = 1
    rb.discarded = disc
    cfg = default_config(k, estimated_raw_kmers=2500 * 71)
    pa = product(cfg, 3); add(pa, rb); pa.finalize(1)
    tw, to, (mp, mc, mo) = _pack_twobit(rb)
    dev = torch.device("cuda", 0)
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a).view(dt) if dt else np.ascontiguousarray(a)).to(dev)
    d_tw, d_off, d_mo, d_mp, d_mc, d_q, d_d = t(tw), t(rb.offsets, np.int64), t(mo, np.int64), t(mp, np.int32), t(mc), t(rb.quals), t(disc)
    pt = product(cfg, 3)
    half = 1200
    for r0, r1 in ((0, half), (half, rb.n)):
        b0, b1 = int(rb.offsets[r0]), int(rb.offsets[r1])
        pt.buildKmerSpectrumTwoBitDevice(d_tw.data_ptr() + int(to[r0]), None, d_off.data_ptr() + 8 * r0, r1 - r0, b1 - b0, quals_ptr=d_q.data_ptr() + b0,
                                         markup_offsets_ptr=d_mo.data_ptr() + 8 * r0, markup_pos_ptr=d_mp.data_ptr(), markup_char_ptr=d_mc.data_ptr(),
                                         first_read_idx=r0, discarded_ptr=d_d.data_ptr() + r0)
    pt.sync(); pt.finalize(1)
    assert pa.stats() == pt.stats()
    for which in (KMR_MAP_WEAK, KMR_MAP_SINGLETON):
        assert np.array_equal(pa.image(which), pt.image(which))


@pytest.mark.parametrize("k,ext", [(31, False), (21, False), (51, False), (21, True)])
def test_list_counts_that_are_not_powers_of_two(k, ext):
    """build_mode 3 on one GPU sizes its lists for the count pass (sk_list_of's code above 32: the count itself, multiply-shift) instead of
    taking a power of two: the maps must not know -- held to the power-of-two lists (tune pow2_lists), to two other list lengths
    (tune list_aim, one of them short enough to be many lists, one long enough to overflow tables) and to the oracle"""
    rb = synth_reads(40000, read_len=150, genome_len=150000, seed=300 + k, quality="noisy", n_rate=0.002)
    kw = dict(value_kind=KMR_VALUE_EXT, min_weight=0.0, min_quality_score=2) if ext else {}
    cfg = default_config(k, estimated_raw_kmers=40000 * (150 - k + 1), **kw)
    o = OracleSpectrum(cfg); o.add_reads(rb); o.finalize(2)
    ref = None
    for tune in (dict(pow2_lists=1), dict(), dict(list_aim=333), dict(list_aim=5000)):
        p = product(cfg, 3, **tune); add(p, rb); p.finalize(2)
        assert p.stats() == o.stats(), tune
        img = (p.image(KMR_MAP_WEAK), p.image(KMR_MAP_SINGLETON))
        if ref is None:
            ref = img
            assert compare_weak_images(o.image(KMR_MAP_WEAK), img[0], p.kb, ext) == o.stats()["weak_entries"]
        else:
            assert np.array_equal(ref[0], img[0]) and np.array_equal(ref[1], img[1]), tune


@pytest.mark.parametrize("k", [21, 31, 51])
@pytest.mark.parametrize("uq", [ord("I"), 0])
def test_twobit_packed_bytes_staged_directly(k, uq):
    """kmr_add_reads_twobit_dev with one quality character (or none) on the lists: sk_extract_lean_kernel<.., PACKED> stages the packed bytes
    as they are -- no unpacked copy.  Held to the unpack-first path (tune packed_direct = 0) and to kmr_add_reads on the text: ragged reads,
    N markups, a markup that names a base, a discarded read, reads of 9 800 (between the two tile spans) and 25 000 bases (cut into
    units, with N's inside), the reads' bytes with gaps between them and in the opposite order of the reads, two calls whose offsets
    do not start at zero.  Window of 8 (k = 21) and 16, one- and two-word keys."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(100 + k)
    rb0 = synth_reads(2000, read_len=150, genome_len=20000, seed=50 + k, quality="flat", n_rate=0.004)
    seqs = []
    for i in range(rb0.n):
        L = int(rng.choice([150, 149, 147, 121, k + 2, k, k - 1, 5, 1, 0], p=[0.5, 0.1, 0.1, 0.1, 0.05, 0.05, 0.04, 0.03, 0.02, 0.01]))
        seqs.append(bytes(rb0.seq(i)[:L]))
    for L in (9800, 25000):
        g = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)].copy()
        g[rng.integers(0, L, 12)] = ord("N")
        seqs.insert(len(seqs) // 3, g.tobytes())
        seqs.append(g[::-1].tobytes())
    rb = ReadBatch(seqs, None if uq == 0 else [bytes([uq]) * len(s) for s in seqs])
    disc = np.zeros(rb.n, np.uint8); disc[11] = 1
    rb.discarded = disc
    cfg = default_config(k, estimated_raw_kmers=int(rb.bases.size))
    pa = product(cfg, 3); add(pa, rb); pa.finalize(1)
    tw, to, (mp, mc, mo) = _pack_twobit(rb)
    # a markup that names a base: the packed bits say A, the markup says what the text holds (applyMarkup writes the character back)
    mp, mc, mo = list(mp), list(mc), [int(x) for x in mo]
    done = 0
    for i in range(rb.n):
        a, L = int(rb.offsets[i]), int(rb.offsets[i + 1] - rb.offsets[i])
        if L < 40 or done >= 20 or mo[i + 1] != mo[i]:
            continue
        j = 17
        if rb.bases[a + j] in b"CGT":
            byte, sh = int(to[i]) + j // 4, 6 - 2 * (j % 4)
            tw[byte] &= ~(3 << sh) & 0xff
            at = mo[i]
            mp.insert(at, j); mc.insert(at, int(rb.bases[a + j]))
            for r in range(i + 1, rb.n + 1):
                mo[r] += 1
            done += 1
    assert done == 20
    mp, mc, mo = np.array(mp, np.uint32), np.array(mc, np.uint8), np.array(mo, np.uint64)
    # the reads' bytes in the opposite order of the reads, seven bytes of something else between them
    to2 = np.zeros(rb.n + 1, np.uint64)
    tw2 = np.full(int(tw.size) + 7 * rb.n + 64, 0xa7, np.uint8)
    at = 13
    for i in range(rb.n - 1, -1, -1):
        nb = int(to[i + 1] - to[i])
        tw2[at:at + nb] = tw[int(to[i]):int(to[i + 1])]
        to2[i] = at
        at += nb + 7
    dev = torch.device("cuda", 0)
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a).view(dt) if dt else np.ascontiguousarray(a)).to(dev)
    d_off, d_mo, d_mp, d_mc, d_d = t(rb.offsets, np.int64), t(mo, np.int64), t(mp, np.int32), t(mc), t(disc)
    images = []
    for layout, direct in (("packed", 1), ("packed", 0), ("scattered", 1)):
        d_tw = t(tw if layout == "packed" else tw2)
        d_to = None if layout == "packed" else t(to2, np.int64)
        pt = product(cfg, 3, packed_direct=direct)
        half = rb.n // 2 + 1
        for r0, r1 in ((0, half), (half, rb.n)):
            b0, b1 = int(rb.offsets[r0]), int(rb.offsets[r1])
            pt.buildKmerSpectrumTwoBitDevice(d_tw.data_ptr() + (int(to[r0]) if d_to is None else 0), None if d_to is None else d_to.data_ptr() + 8 * r0,
                                             d_off.data_ptr() + 8 * r0, r1 - r0, b1 - b0, uniform_quality=uq,
                                             markup_offsets_ptr=d_mo.data_ptr() + 8 * r0, markup_pos_ptr=d_mp.data_ptr(), markup_char_ptr=d_mc.data_ptr(),
                                             first_read_idx=r0, discarded_ptr=d_d.data_ptr() + r0)
        pt.sync(); pt.finalize(1)
        assert pa.stats() == pt.stats(), (layout, direct)
        for which in (KMR_MAP_WEAK, KMR_MAP_SINGLETON):
            assert np.array_equal(pa.image(which), pt.image(which)), (layout, direct, which)


@pytest.mark.parametrize("mode", [3, 2])
def test_build_score_reset_build_again(mode):
    """the streaming lookups borrow the handle's list pool and list state after kmr_finalize: a kmr_reset and a second build on the same
    handle must not see anything of them (and a second scoring call reuses the list index of the map)"""
    k = 31
    rb1 = synth_reads(6000, read_len=150, genome_len=50000, seed=91, quality="noisy", n_rate=0.002)
    rb2 = synth_reads(5000, read_len=120, genome_len=40000, seed=92, quality="noisy", n_rate=0.002)
    cfg = default_config(k, estimated_raw_kmers=6000 * 120)
    p = product(cfg, mode)
    for rb in (rb1, rb2, rb1):
        o = OracleSpectrum(cfg)
        o.add_reads(rb)
        o.finalize(2)
        p.reset()
        add(p, rb)
        p.finalize(2)
        assert o.stats() == p.stats()
        assert compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), p.kb, False) == o.stats()["weak_entries"]
        a = p.scoreAndTrimReads(rb.bases, rb.offsets, 2, "MEDIAN")
        b = p.scoreAndTrimReads(rb2.bases, rb2.offsets, 2, "AVG")
        counts, off = p.getCountsForReads(rb.bases, rb.offsets)
        for i in range(0, rb.n, 37):
            eo, el, es, et = score_and_trim(counts[int(off[i]):int(off[i + 1])], rb.seq(i), k, 2, "MEDIAN")
            assert (int(a[0][i]), int(a[1][i]), bool(a[3][i])) == (eo, el, et) and abs(float(a[2][i]) - es) <= 1e-5 * max(1.0, abs(es))
        assert len(b[0]) == rb2.n
