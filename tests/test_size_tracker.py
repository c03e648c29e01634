"""f3: KmerSpectrum::SizeTracker (src/KmerSpectrum.h:812-900, trackSpectrum :1574-1581, the --size-history-file of FilterReads).

The reference samples its four counters before every k-mer it appends, and so does the product (round 3: the thresholds are located
at the exact k-mer inside a read; round 2 applied the rule at read ends).  The oracle keeps both histories.  No fixture of the
reference holds a size history (its test scripts do not write one), so what pins the restatement is the source text; what is
tested: the rule's arithmetic (thresholds 128, 134, 140 ... as `long *= 1.05` truncates), that the per-read history is the
per-k-mer one sampled at most one read later, and -- on the GPU -- that the product's history equals the oracle's PER-K-MER
history, the reference's own, element for element (with sub-sampling, without a singleton map, over several calls, long reads)."""
import numpy as np
import pytest

from helpers import OracleSpectrum, default_config, read_fastq, synth_reads, GOLDEN
import os


def thresholds(n):
    t, out = 128, []
    for _ in range(n):
        out.append(t)
        t = int(t * 1.05)
    return out


def test_oracle_histories_follow_the_rule():
    rb = synth_reads(4000, read_len=100, genome_len=40000, seed=9, quality="noisy")
    o = OracleSpectrum(default_config(25, num_buckets_weak=256, num_buckets_singleton=512))
    o.add_reads(rb)
    per_kmer = o.size_tracker(per_read=False, force_last=False)
    per_read = o.size_tracker(per_read=True, force_last=False)
    st = o.stats()
    # before every k-mer: rawKmers grows by one per call, so element i is taken exactly at the i-th threshold
    th = thresholds(len(per_kmer))
    assert len(per_kmer) > 150 and np.array_equal(per_kmer[:, 0], np.array(th, dtype=np.uint64))
    # after every read: the first read end at or behind the threshold -- less than one read's k-mers later -- and never two
    # elements for one read
    assert len(per_read) <= len(per_kmer)
    th = thresholds(len(per_read))
    assert np.all(per_read[:, 0] >= np.array(th, dtype=np.uint64))
    assert np.all(np.diff(per_read[:, 0].astype(np.int64)) > 0)
    # while the thresholds are closer together than one read's k-mers (76 here: up to ~1500 raw k-mers) every read end makes an
    # element and the history runs behind; once they are further apart an element is less than one read late
    late = per_read[:, 0].astype(np.int64) - np.array(th, dtype=np.int64)
    assert late[-60:].max() < 100 - 25 + 1 and late.min() >= 0
    # monotone counters, singletons never above uniques, and the forced last element is the spectrum's totals
    for el in (per_kmer, per_read):
        assert np.all(np.diff(el[:, 1].astype(np.int64)) >= 0) and np.all(np.diff(el[:, 2].astype(np.int64)) >= 0)
        assert np.all(el[:, 3] <= el[:, 2]) and np.all(el[:, 1] <= el[:, 0])
    last = o.size_tracker(per_read=True, force_last=True)[-1]
    assert tuple(int(v) for v in last) == (st["raw_kmers"], st["raw_good_kmers"], st["unique_kmers"], st["singleton_kmers"])


def test_world2_reduce_repeats_the_last_element(tmp_path):
    """reduceSizeTracker (src/DistributedFunctions.h:460-491) over gloo"""
    import torch.distributed as dist
    import torch.multiprocessing as mp
    mp.spawn(_reduce_worker, args=(2, 29650 + os.getpid() % 300, str(tmp_path)), nprocs=2, join=True)
    a, b = (np.load(os.path.join(str(tmp_path), "red.%d.npy" % r)) for r in range(2))
    want = np.array([[11, 7, 5, 3], [22, 14, 10, 6], [33 + 20, 21 + 10, 15 + 5, 9 + 1]], dtype=np.uint64)
    want[0] += np.array([10, 5, 3, 1], dtype=np.uint64)
    want[1] += np.array([20, 10, 5, 1], dtype=np.uint64)
    assert np.array_equal(a, want) and np.array_equal(b, want)


def _reduce_worker(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kmernator_amd.distributed import reduce_size_tracker
        from kmernator_amd.spectrum import SizeTracker
        mine = [[11, 7, 5, 3], [22, 14, 10, 6], [33, 21, 15, 9]] if rank == 0 else [[10, 5, 3, 1], [20, 10, 5, 1]]
        red = reduce_size_tracker(SizeTracker(np.array(mine, dtype=np.uint64)))
        np.save(os.path.join(tmp, "red.%d.npy" % rank), red.elements)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("k,sub,sing", [(31, 1, 1), (51, 1, 1), (25, 3, 1), (31, 1, 0)])
def test_product_history_equals_the_oracles_per_kmer_history(k, sub, sing):
    import kmernator_amd as ka
    rb = synth_reads(30000, read_len=150, genome_len=200000, seed=k, quality="noisy", n_rate=0.002)
    kw = dict(estimated_raw_kmers=30000 * (150 - k + 1), kmer_subsample=sub, separate_singletons=sing)
    o = OracleSpectrum(default_config(k, **kw))
    p = ka.KmerSpectrum(ka.default_config(k, size_tracker=1, **kw))
    cuts = [0, 7000, 7001, 19000, 30000]          # several kmr_add_reads calls, one of a single read
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        part = rb.slice(lo, hi)
        o.add_reads(part, first_idx=lo)
        p.buildKmerSpectrum(part.bases, part.quals, part.offsets, first_read_idx=lo)
    o.finalize(2)
    p.finalize(2)
    for force in (False, True):
        want = o.size_tracker(per_read=False, force_last=force)
        got = p.getSizeTracker(force_last=force).elements
        assert got.shape == want.shape and len(want) > 150
        assert np.array_equal(got, want), np.argwhere(got != want)[:5]
    text = p.getSizeTracker().toString().splitlines()
    assert text[0] == "rawKmers\trawGoodKmers\tuniqueKmers\tsingletonKmers" and len(text) == len(want) + 1
    assert text[-1] == "\t".join(str(int(v)) for v in want[-1])


@pytest.mark.gpu
def test_long_reads_and_the_phix_fixture():
    """reads longer than an LDS tile are cut into units whose records add up; the reference's own 1000.fastq"""
    import kmernator_amd as ka
    long_rb = synth_reads(40, read_len=15000, genome_len=100000, seed=2, quality="noisy")
    fx = read_fastq(os.path.join(GOLDEN, "1000.fastq"))
    # (buckets sized for the input: the oracle re-sorts a bucket on every new key, as the reference does)
    for rb, kw in ((long_rb, dict(k=31, estimated_raw_kmers=600000)), (fx, dict(k=31, fastq_start_char=64, estimated_raw_kmers=120000))):
        k = kw.pop("k")
        o = OracleSpectrum(default_config(k, **kw))
        p = ka.KmerSpectrum(ka.default_config(k, size_tracker=1, **kw))
        o.add_reads(rb)
        p.buildKmerSpectrum(rb.bases, rb.quals, rb.offsets)
        o.finalize(2)
        p.finalize(2)
        assert np.array_equal(p.getSizeTracker().elements, o.size_tracker(per_read=False))


@pytest.mark.gpu
def test_refused_where_it_is_not_kept():
    import kmernator_amd as ka
    from helpers import KMR_VALUE_EXT
    for kw in (dict(build_mode=2), dict(value_kind=KMR_VALUE_EXT), dict(rank=0, world_size=2, build_mode=3)):
        with pytest.raises(ka.KmerSpectrumError):
            ka.KmerSpectrum(ka.default_config(31, size_tracker=1, **kw))
    p = ka.KmerSpectrum(ka.default_config(31))
    rb = synth_reads(100, read_len=80, seed=1)
    p.buildKmerSpectrum(rb.bases, rb.quals, rb.offsets)
    p.finalize(2)
    with pytest.raises(ka.KmerSpectrumError):
        p.getSizeTracker()
