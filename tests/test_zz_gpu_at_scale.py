"""At-scale and full-size parity of the HIP build, run LAST (the file name sorts behind every other test file: a memory-hungry
case must not stand between `pytest -x` and the small oracle comparisons).

What these tests compare with is the ORACLE, not a second build on the device: tests/golden/full_size_digests.json holds what
the serial oracle (oracle/kmr_oracle.cpp) makes of the same reads -- SURVEY.md 8(d)'s generator gives the same bytes on the CPU
(orc_synth_reads) and on the device (kmr_synth_reads_dev) -- as statistics and as the order-independent map digest of
include/kmernator_amd.h (kmr_map_digest); tests/golden/make_full_size_digests.py wrote it, part by part, in the build container.
Integer fields (keys, counts, direction biases, extension tallies, singleton bytes) are held bit for bit through the digest's
hash; weightedCount is a float accumulation whose order the reference itself does not fix (src/KmerTrackingData.h:427-448), so
its sum over the map is held to WEIGHT_REL.  The build modes are still compared with each other where that is cheap."""
import ctypes as C

import numpy as np
import pytest

import kmernator_amd as ka
from helpers import KMR_MAP_SINGLETON, KMR_MAP_WEAK, add_digests, digests_agree, full_size_golden, synth_reads

pytestmark = pytest.mark.gpu

MODES = [1, 2, 3]
# the sum of weightedCount over a map: the oracle adds floats in arrival order, the product rounds an exact sum once per entry;
# measured differences are below 1e-7 of the sum
WEIGHT_REL = 1e-6


@pytest.fixture(autouse=True)
def _free_memory_note(request, capsys):
    """free / total device memory at the start of every at-scale test goes into the test's captured output (shown with a failure)"""
    import torch
    fr, tot = torch.cuda.mem_get_info(0)
    print("[%s] device memory at start: %.1f GB free of %.1f GB" % (request.node.name, fr / 1e9, tot / 1e9))
    yield


def _golden_reads(name, dev=0):
    """(golden entry, bases, quals, offsets, n, read_len) with the reads of that entry generated on the device"""
    import torch
    g = full_size_golden(name)
    c = g["config"]
    b, q, o = ka.synth_reads_device(torch, c["seed"], 0, c["reads"], c["read_len"], c["genome"], c["noisy"], torch.device("cuda", dev))
    return g, b, q, o, c["reads"], c["read_len"]


def _build(g, b, q, o, mode=0, **kw):
    c = g["config"]
    per = c["read_len"] - c["k"] + 1
    cfg = dict(c.get("cfg", {}))
    cfg.update(kw)
    p = ka.KmerSpectrum(ka.default_config(c["k"], estimated_raw_kmers=c["reads"] * per, device=0, build_mode=mode, **cfg))
    p.buildKmerSpectrumDevice(b.data_ptr(), q.data_ptr(), o.data_ptr(), c["reads"], c["reads"] * c["read_len"], 0)
    p.finalize(c["min_depth"])
    return p


def _assert_oracle(p, g):
    """statistics and map digests of a finalized product spectrum == the oracle's"""
    st = p.stats()
    assert st == g["stats"], (st, g["stats"])
    d = p.digest(KMR_MAP_WEAK)
    assert digests_agree(d, g["weak_digest"], WEIGHT_REL), (d, g["weak_digest"])
    if g["config"]["min_depth"] == 1 and g.get("singleton_digest"):
        ds = p.digest(KMR_MAP_SINGLETON)
        assert digests_agree(ds, g["singleton_digest"], 1e-9), (ds, g["singleton_digest"])


def _image_digest(sp, which=KMR_MAP_WEAK):
    import hashlib
    img = sp.image(which)
    return img.size, hashlib.blake2b(memoryview(img), digest_size=16).hexdigest()


def _bookkeeping(p, st):
    hist = p.histogram(4096)[0]
    assert int(hist.sum()) == st["weak_entries"]
    assert int((hist * np.arange(hist.size, dtype=np.uint64)).sum()) + st["singleton_kmers"] == st["raw_good_kmers"]
    assert st["unique_kmers"] == st["weak_entries"] + st["singleton_kmers"]


def test_generator_on_the_device_equals_the_cpu_statement():
    """kmr_synth_reads_dev == orc_synth_reads byte for byte: the start of C2, a slice of C3's last reads (global read indices near
    10^8, genome positions near 5e8), C4's seed, other read lengths, both quality modes"""
    import torch
    from helpers import synth_reads_8d
    dev = torch.device("cuda", 0)
    for seed, first, n, L, G, noisy in [(1, 0, 200_000, 150, 50_000_000, True), (2, 99_900_000, 100_000, 150, 500_000_000, False),
                                         (3, 49_990_000, 10_000, 150, 250_000_000, True), (9, 5, 3000, 76, 1000, True), (9, 0, 1, 31, 31, False),
                                         (10, 0, 2000, 1001, 100_000, True)]:
        b, q, o = ka.synth_reads_device(torch, seed, first, n, L, G, noisy, dev)
        rb = synth_reads_8d(seed, first, n, L, G, noisy)
        assert np.array_equal(b[:n * L].cpu().numpy(), rb.bases), (seed, first)
        assert np.array_equal(q[:n * L].cpu().numpy(), rb.quals), (seed, first)
        assert np.array_equal(o.cpu().numpy().astype(np.uint64), rb.offsets)


@pytest.mark.parametrize("name", ["small_k31_noisy", "small_k51_flat"])
@pytest.mark.parametrize("mode", MODES)
def test_small_golden_digests_every_mode(name, mode):
    """the digest plumbing at a size the CPU suite rebuilds too (tests/test_full_size_digests.py): every build mode against the
    committed oracle digest, and kmr_map_digest against the numpy restatement over the image bytes"""
    from helpers import digest_of_image
    g, b, q, o, n, L = _golden_reads(name)
    p = _build(g, b, q, o, mode)
    _assert_oracle(p, g)
    assert digests_agree(p.digest(KMR_MAP_WEAK), digest_of_image(p.image(KMR_MAP_WEAK), p.kb), 1e-9)
    p.close()


def test_build_modes_agree_at_scale():
    """3M reads (360M k-mers, two sub-batches, ~2.6e5 final lists): the device-table path and the streaming
    partition path are independent algorithms; their statistics and weak images must be byte-identical
    (this caught a list hand-off race that only showed above ~1e8 k-mers)."""
    n = 3000000
    rb = synth_reads(n, read_len=150, genome_len=5 * n, seed=1)
    res = []
    for mode in MODES:
        c = ka.default_config(31, estimated_raw_kmers=n * 120, build_mode=mode)
        p = ka.KmerSpectrum(c)
        p.buildKmerSpectrum(rb.bases, rb.quals, rb.offsets, 0, rb.discarded)
        p.finalize(2)
        res.append((p.stats(), p.image(KMR_MAP_WEAK), p.histogram(1024)[0]))
        p.close()
    for other in res[1:]:                # all three build modes, the default one (3) included
        assert res[0][0] == other[0]
        assert np.array_equal(res[0][1], other[1])
        assert np.array_equal(res[0][2], other[2])
    st, _, hist = res[2]
    # size-independent bookkeeping: every good occurrence is in exactly one entry
    assert int(hist.sum()) == st["weak_entries"]
    assert int((hist * np.arange(hist.size, dtype=np.uint64)).sum()) + st["singleton_kmers"] == st["raw_good_kmers"]
    assert st["unique_kmers"] == st["weak_entries"] + st["singleton_kmers"]


@pytest.mark.parametrize("k", [31, 127])
def test_mostly_distinct_kmers_both_modes_agree(k):
    """Low coverage (most k-mers seen once): the streaming path sizes its final lists from the measured share of
    distinct keys (distinct_probe_kernel) instead of overflowing the count pass's LDS table into sub-passes; the
    result must not depend on that choice -- byte-identical weak image and statistics against the table path."""
    n = 400000
    rb = synth_reads(n, read_len=150, genome_len=60 * n, seed=11)      # ~2.5x coverage
    res = []
    for mode in MODES:
        c = ka.default_config(k, estimated_raw_kmers=n * (150 - k + 1), build_mode=mode)
        p = ka.KmerSpectrum(c)
        p.buildKmerSpectrum(rb.bases, rb.quals, rb.offsets, 0, rb.discarded)
        p.finalize(2)
        res.append((p.stats(), p.image(KMR_MAP_WEAK)))
        p.close()
    assert len(res) == len(MODES) == 3
    for other in res[1:]:                # every mode against the device-table path
        assert res[0][0] == other[0]
        assert np.array_equal(res[0][1], other[1])
    assert res[0][0]["unique_kmers"] > 0.6 * res[0][0]["raw_good_kmers"]


@pytest.mark.parametrize("name", ["k51_noisy_3m", "k64_flat_3m", "k96_flat_3m", "k96_noisy_3m", "k127_flat_3m", "k127_noisy_3m"])
def test_multiword_keys_at_scale_against_the_oracle(name):
    """two-, three- and four-word keys at 3 M reads x 150 bp of a 30x genome (up to 3e8 k-mers), flat qualities and qualities of
    their own: the default build against the serial oracle's statistics and digest; the device-table build (build_mode 1, an
    independent algorithm) against the same."""
    g, b, q, o, n, L = _golden_reads(name)
    for mode in (0, 1):
        p = _build(g, b, q, o, mode)
        _assert_oracle(p, g)
        assert p.stats()["weak_entries"] > 100_000
        p.close()


@pytest.mark.parametrize("name", ["sing_k31_d1", "sing_k31_d3", "sing_k31_d1_one_map", "sing_k51_d1_noisy"])
def test_singleton_maps_at_scale_against_the_oracle(name):
    """3 M reads with the singleton map kept (min-depth 1) or purged below 3, with and without a separate singleton map, flat and
    noisy qualities: statistics, weak digest and singleton digest (1-byte values: the first sighting's quantised weight, which
    depends on which sighting was the first) against the serial oracle, for the default build and the device table."""
    g, b, q, o, n, L = _golden_reads(name)
    for mode in (0, 1):
        p = _build(g, b, q, o, mode)
        _assert_oracle(p, g)
        assert p.stats()["unique_kmers"] > 1_000_000
        p.close()


@pytest.mark.parametrize("name", ["ext_k21_5m", "ext_k21_noisy_2m"])
def test_extension_values_at_scale_against_the_oracle(name):
    """MeraculousCounter's settings (BASELINE.json configs[4]: k = 21, extension values, min quality 2, no weight floor) at 5 M
    synthetic reads = 6.5e8 k-mers (flat qualities: every extension counts) and at 2 M reads with noisy qualities (the quality >= 20
    rule of trackExtension decides, singleton packets kept): the default build (extension records on the super-k-mer lists) and the
    device-table build against the serial oracle -- the digest folds all twelve tallies of every 60-byte value."""
    g, b, q, o, n, L = _golden_reads(name)
    imgs = []
    for mode in (0, 1):
        p = _build(g, b, q, o, mode)
        _assert_oracle(p, g)
        imgs.append(_image_digest(p))
        p.close()
    if not g["config"]["noisy"]:
        assert imgs[0] == imgs[1]          # flat qualities: byte for byte, weightedCount included


@pytest.mark.parametrize("name,world", [("xchg_k31_4m", 2), ("xchg_k31_8m", 4), ("xchg_k51_4m", 2)])
def test_list_exchange_at_scale_on_one_gpu(name, world):
    """The N > 1 build of the default mode at a size where a rank packs and adopts millions of chunks (the two-rank tests on one GPU
    use 6e4 reads): `world` handles on one GPU, 2 M reads each (consecutive slices of one job), every rank extracts into the job's
    lists with global stream ordinals, packs what the others own (kmr_sk_exchange_counts / _pack_dev), the segments are handed over
    in device memory as the all-to-all would deliver them, every owner adopts and finalizes.  The owners' maps must partition the
    SERIAL ORACLE's spectrum of the whole job: statistics add up and the owners' digests add up to the oracle's."""
    import torch
    g = full_size_golden(name)
    c = g["config"]
    k, L = c["k"], c["read_len"]
    n = c["reads"] // world
    dev = torch.device("cuda", 0)
    reads = [ka.synth_reads_device(torch, c["seed"], r * n, n, L, c["genome"], c["noisy"], dev) for r in range(world)]
    per = L - k + 1
    hs, packed = [], []
    for r in range(world):
        h = ka.KmerSpectrum(ka.default_config(k, estimated_raw_kmers=n * per * world, device=0, rank=r, world_size=world, build_mode=3))
        b, q, o = reads[r]
        h.sk_exchange_begin()
        h.set_stream_origin(r * n * L)
        h.buildKmerSpectrumDevice(b.data_ptr(), q.data_ptr(), o.data_ptr(), n, n * L, r * n)
        chunks, granules = h.sk_exchange_counts()
        sc = [int(x) if j != r else 0 for j, x in enumerate(chunks)]
        sg = [int(x) if j != r else 0 for j, x in enumerate(granules)]
        goff = [int(x) for x in np.concatenate([[0], np.cumsum(sg)[:-1]])]
        coff = [int(x) for x in np.concatenate([[0], np.cumsum(sc)[:-1]])]
        data = torch.empty((max(sum(sg), 1), 4), dtype=torch.int32, device=dev)
        meta = torch.empty((max(sum(sc), 1), 2), dtype=torch.int32, device=dev)
        h.sk_exchange_pack(data.data_ptr(), meta.data_ptr(), goff, coff)
        assert sum(sc) > 100_000                                              # the size this test is about
        hs.append(h)
        packed.append((data, meta, sc, sg, goff, coff))
    for owner in range(world):
        for r in range(world):
            if r == owner:
                continue
            data, meta, sc, sg, goff, coff = packed[r]
            if sc[owner]:
                if owner == 0:      # one owner is told by the senders (as the drivers do), the others look at what they receive
                    hs[owner].sk_exchange_peer_uniform(hs[r].sk_exchange_uniform())
                hs[owner].sk_exchange_adopt(data[goff[owner]:].data_ptr(), meta[coff[owner]:].data_ptr(), sc[owner], sg[owner])
        hs[owner].sync()
    tot = {"unique_kmers": 0, "weak_entries": 0, "singleton_kmers": 0}
    dig = None
    for h in hs:
        h.finalize(2)
        assert h.build_info("uniform_count") == 1.0      # one weight on every rank: the owners count with the one-weight form, told or not
        st = h.stats()
        for key in tot:
            tot[key] += st[key]
        dig = add_digests(dig, h.digest(KMR_MAP_WEAK))
        h.close()
    del packed
    assert {key: g["stats"][key] for key in tot} == tot
    assert digests_agree(dig, g["weak_digest"], WEIGHT_REL), (dig, g["weak_digest"])


@pytest.mark.parametrize("name,world", [("xchg_k31_4m", 2), ("xchg_k31_8m", 4)])
def test_list_exchange_in_two_steps_with_an_early_count(name, world):
    """The exchange in two steps over the list space (kmr_sk_exchange_range): the lower half of the lists travels and is adopted, every
    owner counts it (kmr_count_lists_prefix) -- on real devices while the upper half is on the wire --, then the upper half travels
    and kmr_finalize counts it and takes the early entries over.  `world` handles on one GPU, segments handed over in device memory:
    the owners' statistics and digests must add up to the SERIAL ORACLE's of the whole job, as in the one-step exchange."""
    import torch
    g = full_size_golden(name)
    c = g["config"]
    k, L = c["k"], c["read_len"]
    n = c["reads"] // world
    dev = torch.device("cuda", 0)
    per = L - k + 1
    hs = []
    for r in range(world):
        h = ka.KmerSpectrum(ka.default_config(k, estimated_raw_kmers=n * per * world, device=0, rank=r, world_size=world, build_mode=3))
        b, q, o = ka.synth_reads_device(torch, c["seed"], r * n, n, L, c["genome"], c["noisy"], dev)
        h.sk_exchange_begin()
        h.set_stream_origin(r * n * L)
        h.buildKmerSpectrumDevice(b.data_ptr(), q.data_ptr(), o.data_ptr(), n, n * L, r * n)
        h.sync()
        del b, q, o
        hs.append(h)
    nl = int(hs[0].build_info("lists"))
    mid = nl // 2
    for lo, hi in ((0, mid), (mid, nl)):
        packed = []
        for r, h in enumerate(hs):
            h.sk_exchange_range(lo, hi)
            chunks, granules = h.sk_exchange_counts()
            sc = [int(x) if j != r else 0 for j, x in enumerate(chunks)]
            sg = [int(x) if j != r else 0 for j, x in enumerate(granules)]
            goff = [int(x) for x in np.concatenate([[0], np.cumsum(sg)[:-1]])]
            coff = [int(x) for x in np.concatenate([[0], np.cumsum(sc)[:-1]])]
            data = torch.empty((max(sum(sg), 1), 4), dtype=torch.int32, device=dev)
            meta = torch.empty((max(sum(sc), 1), 2), dtype=torch.int32, device=dev)
            h.sk_exchange_pack(data.data_ptr(), meta.data_ptr(), goff, coff)
            assert sum(sc) > 10_000
            packed.append((data, meta, sc, sg, goff, coff))
        for owner in range(world):
            for r in range(world):
                if r == owner:
                    continue
                data, meta, sc, sg, goff, coff = packed[r]
                if sc[owner]:
                    hs[owner].sk_exchange_peer_uniform(hs[r].sk_exchange_uniform())
                    hs[owner].sk_exchange_adopt(data[goff[owner]:].data_ptr(), meta[coff[owner]:].data_ptr(), sc[owner], sg[owner])
            if hi == mid:
                hs[owner].count_lists_prefix(2, mid)          # the lower half is complete on this owner
        for h in hs:
            h.sync()
        del packed
    tot = {"unique_kmers": 0, "weak_entries": 0, "singleton_kmers": 0}
    dig = None
    for h in hs:
        h.finalize(2)
        assert h.build_info("early_lists") == mid and h.build_info("early_entries") > 100_000      # the lower half's entries came from the early count
        st = h.stats()
        for key in tot:
            tot[key] += st[key]
        dig = add_digests(dig, h.digest(KMR_MAP_WEAK))
        h.close()
    assert {key: g["stats"][key] for key in tot} == tot
    assert digests_agree(dig, g["weak_digest"], WEIGHT_REL), (dig, g["weak_digest"])


def test_c2_full_size_against_the_oracle():
    """BASELINE.json configs[1] at full size (10M reads x 150 bp, k=31, 1.2e9 k-mers; SURVEY 8(d): seed 1, 50 Mbp genome) through
    the device-pointer entry point bench.py times.  Statistics and the weak map's digest equal the SERIAL ORACLE's (keys, counts and
    direction biases bit for bit); counts conserve the k-mers, the image is sorted and bucket-consistent (sampled), lookups of
    k-mers taken from the image return their counts, and a second build is bit-identical."""
    g, bases, quals, offsets, n, rl = _golden_reads("c2_flat")
    c = ka.default_config(31, estimated_raw_kmers=n * 120, device=0)
    p = ka.KmerSpectrum(c)
    imgs = []
    for rep in range(2):
        p.reset()
        p.buildKmerSpectrumDevice(bases.data_ptr(), quals.data_ptr(), offsets.data_ptr(), n, n * 150, 0)
        p.finalize(2)
        st = p.stats()
        assert st["raw_kmers"] == n * 120 == st["raw_good_kmers"]
        _bookkeeping(p, st)
        _assert_oracle(p, g)
        imgs.append(p.image(KMR_MAP_WEAK))
    assert np.array_equal(imgs[0], imgs[1])          # deterministic, including the f32 weight sums
    img = imgs[0]
    nb = int(np.frombuffer(img[:8].tobytes(), dtype=np.uint64)[0])
    assert nb == 1 << 21                              # reference sizing for 1.2e9 raw k-mers
    offs = np.frombuffer(img[16:16 + 8 * nb].tobytes(), dtype=np.uint64)
    lib = ka.load()
    rng = np.random.default_rng(5)
    checked = 0
    for b in rng.integers(0, nb, 300):
        o = int(offs[b])
        cnt = int(np.frombuffer(img[o:o + 4].tobytes(), dtype=np.uint32)[0])
        keys = img[o + 4:o + 4 + 8 * cnt].reshape(cnt, 8)
        vals = img[o + 4 + 8 * cnt:o + 4 + 20 * cnt].reshape(cnt, 12)
        prev = None
        for kk in keys:
            kb_ = kk.tobytes()
            assert lib.kmr_hash(kb_, 8) & (nb - 1) == b
            assert prev is None or prev < kb_
            prev = kb_
        if cnt:
            counts = np.ascontiguousarray(vals[:, :2]).view(np.uint16).reshape(-1)
            assert np.array_equal(p.getCount(keys), counts.astype(np.uint32))
            assert counts.min() >= 2
            checked += cnt
    assert checked > 1000
    p.close()


def test_c2_full_size_noisy_qualities_against_the_oracle():
    """BASELINE.json configs[1] at full size with qualities of their own (bench.py --quality noisy: the general extraction with the
    fp64 weight chain, the 0.10 weight floor discarding k-mers, 9-byte records): statistics (raw / good / discarded / unique /
    singleton counts) and the weak digest against the serial oracle; the device-table build (build_mode 1 with a table sized from
    the measured distinct count) against the same."""
    g, bases, quals, offsets, n, rl = _golden_reads("c2_noisy")
    for mode in (0, 1):
        p = _build(g, bases, quals, offsets, mode, **({"max_table_entries": g["stats"]["unique_kmers"]} if mode == 1 else {}))
        st = p.stats()
        assert st["raw_kmers"] == n * 120 and 0 < st["raw_good_kmers"] < st["raw_kmers"]      # the weight floor discards some
        _bookkeeping(p, st)
        _assert_oracle(p, g)
        assert st["weak_entries"] > 10_000_000
        p.close()


def test_c2_full_size_packed_feed_in_pieces():
    """The PCIe-inclusive leg of bench.py at full size (configs[1]): the batch handed over 2-bit packed as the reference's Read
    keeps it (TwoBitSequence::compressSequence, one quality character for all bases), in four calls, staged by the extraction as
    it lies -- statistics and weak digest must equal the oracle's of the text feed (the same stream ordinals, hence the same first
    sightings)."""
    import torch
    g, bases, quals, offsets, n, rl = _golden_reads("c2_flat")
    pieces = 4
    dev = torch.device("cuda", 0)
    PB = (rl + 3) // 4
    db = torch.empty(n * PB + 64, dtype=torch.uint8, device=dev)
    for lo in range(0, n, 1 << 20):
        m = min(1 << 20, n - lo)
        c = bases[lo * rl:(lo + m) * rl].view(m, rl)
        c = ((c >> 1) & 3) ^ ((c >> 2) & 1)                                # A C G T -> 0 1 2 3
        c = torch.nn.functional.pad(c, (0, PB * 4 - rl)).view(m, PB, 4)
        db[lo * PB:(lo + m) * PB] = (c[:, :, 0] << 6 | c[:, :, 1] << 4 | c[:, :, 2] << 2 | c[:, :, 3]).reshape(-1)
    tb_off = torch.arange(n + 1, device=dev, dtype=torch.int64) * PB
    torch.cuda.synchronize()
    p = ka.KmerSpectrum(ka.default_config(31, estimated_raw_kmers=n * 120, device=0))
    per = (n + pieces - 1) // pieces
    for c in range(pieces):
        r0, r1 = c * per, min(n, (c + 1) * per)
        p.buildKmerSpectrumTwoBitDevice(db.data_ptr(), tb_off.data_ptr() + 8 * r0, offsets.data_ptr() + 8 * r0, r1 - r0, (r1 - r0) * rl,
                                        quals_ptr=None, uniform_quality=33 + 40, first_read_idx=r0)
    p.finalize(2)
    _assert_oracle(p, g)
    p.close()


def test_c4_full_size_k51_against_the_oracle():
    """BASELINE.json configs[3] exactly as SURVEY 8(d) defines it: k = 51 (two-word keys), 50 M synthetic 150 bp reads of a 250 Mbp
    genome, seed 3 = 5e9 k-mers over 7.5e9 input bases -- more than 2^32, so the stream ordinal that decides which sighting of a
    k-mer was its first (directionBias, the quantised first weight) has to be wider than 32 bits.  The default build (super-k-mer
    lists) must give the SERIAL ORACLE's statistics and weak digest (2.2e9 distinct k-mers: keys, counts and direction biases bit
    for bit), conserve the k-mers and be consistent with its own histogram and lookups; build_mode 2's 16-byte records carry 32
    ordinal bits and must refuse the input instead of being quietly wrong."""
    g, bases, quals, offsets, n, rl = _golden_reads("c4_flat")
    k = 51
    per = rl - k + 1
    p = ka.KmerSpectrum(ka.default_config(k, estimated_raw_kmers=n * per, device=0))
    p.buildKmerSpectrumDevice(bases.data_ptr(), quals.data_ptr(), offsets.data_ptr(), n, n * rl, 0)
    p.finalize(2)
    st = p.stats()
    assert st["raw_kmers"] == n * per == st["raw_good_kmers"]
    _bookkeeping(p, st)
    _assert_oracle(p, g)
    # lookups of k-mers cut out of the reads
    host = bases[:200 * rl].cpu().numpy().tobytes()
    lib = ka.load()
    keys = np.zeros((200, p.kb), dtype=np.uint8)
    for r in range(200):
        packed = np.zeros(p.kb, dtype=np.uint8)
        lib.kmr_compress_sequence(host[r * rl + 7:r * rl + 7 + k], k, packed.ctypes.data_as(C.POINTER(C.c_uint8)), None, None, 0)
        lib.kmr_least_complement(packed.ctypes.data_as(C.POINTER(C.c_uint8)), k, keys[r].ctypes.data_as(C.POINTER(C.c_uint8)))
    got = p.getCount(keys)
    assert (got >= 1).sum() > 100 and got.max() < 200      # ~20x coverage of a random genome
    p.close()
    del p
    p = ka.KmerSpectrum(ka.default_config(k, estimated_raw_kmers=n * per, device=0, build_mode=2))
    with pytest.raises(ka.KmerSpectrumError, match="32-bit stream ordinal"):
        p.buildKmerSpectrumDevice(bases.data_ptr(), quals.data_ptr(), offsets.data_ptr(), n, n * rl, 0)
    p.close()


def test_device_memory_returns_after_destroy():
    """create / build / finalize / destroy in every build mode hands all device memory back (round 3 leaked the packed entry
    buffers of the default build: 2-6 GB per C2-size handle)"""
    import torch
    g, bases, quals, offsets, n, rl = _golden_reads("k51_noisy_3m")
    torch.cuda.synchronize()
    lost = []
    for mode in (0, 1, 2, 3, 0):
        torch.cuda.empty_cache()
        before = torch.cuda.mem_get_info(0)[0]
        p = _build(g, bases, quals, offsets, mode)
        p.getCount(np.zeros((4, p.kb), dtype=np.uint8))
        p.image(KMR_MAP_WEAK)
        p.close()
        del p
        torch.cuda.synchronize()
        lost.append(before - torch.cuda.mem_get_info(0)[0])
    assert max(lost) < (64 << 20), lost          # a few MB of runtime bookkeeping at most


def test_config3_whole_input_on_one_gpu_against_the_oracle():
    """BASELINE.json configs[2]'s INPUT -- 100 M synthetic 150 bp reads of a 500 Mbp genome, seed 2, k = 31: 1.2e10 k-mers, what
    `bench.py --gpus 8` spreads over eight GPUs -- built on ONE MI355X in eight calls of 12.5 M reads (each rank's batch generated, fed
    and dropped in turn; ~150 GB of lists): statistics and weak-map digest (3.6e9 distinct k-mers) equal the SERIAL ORACLE's.  The
    8-GPU run itself is the driver's; its bench line checks the ranks' summed digests against this same golden entry.  Runs last."""
    import torch
    try:
        g = full_size_golden("c3_flat")
    except KeyError:
        pytest.skip("no oracle digest of config 3 committed (tests/golden/make_full_size_digests.py c3_flat: two hours of the build container)")
    c = g["config"]
    world, n, L, k = 8, c["reads"] // 8, c["read_len"], c["k"]
    dev = torch.device("cuda", 0)
    sp = ka.KmerSpectrum(ka.default_config(k, estimated_raw_kmers=c["reads"] * (L - k + 1), device=0))
    for r in range(world):
        b, q, o = ka.synth_reads_device(torch, c["seed"], r * n, n, L, c["genome"], c["noisy"], dev)
        sp.buildKmerSpectrumDevice(b.data_ptr(), q.data_ptr(), o.data_ptr(), n, n * L, r * n)
        sp.sync()
        del b, q, o
    sp.finalize(c["min_depth"])
    _assert_oracle(sp, g)
    sp.close()
