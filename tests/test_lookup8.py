"""kmr_config.hash_kind = lookup8 ("lookup8 = implement as selectable").

PARITY UNPINNED: the reference carries lookup8 as src/lookup8.h but calls it nowhere (a comment in KmerHasher::getHash,
src/Kmer.h:210-212), its header cannot be compiled here (its <config.h> wants Boost) and none of its tests holds a lookup8 value.
The oracle restates Bob Jenkins' published lookup8.c; what these tests can hold it to is the algorithm's own documented property
(hash2 over 64-bit words == hash over the same bytes on a little-endian machine, src/lookup8.h:163-168), that a spectrum's content
does not depend on the hash that places it, and that the device code equals the restatement bit for bit."""
import ctypes as C
import struct

import numpy as np
import pytest

from helpers import (KMR_HASH_LOOKUP8, KMR_MAP_WEAK, OracleSpectrum, default_config, oracle_lib, parse_image, synth_reads)


def test_byte_and_word_forms_agree():
    lib = oracle_lib()
    rng = np.random.default_rng(5)
    for n in range(0, 9):
        for level in (0, 0xDEADBEEF, 0x0123456789ABCDEF):
            words = rng.integers(0, 2 ** 63, size=max(n, 1), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=max(n, 1), dtype=np.uint64)
            data = words[:n].tobytes()
            assert lib.orc_hash8(data, len(data), level) == lib.orc_hash8_words(words.ctypes.data_as(C.POINTER(C.c_uint64)), n, level)


def test_every_tail_length_reads_every_byte_once():
    """flipping any single byte of a key of any length 1..60 changes the value; bytes behind the key's end do not matter"""
    lib = oracle_lib()
    rng = np.random.default_rng(6)
    for n in range(1, 61):
        key = bytearray(rng.integers(0, 256, size=n + 8, dtype=np.uint8).tobytes())
        h = lib.orc_hash8(bytes(key), n, 0xDEADBEEF)
        for i in range(n):
            k2 = bytearray(key)
            k2[i] ^= 0x40
            assert lib.orc_hash8(bytes(k2), n, 0xDEADBEEF) != h
        k2 = bytearray(key)
        k2[n] ^= 0xff
        assert lib.orc_hash8(bytes(k2), n, 0xDEADBEEF) == h
    # the length is part of the value: a key and the same key with a zero byte appended differ
    assert lib.orc_hash8(b"\x01\x02\x03", 3, 0) != lib.orc_hash8(b"\x01\x02\x03\x00", 4, 0)


def test_spectrum_content_does_not_depend_on_the_hash():
    rb = synth_reads(1500, read_len=90, seed=3, quality="noisy")
    a = OracleSpectrum(default_config(27, num_buckets_weak=128, num_buckets_singleton=256))
    b = OracleSpectrum(default_config(27, num_buckets_weak=128, num_buckets_singleton=256, hash_kind=KMR_HASH_LOOKUP8))
    for o in (a, b):
        o.add_reads(rb)
        o.finalize(2)
    assert a.stats() == b.stats()
    lib = oracle_lib()

    def entries(o, check_kind):
        nb, mask, buckets = parse_image(o.image(KMR_MAP_WEAK), 7, 12)
        out = {}
        for i, (keys, vals) in enumerate(buckets):
            for kk, v in zip(keys, vals):
                if check_kind:
                    assert lib.orc_hash8(bytes(kk), 7, 0xDEADBEEF) & mask == i          # the bucket of a key is lookup8 & mask
                out[bytes(kk)] = bytes(v[:10])          # count, weightedCount, directionBias (the last two bytes are padding)
        return out
    ea, eb = entries(a, False), entries(b, True)
    assert ea == eb and len(ea) == a.stats()["weak_entries"]
    assert not np.array_equal(a.image(KMR_MAP_WEAK), b.image(KMR_MAP_WEAK))


@pytest.mark.gpu
def test_device_hash_equals_the_restatement():
    import kmernator_amd as ka
    lib, olib = ka.load(), oracle_lib()
    lib.kmr_hash_of_kind.restype = C.c_uint64
    lib.kmr_hash_of_kind.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32]
    rng = np.random.default_rng(7)
    for n in range(1, 33):
        for _ in range(20):
            key = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
            assert lib.kmr_hash_of_kind(key, n, 1) == olib.orc_hash8(key, n, 0xDEADBEEF)
            assert lib.kmr_hash_of_kind(key, n, 0) == olib.orc_hash(key, n)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("k", [21, 31, 51, 95, 127])
def test_build_with_lookup8_equals_the_oracle(k, mode):
    from test_gpu_parity import compare_weak_images, run_both
    rb = synth_reads(1200, read_len=160, seed=k, quality="noisy", n_rate=0.002)
    kw = dict(num_buckets_weak=256, num_buckets_singleton=512, hash_kind=KMR_HASH_LOOKUP8)
    o, p = run_both(default_config(k, **kw), rb, mode=mode)
    assert compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), (k + 3) // 4, False) == o.stats()["weak_entries"]
    # owner and part filters go through the same hash
    for extra in (dict(rank=1, world_size=3), dict(num_parts=4, part_idx=1), dict(kmer_subsample=3)):
        o, p = run_both(default_config(k, **kw, **extra), rb, mode=mode)
        assert compare_weak_images(o.image(KMR_MAP_WEAK), p.image(KMR_MAP_WEAK), (k + 3) // 4, False) == o.stats()["weak_entries"]
    # scoring looks k-mers up through the lookup table built with the same hash
    got = p.getCount(np.concatenate([kk for kk, _ in parse_image(p.image(KMR_MAP_WEAK), (k + 3) // 4, 12)[2] if len(kk)]))
    assert np.all(got >= 2)
