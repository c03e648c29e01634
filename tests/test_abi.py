"""CPU-side checks of the product library: it loads, exports every symbol the header
declares, the stateless helpers are bit-identical to the oracle, and compute entry
points fail loudly (KMR_ERR_NO_DEVICE) instead of falling back when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import kmernator_amd as ka
from kmernator_amd import _lib
from helpers import ROOT, oracle_lib, _ptr


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "kmernator_amd.h")).read()
    declared = set(re.findall(r"\b(kmr_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"kmr_handle"}
    lib = ka.load()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(_lib.EXPORTS)
    assert lib.kmr_abi_version() == 1


def test_config_struct_matches_header():
    c = ka.default_config(31)
    assert c.struct_size == C.sizeof(ka.KmrConfig)
    assert (c.min_quality_score, c.fastq_start_char, c.ext_min_quality, c.separate_singletons) == (3, 33, 20, 1)
    assert abs(c.min_weight - 0.10) < 1e-7 and c.kmers_per_bucket == 32


def test_host_helpers_match_oracle():
    lib, orc = ka.load(), oracle_lib()
    rng = np.random.default_rng(2)
    for n in range(1, 33):
        for _ in range(20):
            key = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
            h = lib.kmr_hash(key, n)
            assert h == orc.orc_hash(key, n)
            assert lib.kmr_bucket_idx(h, 4096) == orc.orc_bucket_idx(h, 4096)
            assert lib.kmr_distributed_thread_id(h, 8) == orc.orc_distributed_thread_id(h, 8)
            assert lib.kmr_local_thread_id(h, 4096, 7) == orc.orc_local_thread_id(h, 4096, 7)
    for k in (1, 4, 21, 31, 32, 33, 51, 64, 95, 128):
        kb = (k + 3) // 4
        for _ in range(20):
            s = "".join("ACGT"[i] for i in rng.integers(0, 4, k)).encode()
            a = np.zeros(kb, np.uint8)
            b = np.zeros(kb, np.uint8)
            assert lib.kmr_compress_sequence(s, k, _ptr(a, C.c_uint8), None, None, 0) == 0
            assert orc.orc_compress_sequence(s, k, _ptr(b, C.c_uint8), None, None, 0) == 0
            assert a.tobytes() == b.tobytes()
            ca, cb = np.zeros(kb, np.uint8), np.zeros(kb, np.uint8)
            assert lib.kmr_least_complement(_ptr(a, C.c_uint8), k, _ptr(ca, C.c_uint8)) == orc.orc_least_complement(_ptr(b, C.c_uint8), k, _ptr(cb, C.c_uint8))
            assert ca.tobytes() == cb.tobytes()
    mp = np.zeros(8, np.uint32)
    mc = C.create_string_buffer(8)
    assert lib.kmr_compress_sequence(b"AC.TNX", 6, None, _ptr(mp, C.c_uint32), mc, 8) == 3
    assert list(mp[:3]) == [2, 4, 5] and mc.raw[:3] == b"NNX"


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ka.KmerSpectrumError, match="NO_DEVICE"):
        ka.KmerSpectrum(ka.default_config(31))
