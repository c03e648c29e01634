"""Shared test plumbing: ctypes bindings for the oracle (oracle/libkmr_oracle.so)
and the product C-ABI (kmernator_amd/csrc/libkmernator_amd.so), a FASTQ reader and
the synthetic read generator described in SURVEY.md section 8(d).

The oracle is test infrastructure: it is loaded here, by __graft_entry__.smoke()
and by bench.py's cpu_baseline leg only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libkmr_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_lookup3.so")
PRODUCT_SO = os.path.join(ROOT, "kmernator_amd", "csrc", "libkmernator_amd.so")

KMR_VALUE_COUNT_DIR, KMR_VALUE_EXT = 0, 1
KMR_HASH_LOOKUP3, KMR_HASH_LOOKUP8 = 0, 1
KMR_MAP_WEAK, KMR_MAP_SINGLETON, KMR_MAP_SOLID = 0, 1, 2


class KmrConfig(C.Structure):
    """Mirror of kmr_config in include/kmernator_amd.h."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("k", C.c_uint32),
        ("num_buckets_weak", C.c_uint64), ("num_buckets_singleton", C.c_uint64),
        ("estimated_raw_kmers", C.c_uint64),
        ("value_kind", C.c_uint32), ("min_weight", C.c_float),
        ("min_quality_score", C.c_uint32), ("fastq_start_char", C.c_uint32),
        ("ext_min_quality", C.c_uint32), ("separate_singletons", C.c_uint32),
        ("kmer_subsample", C.c_uint32), ("device", C.c_int32),
        ("rank", C.c_uint32), ("world_size", C.c_uint32),
        ("estimated_depth", C.c_double), ("estimated_error_rate", C.c_double),
        ("kmers_per_bucket", C.c_uint32), ("num_parts", C.c_uint32),
        ("part_idx", C.c_uint32), ("build_mode", C.c_uint32),
        ("max_table_entries", C.c_uint64),
        ("hash_kind", C.c_uint32), ("size_tracker", C.c_uint32),
    ]


class KmrStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "raw_kmers", "raw_good_kmers", "unique_kmers", "singleton_kmers",
        "discarded", "weak_entries", "singleton_entries", "reads")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class KmrDigest(C.Structure):
    """Mirror of kmr_digest in include/kmernator_amd.h."""
    _fields_ = [(n, C.c_uint64) for n in ("entries", "count_sum", "dir_sum", "hash_sum", "hash_xor")] + [("weighted_sum", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class KmrArtifactConfig(C.Structure):
    """Mirror of kmr_artifact_config in include/kmernator_amd.h."""
    _fields_ = [(n, C.c_uint32) for n in ("match_length", "edit_distance", "build_edits", "simple_repeat_begin", "simple_repeat_end",
                                          "phix_idx", "reference_begin", "min_quality", "fastq_start_char")] + [("min_read_length", C.c_float)]


def artifact_config(**kw):
    """the reference's defaults (FilterKnownOdditiesOptions, src/FilterKnownOddities.h:72-74; ReadSelectorOptions min-read-length)"""
    c = KmrArtifactConfig(24, 2, 2, 0, 0, 0, 0, 3, 33, 0.40)
    for name, v in kw.items():
        setattr(c, name, v)
    return c


def default_config(k, **kw):
    c = KmrConfig()
    c.struct_size = C.sizeof(KmrConfig)
    c.k = k
    c.value_kind = KMR_VALUE_COUNT_DIR
    c.min_weight = 0.10
    c.min_quality_score = 3
    c.fastq_start_char = 33
    c.ext_min_quality = 20
    c.separate_singletons = 1
    c.kmer_subsample = 1
    c.device = -1
    c.rank = 0
    c.world_size = 1
    c.estimated_depth = 20.0
    c.estimated_error_rate = 0.35
    c.kmers_per_bucket = 32
    c.num_parts = 1
    c.part_idx = 0
    for name, v in kw.items():
        setattr(c, name, v)
    return c


def build_oracle():
    if os.environ.get("KMR_ORACLE_SO"):      # a build of the oracle made elsewhere (the sanitizer run: oracle compiled with -fsanitize=address,undefined)
        return os.environ["KMR_ORACLE_SO"]
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
            os.path.join(ORACLE_DIR, "kmr_oracle.cpp")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return ORACLE_SO


_oracle = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(build_oracle())
        u8p, u32p, u64p, f32p, f64p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint32, C.c_uint64, C.c_float, C.c_double))
        lib.orc_create.restype = C.c_void_p
        lib.orc_create.argtypes = [C.POINTER(KmrConfig)]
        lib.orc_destroy.argtypes = [C.c_void_p]
        lib.orc_add_reads.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, u64p, C.c_uint64, C.c_uint64, u8p, C.c_int]
        lib.orc_finalize.argtypes = [C.c_void_p, C.c_uint32]
        lib.orc_get_stats.argtypes = [C.c_void_p, C.POINTER(KmrStats)]
        lib.orc_num_buckets.argtypes = [C.c_void_p, C.c_int, u64p]
        lib.orc_lookup.argtypes = [C.c_void_p, u8p, C.c_uint64, u32p]
        lib.orc_image_size.argtypes = [C.c_void_p, C.c_int, u64p]
        lib.orc_write_image.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
        lib.orc_load_image.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
        lib.orc_count_histogram.argtypes = [C.c_void_p, u64p, f64p, C.c_uint32]
        lib.orc_dump.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_int]
        lib.orc_export_entries.restype = C.c_uint64
        lib.orc_export_entries.argtypes = [C.c_void_p, u8p, u32p, u32p, f32p, u32p, C.c_uint64]
        lib.orc_quality_table.argtypes = [C.c_uint, C.c_uint, f64p]
        lib.orc_hash.restype = C.c_uint64
        lib.orc_hash.argtypes = [C.c_char_p, C.c_uint32]
        lib.orc_hashlittle2.argtypes = [C.c_char_p, C.c_uint64, u32p, u32p]
        lib.orc_size_tracker.restype = C.c_uint64
        lib.orc_size_tracker.argtypes = [C.c_void_p, C.c_int, C.c_int, u64p, C.c_uint64]
        lib.orc_hash8.restype = C.c_uint64
        lib.orc_hash8.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64]
        lib.orc_hash8_words.restype = C.c_uint64
        lib.orc_hash8_words.argtypes = [u64p, C.c_uint64, C.c_uint64]
        lib.orc_compress_sequence.restype = C.c_int64
        lib.orc_compress_sequence.argtypes = [C.c_char_p, C.c_uint64, u8p, u32p, C.c_char_p, C.c_uint64]
        lib.orc_reverse_complement.argtypes = [u8p, u8p, C.c_uint32]
        lib.orc_shift_left.argtypes = [u8p, u8p, C.c_uint32, C.c_uint32, C.c_int]
        lib.orc_least_complement.argtypes = [u8p, C.c_uint32, u8p]
        lib.orc_build_weighted_kmers.restype = C.c_int64
        lib.orc_build_weighted_kmers.argtypes = [C.POINTER(KmrConfig), C.c_char_p, C.c_char_p, C.c_uint32, u8p, f32p, u8p, C.c_uint64]
        lib.orc_bucket_idx.restype = C.c_uint64
        lib.orc_bucket_idx.argtypes = [C.c_uint64, C.c_uint64]
        lib.orc_local_thread_id.restype = C.c_uint32
        lib.orc_local_thread_id.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        lib.orc_distributed_thread_id.restype = C.c_uint32
        lib.orc_distributed_thread_id.argtypes = [C.c_uint64, C.c_uint32]
        lib.orc_min_power_of_2.restype = C.c_uint64
        lib.orc_min_power_of_2.argtypes = [C.c_uint64]
        lib.orc_extract_records_by_owner.restype = C.c_int64
        lib.orc_extract_records_by_owner.argtypes = [C.POINTER(KmrConfig), C.c_char_p, C.c_char_p, u64p, C.c_uint64, u8p, u8p, C.c_uint64, u64p]
        lib.orc_insert_records.argtypes = [C.c_void_p, u8p, C.c_uint64]
        lib.orc_derive_buckets.argtypes = [C.POINTER(KmrConfig), u64p, u64p]
        lib.orc_histogram.argtypes = [C.c_void_p, C.c_uint32, C.c_double, u64p, u64p, f64p]
        lib.orc_subtract_reference.argtypes = [C.c_void_p, C.c_void_p]
        lib.orc_subtracted.restype = C.c_uint64
        lib.orc_subtracted.argtypes = [C.c_void_p]
        lib.orc_parse_fastq.restype = C.c_int64
        lib.orc_parse_fastq.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, u64p, u64p, u32p,
                                        C.c_uint64, C.c_uint64, u32p]
        lib.orc_map_digest.argtypes = [C.c_void_p, C.c_int, C.POINTER(KmrDigest)]
        lib.orc_merge_add.argtypes = [C.c_void_p, C.c_void_p]
        lib.orc_synth_reads.restype = None
        lib.orc_synth_reads.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, u64p, C.c_int]
        _oracle = lib
    return _oracle


class OracleArtifactFilter:
    """oracle restatement of FilterKnownOddities (f4)"""

    def __init__(self, cfg, fasta):
        lib = oracle_lib()
        lib.orc_artifact_create.restype = C.c_void_p
        lib.orc_artifact_create.argtypes = [C.POINTER(KmrArtifactConfig), C.c_char_p, C.c_uint64]
        lib.orc_artifact_free.argtypes = [C.c_void_p]
        lib.orc_artifact_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        lib.orc_artifact_entries.restype = C.c_uint64
        lib.orc_artifact_entries.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_uint64]
        lib.orc_artifact_apply.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_int64)] + \
            [C.POINTER(C.c_uint32)] * 3 + [C.POINTER(C.c_uint8)] + [C.POINTER(C.c_uint32)] * 2
        self.lib = lib
        self.cfg = cfg
        fasta = bytes(fasta)
        self.h = lib.orc_artifact_create(C.byref(cfg), fasta, len(fasta))
        assert self.h

    def info(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint32()
        self.lib.orc_artifact_info(self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def entries(self):
        n = self.info()[1]
        keys = np.zeros(n, dtype=np.uint64)
        vals = np.zeros(n, dtype=np.uint32)
        assert self.lib.orc_artifact_entries(self.h, _ptr(keys, C.c_uint64), _ptr(vals, C.c_uint32), n) == n
        return keys, vals

    def apply(self, rb, mate=None):
        """dict of per-read arrays: value, min_pass, max_pass, action, remnant_off, remnant_len"""
        n = rb.n
        out = {k: np.zeros(n, dtype=np.uint32) for k in ("value", "min_pass", "max_pass", "remnant_off", "remnant_len")}
        out["action"] = np.zeros(n, dtype=np.uint8)
        m = None if mate is None else _ptr(np.ascontiguousarray(mate, dtype=np.int64), C.c_int64)
        self.lib.orc_artifact_apply(self.h, rb.bases.ctypes.data_as(C.c_char_p), rb.quals.ctypes.data_as(C.c_char_p), _ptr(rb.offsets, C.c_uint64), n, m,
                                    _ptr(out["value"], C.c_uint32), _ptr(out["min_pass"], C.c_uint32), _ptr(out["max_pass"], C.c_uint32),
                                    _ptr(out["action"], C.c_uint8), _ptr(out["remnant_off"], C.c_uint32), _ptr(out["remnant_len"], C.c_uint32))
        return out

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_artifact_free(self.h)
            self.h = None


def apply_artifact_result(rb, res):
    """the read set after applyFilter: trimmed / emptied reads in place, remnants appended (host-side model for the tests)"""
    seqs, quals = [], []
    for i in range(rb.n):
        s, q = rb.seq(i), rb.qual(i)
        a = int(res["action"][i])
        if a == 1:
            s, q = s[int(res["min_pass"][i]):int(res["max_pass"][i])], q[int(res["min_pass"][i]):int(res["max_pass"][i])]
        elif a == 2:
            s, q = b"", b""
        seqs.append(s)
        quals.append(q)
    for i in range(rb.n):
        if res["remnant_len"][i]:
            o, l = int(res["remnant_off"][i]), int(res["remnant_len"][i])
            seqs.append(rb.seq(i)[o:o + l])
            quals.append(rb.qual(i)[o:o + l])
    return ReadBatch(seqs, quals)


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class ReadBatch:
    """Flat read arrays in the layout kmr_add_reads takes."""

    def __init__(self, seqs, quals=None, discarded=None):
        lens = np.array([len(s) for s in seqs], dtype=np.uint64)
        self.offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
        np.cumsum(lens, out=self.offsets[1:])
        self.bases = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy() if len(seqs) else np.zeros(0, np.uint8)
        self.quals = None
        if quals is not None:
            self.quals = np.frombuffer(b"".join(quals), dtype=np.uint8).copy() if len(quals) else np.zeros(0, np.uint8)
            assert self.quals.size == self.bases.size
        self.discarded = None if discarded is None else np.asarray(discarded, dtype=np.uint8)
        self.n = len(seqs)

    @classmethod
    def from_arrays(cls, bases, quals, offsets):
        self = cls.__new__(cls)
        self.bases, self.quals, self.offsets = bases, quals, offsets
        self.discarded = None
        self.n = len(offsets) - 1
        return self

    def slice(self, lo, hi):
        b0, b1 = int(self.offsets[lo]), int(self.offsets[hi])
        out = ReadBatch.__new__(ReadBatch)
        out.offsets = (self.offsets[lo:hi + 1] - self.offsets[lo]).copy()
        out.bases = self.bases[b0:b1].copy()
        out.quals = None if self.quals is None else self.quals[b0:b1].copy()
        out.discarded = None if self.discarded is None else self.discarded[lo:hi].copy()
        out.n = hi - lo
        return out

    def seq(self, i):
        return self.bases[int(self.offsets[i]):int(self.offsets[i + 1])].tobytes()

    def qual(self, i):
        return self.quals[int(self.offsets[i]):int(self.offsets[i + 1])].tobytes()


def read_fastq(path):
    seqs, quals, names = [], [], []
    with open(path, "rb") as f:
        while True:
            name = f.readline()
            if not name:
                break
            seq = f.readline().rstrip(b"\r\n")
            f.readline()
            q = f.readline().rstrip(b"\r\n")
            names.append(name.rstrip(b"\r\n")[1:])
            seqs.append(seq)
            quals.append(q)
    rb = ReadBatch(seqs, quals)
    rb.names = names
    return rb


class _SpectrumCommon:
    """Method set shared by the oracle and the product wrappers so parity tests
    drive both with the same code."""

    def stats(self):
        s = KmrStats()
        self._call("get_stats", self.h, C.byref(s))
        return s.as_dict()

    def num_buckets(self, which):
        v = C.c_uint64()
        self._call("num_buckets", self.h, which, C.byref(v))
        return v.value

    def finalize(self, min_depth=2):
        self._call("finalize", self.h, min_depth)

    def lookup(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint8)
        n = keys.size // self.kb
        out = np.zeros(n, dtype=np.uint32)
        if n:
            self._call("lookup", self.h, _ptr(keys, C.c_uint8), n, _ptr(out, C.c_uint32))
        return out

    def image(self, which=KMR_MAP_WEAK):
        sz = C.c_uint64()
        self._call("image_size", self.h, which, C.byref(sz))
        buf = np.zeros(sz.value, dtype=np.uint8)
        self._call("write_image", self.h, which, buf.ctypes.data_as(C.c_void_p), sz.value)
        return buf

    def load_image(self, which, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        self._call("load_image", self.h, which, buf.ctypes.data_as(C.c_void_p), buf.size)

    def histogram(self, nbins=256):
        counts = np.zeros(nbins, dtype=np.uint64)
        weights = np.zeros(nbins, dtype=np.float64)
        self._call("count_histogram", self.h, _ptr(counts, C.c_uint64), _ptr(weights, C.c_double), nbins)
        return counts, weights


class OracleSpectrum(_SpectrumCommon):
    def __init__(self, cfg):
        self.lib = oracle_lib()
        self.cfg = cfg
        self.k = cfg.k
        self.kb = (cfg.k + 3) // 4
        self.h = self.lib.orc_create(C.byref(cfg))
        assert self.h

    def _call(self, name, *a):
        rc = getattr(self.lib, "orc_" + name)(*a)
        if rc != 0:
            raise RuntimeError("orc_%s -> %d" % (name, rc))

    def add_reads(self, rb, first_idx=0, threads=1):
        self._call("add_reads", self.h, rb.bases.ctypes.data_as(C.c_char_p),
                   None if rb.quals is None else rb.quals.ctypes.data_as(C.c_char_p),
                   _ptr(rb.offsets, C.c_uint64), rb.n, first_idx,
                   None if rb.discarded is None else _ptr(rb.discarded, C.c_uint8), threads)

    def insert_records(self, recs, n):
        recs = np.ascontiguousarray(recs, dtype=np.uint8)
        self._call("insert_records", self.h, _ptr(recs, C.c_uint8), n)

    def size_tracker(self, per_read=True, force_last=True):
        """the size history: the reference's own sampling (before every k-mer) or after every read; [n][4] uint64"""
        lib = oracle_lib()
        n = lib.orc_size_tracker(self.h, 1 if per_read else 0, 1 if force_last else 0, None, 0)
        el = np.zeros((n, 4), dtype=np.uint64)
        if n:
            lib.orc_size_tracker(self.h, 1 if per_read else 0, 1 if force_last else 0, el.ctypes.data_as(C.POINTER(C.c_uint64)), n)
        return el

    def ref_histogram(self, zoom_max=256, log_base=2.0):
        nb = (1 << 16) + 2 + zoom_max
        v, c, w = np.zeros(nb, dtype=np.uint64), np.zeros(nb, dtype=np.uint64), np.zeros(nb, dtype=np.float64)
        self._call("histogram", self.h, zoom_max, log_base, _ptr(v, C.c_uint64), _ptr(c, C.c_uint64), _ptr(w, C.c_double))
        return v, c, w

    def dump(self, path, min_depth, graph):
        self._call("dump", self.h, path.encode(), min_depth, 1 if graph else 0)

    def merge_add(self, other):
        """KmerMapByKmerArrayPair::mergeAdd of the weak maps (other's weak map is emptied)"""
        self._call("merge_add", self.h, other.h)

    def digest(self, which=KMR_MAP_WEAK):
        d = KmrDigest()
        self._call("map_digest", self.h, which, C.byref(d))
        return d.as_dict()

    def entries(self):
        n = self.stats()["weak_entries"]
        keys = np.zeros(n * self.kb, dtype=np.uint8)
        count = np.zeros(n, dtype=np.uint32)
        dirb = np.zeros(n, dtype=np.uint32)
        w = np.zeros(n, dtype=np.float32)
        ext = np.zeros(n * 12, dtype=np.uint32)
        got = self.lib.orc_export_entries(self.h, _ptr(keys, C.c_uint8), _ptr(count, C.c_uint32), _ptr(dirb, C.c_uint32),
                                          _ptr(w, C.c_float), _ptr(ext, C.c_uint32), n)
        assert got == n
        return keys.reshape(n, self.kb), count, dirb, w, ext.reshape(n, 12)

    def close(self):
        if self.h:
            self.lib.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def oracle_weighted_kmers(cfg, seq, qual):
    lib = oracle_lib()
    kb = (cfg.k + 3) // 4
    n = max(0, len(seq) - cfg.k + 1)
    keys = np.zeros(max(n, 1) * kb, dtype=np.uint8)
    w = np.zeros(max(n, 1), dtype=np.float32)
    ext = np.zeros(max(n, 1) * 4, dtype=np.uint8)
    got = lib.orc_build_weighted_kmers(C.byref(cfg), seq, qual, len(seq), _ptr(keys, C.c_uint8), _ptr(w, C.c_float),
                                       _ptr(ext, C.c_uint8), max(n, 1))
    assert got == n, (got, n)
    return keys[:n * kb].reshape(n, kb), w[:n], ext[:n * 4].reshape(n, 4)


# ---------------------------------------------------------------- synthetic reads
def synth_reads(n_reads, read_len=150, genome_len=None, seed=1, err=0.01, quality="flat", n_rate=0.0):
    """Synthetic reads per SURVEY.md 8(d): uniform random genome, uniform start, random
    strand, per-base substitution with probability err, Phred-33 quals.  quality:
    'flat' ('I' everywhere) or 'noisy' (Q in {40,30,20,10,2} with probs
    {.80,.10,.05,.04,.01}, substituted bases forced to Q10).  n_rate adds N bases."""
    rng = np.random.default_rng(seed)
    if genome_len is None:
        genome_len = max(read_len * 2, n_reads * read_len // 30)
    genome = rng.integers(0, 4, size=genome_len, dtype=np.uint8)
    starts = rng.integers(0, genome_len - read_len + 1, size=n_reads)
    idx = starts[:, None] + np.arange(read_len)[None, :]
    codes = genome[idx]
    strand = rng.integers(0, 2, size=n_reads).astype(bool)
    codes[strand] = (3 - codes[strand])[:, ::-1]
    errs = rng.random(codes.shape) < err
    shift = rng.integers(1, 4, size=codes.shape, dtype=np.uint8)
    codes = np.where(errs, (codes + shift) & 3, codes).astype(np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    bases = lut[codes]
    if quality == "flat":
        quals = np.full(codes.shape, ord("I"), dtype=np.uint8)
    else:
        qv = np.array([40, 30, 20, 10, 2], dtype=np.uint8)
        pick = rng.choice(5, size=codes.shape, p=[0.80, 0.10, 0.05, 0.04, 0.01])
        quals = (qv[pick] + 33).astype(np.uint8)
        quals[errs] = 10 + 33
    if n_rate > 0:
        ns = rng.random(codes.shape) < n_rate
        bases = np.where(ns, ord("N"), bases).astype(np.uint8)
        quals = np.where(ns, 33 + 2, quals).astype(np.uint8)
    offsets = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len))
    return ReadBatch.from_arrays(np.ascontiguousarray(bases.reshape(-1)), np.ascontiguousarray(quals.reshape(-1)), offsets)


def synth_reads_8d(seed, first_read, n_reads, read_len=150, genome_len=None, noisy=False, threads=8):
    """SURVEY.md 8(d)'s generator, CPU statement (orc_synth_reads): the same bytes as kmr_synth_reads_dev for the same arguments"""
    lib = oracle_lib()
    bases = np.zeros(n_reads * read_len, dtype=np.uint8)
    quals = np.zeros(n_reads * read_len, dtype=np.uint8)
    offsets = np.zeros(n_reads + 1, dtype=np.uint64)
    lib.orc_synth_reads(seed, first_read, n_reads, read_len, genome_len, 1 if noisy else 0, bases.ctypes.data_as(C.c_void_p),
                        quals.ctypes.data_as(C.c_void_p), _ptr(offsets, C.c_uint64), threads)
    return ReadBatch.from_arrays(bases, quals, offsets)


def parse_image(buf, kb, vsize):
    """Decode the reference on-disk map layout (src/Kmer.h:3143-3159) into
    (num_buckets, mask, [(keys[n,kb], values[n,vsize])...])."""
    hdr = np.frombuffer(buf[:16].tobytes(), dtype=np.uint64)
    nb, mask = int(hdr[0]), int(hdr[1])
    offs = np.frombuffer(buf[16:16 + 8 * nb].tobytes(), dtype=np.uint64)
    buckets = []
    for i in range(nb):
        o = int(offs[i])
        n = int(np.frombuffer(buf[o:o + 4].tobytes(), dtype=np.uint32)[0])
        keys = buf[o + 4:o + 4 + n * kb].reshape(n, kb)
        vals = buf[o + 4 + n * kb:o + 4 + n * (kb + vsize)].reshape(n, vsize)
        buckets.append((keys, vals))
    return nb, mask, buckets


def _dmix(x):
    """mix() of include/kmernator_amd.h's digest, on uint64 arrays"""
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def digest_of_image(buf, kb, ext=False, singleton=False):
    """kmr_digest as include/kmernator_amd.h defines it, computed from the bytes of a stored map (numpy restatement: what
    kmr_map_digest and orc_map_digest are both held to)"""
    vsize = (5 if ext else 1) if singleton else (60 if ext else 12)
    _, _, buckets = parse_image(buf, kb, vsize)
    keys = np.concatenate([b[0] for b in buckets]) if buckets else np.zeros((0, kb), np.uint8)
    vals = np.concatenate([b[1] for b in buckets]) if buckets else np.zeros((0, vsize), np.uint8)
    n = keys.shape[0]
    W = (kb + 7) // 8
    padded = np.zeros((n, 8 * W), dtype=np.uint8)
    padded[:, :kb] = keys
    words = padded.reshape(n, W, 8)[:, :, ::-1].copy().view(np.uint64).reshape(n, W)      # big-endian words
    out = {"entries": n}
    if singleton:
        w8 = vals[:, 0].astype(np.uint64)
        x = np.uint64(1 << 32) | w8
        if ext:
            x = x | (np.ascontiguousarray(vals[:, 1:5]).view(np.uint32).reshape(n).astype(np.uint64) << np.uint64(40))
        out["count_sum"] = int((w8 != 0).sum())
        out["dir_sum"] = 0
        out["weighted_sum"] = float(np.where(w8 != 0, (w8.astype(np.float64) - 1.0) / 254.0, 0.0).sum())
    else:
        count = np.ascontiguousarray(vals[:, 0:2]).view(np.uint16).reshape(n).astype(np.uint64)
        dirb = np.ascontiguousarray(vals[:, 8:10]).view(np.uint16).reshape(n).astype(np.uint64)
        x = count | (dirb << np.uint64(16))
        out["count_sum"] = int(count.sum())
        out["dir_sum"] = int(dirb.sum())
        out["weighted_sum"] = float(np.ascontiguousarray(vals[:, 4:8]).view(np.float32).reshape(n).astype(np.float64).sum())
    for j in range(W):
        x = _dmix(x ^ words[:, j])
    if ext and not singleton:
        t = np.ascontiguousarray(vals[:, 12:60]).view(np.uint32).reshape(n, 12).astype(np.uint64)
        for j in range(0, 12, 2):
            x = _dmix(x ^ (t[:, j] | (t[:, j + 1] << np.uint64(32))))
    with np.errstate(over="ignore"):
        out["hash_sum"] = int(x.sum(dtype=np.uint64)) if n else 0
    out["hash_xor"] = int(np.bitwise_xor.reduce(x)) if n else 0
    return out


def digests_agree(a, b, rel=1e-6):
    """two kmr_digests: integers equal, weighted_sum within rel (a float accumulation whose order is not part of the contract)"""
    for key in ("entries", "count_sum", "dir_sum", "hash_sum", "hash_xor"):
        if int(a[key]) != int(b[key]):
            return False
    return abs(a["weighted_sum"] - b["weighted_sum"]) <= rel * max(1.0, abs(a["weighted_sum"]))


def add_digests(a, b):
    """part / rank digests combine into the whole map's (a may be None)"""
    if a is None:
        return dict(b)
    out = {key: a[key] + b[key] for key in ("entries", "count_sum", "dir_sum", "weighted_sum")}
    out["hash_sum"] = (a["hash_sum"] + b["hash_sum"]) & 0xFFFFFFFFFFFFFFFF
    out["hash_xor"] = a["hash_xor"] ^ b["hash_xor"]
    return out


def full_size_golden(name):
    """tests/golden/full_size_digests.json[name] (written by tests/golden/make_full_size_digests.py from the oracle)"""
    import json
    with open(os.path.join(GOLDEN, "full_size_digests.json")) as f:
        return json.load(f)[name]


def oracle_extract_by_owner(cfg, rb, seg_capacity):
    """(records uint8 [world*seg_capacity*rec_bytes], counts uint64 [world])"""
    lib = oracle_lib()
    W = (((cfg.k + 3) // 4) + 7) // 8
    recb = 8 * W + (8 if cfg.value_kind == KMR_VALUE_EXT else 4)
    recs = np.zeros(cfg.world_size * seg_capacity * recb, dtype=np.uint8)
    counts = np.zeros(cfg.world_size, dtype=np.uint64)
    n = lib.orc_extract_records_by_owner(C.byref(cfg), rb.bases.ctypes.data_as(C.c_char_p),
                                         None if rb.quals is None else rb.quals.ctypes.data_as(C.c_char_p),
                                         _ptr(rb.offsets, C.c_uint64), rb.n,
                                         None if rb.discarded is None else _ptr(rb.discarded, C.c_uint8),
                                         _ptr(recs, C.c_uint8), seg_capacity, _ptr(counts, C.c_uint64))
    assert n >= 0
    return recs, counts


def oracle_parse_fastq(text, start_char=33, input_base=33, store_comment=True):
    """oracle restatement of the reference's FASTQ stream parser: (ReadBatch with .names, final input base) or None
    where the reference throws"""
    lib = oracle_lib()
    text = bytes(text)
    cap_reads = text.count(b"\n") // 4 + 2
    bases = np.zeros(len(text) + 1, dtype=np.uint8)
    quals = np.zeros(len(text) + 1, dtype=np.uint8)
    offsets = np.zeros(cap_reads + 1, dtype=np.uint64)
    noff = np.zeros(cap_reads, dtype=np.uint64)
    nlen = np.zeros(cap_reads, dtype=np.uint32)
    fb = C.c_uint32()
    n = lib.orc_parse_fastq(text, len(text), start_char, input_base, 1 if store_comment else 0, bases.ctypes.data_as(C.c_void_p),
                            quals.ctypes.data_as(C.c_void_p), _ptr(offsets, C.c_uint64), _ptr(noff, C.c_uint64), _ptr(nlen, C.c_uint32),
                            cap_reads, len(text), C.byref(fb))
    if n < 0:
        return None
    tot = int(offsets[n])
    rb = ReadBatch.from_arrays(bases[:tot].copy(), quals[:tot].copy(), offsets[:n + 1].copy())
    rb.names = [text[int(noff[i]):int(noff[i]) + int(nlen[i])] for i in range(n)]
    return rb, fb.value
