"""Host-side mirror of the reference's KmerSpectrum / KmerMap interface for the
spectrum-build path, over the C-ABI (include/kmernator_amd.h).

Method names follow the reference (src/KmerSpectrum.h): buildKmerSpectrum (:2081-2115),
purgeMinDepth (:1805), getCount (:701), storeMmap / restoreMmap (:476-518), getRawKmers...
(:455-459); MeraculousDistributedKmerSpectrum::dumpCounts/dumpGraphs (src/Meraculous.h:107-134).
Errors surface as KmerSpectrumError (the reference throws LoggedException, src/Log.h:442-484).
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import KMR_MAP_SINGLETON, KMR_MAP_WEAK, KmrStats


class KmerSpectrumError(RuntimeError):
    pass


class SizeTracker:
    """KmerSpectrum::SizeTracker (src/KmerSpectrum.h:812-900): elements[i] = (rawKmers, rawGoodKmers, uniqueKmers, singletonKmers)"""

    def __init__(self, elements):
        self.elements = np.asarray(elements, dtype=np.uint64).reshape(-1, 4)

    def getLastElement(self):
        return tuple(int(v) for v in self.elements[-1]) if len(self.elements) else (0, 0, 0, 0)

    def toString(self):
        """SizeTracker::toString (:873-878): the header, then one tab-separated line per element"""
        return "rawKmers\trawGoodKmers\tuniqueKmers\tsingletonKmers\n" + "".join("%d\t%d\t%d\t%d\n" % tuple(int(v) for v in e) for e in self.elements)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


class KmerSpectrum:
    """One spectrum (weak + singleton maps) resident on one MI355X."""

    def __init__(self, cfg):
        self.lib = _lib.load()
        self.cfg = cfg
        self.k = cfg.k
        self.kb = (cfg.k + 3) // 4
        h = C.c_void_p()
        rc = self.lib.kmr_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise KmerSpectrumError("kmr_create: %s: %s" % (_lib.STATUS.get(rc, rc), self.lib.kmr_last_error(None).decode()))
        self.h = h

    # -- plumbing
    def _check(self, rc, what):
        if rc != 0:
            raise KmerSpectrumError("%s: %s: %s" % (what, _lib.STATUS.get(rc, rc), self.lib.kmr_last_error(self.h).decode()))

    def _call(self, name, *a):
        self._check(getattr(self.lib, "kmr_" + name)(*a), "kmr_" + name)

    def close(self):
        if getattr(self, "h", None):
            self.lib.kmr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tune(self, **knobs):
        """kmr_tune: implementation knobs of this handle (none changes a result)"""
        for name, v in knobs.items():
            self._call("tune", self.h, name.encode(), float(v))
        return self

    # -- build
    def build_info(self, what):
        """kmr_build_info: a figure of the current build ("lists", "uniform_count", "chunk_pool_chunks")"""
        v = C.c_double()
        self._call("build_info", self.h, what.encode(), C.byref(v))
        return v.value

    def buildKmerSpectrum(self, bases, quals, offsets, first_read_idx=0, discarded=None):
        """KmerSpectrum::buildKmerSpectrum(const ReadSet&) on flat host arrays."""
        bases = _u8(bases)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        q = None if quals is None else _u8(quals)
        d = None if discarded is None else _u8(discarded)
        self._call("add_reads", self.h, bases.ctypes.data_as(C.c_void_p), None if q is None else q.ctypes.data_as(C.c_void_p),
                   offsets.ctypes.data_as(C.POINTER(C.c_uint64)), n, first_read_idx,
                   None if d is None else d.ctypes.data_as(C.POINTER(C.c_uint8)))

    def buildKmerSpectrumDevice(self, bases_ptr, quals_ptr, offsets_ptr, n_reads, total_bases, first_read_idx=0, discarded_ptr=None):
        """Same, device pointers (torch tensor .data_ptr()); asynchronous, see sync()."""
        self._call("add_reads_dev", self.h, bases_ptr, quals_ptr, offsets_ptr, n_reads, total_bases, first_read_idx, discarded_ptr)

    def buildKmerSpectrumTwoBit(self, twobit, twobit_offsets, offsets, quals=None, uniform_quality=0, markups=None, first_read_idx=0, discarded=None):
        """KmerSpectrum::buildKmerSpectrum(const ReadSet&) on reads kept as the reference's Read keeps them: 2-bit packed bases (every read on
        bytes of its own) + markups (positions, chars, offsets[n+1]) + qualities (an array, or one character for all, or none)
        -- kmr_add_reads_twobit, host arrays"""
        tw = _u8(twobit)
        to = np.ascontiguousarray(twobit_offsets, dtype=np.uint64)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        q = None if quals is None else _u8(quals)
        d = None if discarded is None else _u8(discarded)
        mp = mc = mo = None
        if markups is not None:
            mp = np.ascontiguousarray(markups[0], dtype=np.uint32); mc = _u8(markups[1]); mo = np.ascontiguousarray(markups[2], dtype=np.uint64)
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        self._call("add_reads_twobit", self.h, vp(tw), vp(to), vp(offsets), vp(mo), vp(mp), vp(mc), vp(q), int(uniform_quality), n, first_read_idx, vp(d))

    def buildKmerSpectrumTwoBitDevice(self, twobit_ptr, twobit_offsets_ptr, offsets_ptr, n_reads, total_bases, quals_ptr=None, uniform_quality=0,
                                      markup_offsets_ptr=None, markup_pos_ptr=None, markup_char_ptr=None, first_read_idx=0, discarded_ptr=None):
        """Same, device pointers (kmr_add_reads_twobit_dev); asynchronous, see sync().  quals_ptr points at the quality of the call's first base."""
        self._call("add_reads_twobit_dev", self.h, twobit_ptr, twobit_offsets_ptr, offsets_ptr, markup_offsets_ptr, markup_pos_ptr, markup_char_ptr,
                   quals_ptr, int(uniform_quality), n_reads, total_bases, first_read_idx, discarded_ptr)

    def buildKmerSpectrumFromReadSet(self, read_set, first_read_idx=0):
        """KmerSpectrum::buildKmerSpectrum(const ReadSet&) on a device-resident ReadSet (kmr_add_read_batch)."""
        self._call("add_read_batch", self.h, read_set.r, first_read_idx)

    def subtractReference(self, other):
        """KmerSpectrum::subtractReference: k-mers of the finalized spectrum `other` are skipped by later builds"""
        self._subtracting = other          # keep it alive
        self._call("subtract_reference", self.h, None if other is None else other.h)

    def getSubtracted(self):
        v = C.c_uint64()
        self._call("subtracted", self.h, C.byref(v))
        return v.value

    def reset(self):
        """weak.reset(false); singleton.reset(false) of buildKmerSpectrum (src/KmerSpectrum.h:2091-2096)"""
        self._call("reset", self.h)

    def release_table(self):
        self._call("release_table", self.h)

    def sync(self):
        self._call("sync", self.h)

    def purgeMinDepth(self, min_depth=2):
        """purgeMinDepth + optimize(); the maps become immutable (kmr_finalize)."""
        self._call("finalize", self.h, min_depth)

    finalize = purgeMinDepth

    # -- owner-partitioned pieces (one process per GPU)
    def extractByOwnerDevice(self, bases_ptr, quals_ptr, offsets_ptr, n_reads, total_bases, first_read_idx,
                             records_ptr, seg_capacity, seg_counts_ptr, discarded_ptr=None):
        self._call("extract_by_owner_dev", self.h, bases_ptr, quals_ptr, offsets_ptr, n_reads, total_bases, first_read_idx,
                   discarded_ptr, records_ptr, seg_capacity, seg_counts_ptr)

    def insertRecordsDevice(self, records_ptr, n):
        self._call("insert_records_dev", self.h, records_ptr, n)

    # the whole owner exchange inside the library (kmr_exchange_*): RCCL called from C++, or a transport the host supplies
    @staticmethod
    def exchange_unique_id():
        """rank 0: the job's RCCL id (KMR_EXCHANGE_ID_BYTES bytes) to hand to every rank"""
        buf = (C.c_uint8 * 128)()
        lib = _lib.load()
        rc = lib.kmr_exchange_unique_id(C.cast(buf, C.c_void_p))
        if rc:
            raise KmerSpectrumError("kmr_exchange_unique_id failed (%d): %s" % (rc, lib.kmr_last_error(None).decode()))
        return bytes(buf)

    def exchange_init(self, unique_id):
        """collective: ncclCommInitRank(world_size, id, rank) on the handle's device (kmr_exchange_init)"""
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._call("exchange_init", self.h, C.cast(buf, C.c_void_p))

    def exchange_init_transport(self, allgather_u64, alltoallv_dev):
        """kmr_exchange_init_transport with Python callables (tests: gloo).  allgather_u64(mine: list[int]) -> list of world rows;
        alltoallv_dev(send_ptr, send_off, send_bytes, recv_ptr, recv_off, recv_bytes, stream) with lists of world ints."""
        world = self.cfg.world_size
        u64p = C.POINTER(C.c_uint64)
        AG = C.CFUNCTYPE(C.c_int, C.c_void_p, u64p, C.c_uint64, u64p)
        AV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, u64p, u64p, C.c_void_p, u64p, u64p, C.c_void_p)

        def ag(user, mine, n, out):
            try:
                rows = allgather_u64([int(mine[i]) for i in range(n)])
                for r in range(world):
                    for j in range(n):
                        out[r * n + j] = int(rows[r][j])
                return 0
            except Exception:          # noqa: BLE001 -- the exception cannot cross the C frame
                import traceback
                traceback.print_exc()
                return -1

        def av(user, send, soff, sbytes, recv, roff, rbytes, stream):
            try:
                alltoallv_dev(send, [int(soff[r]) for r in range(world)], [int(sbytes[r]) for r in range(world)],
                              recv, [int(roff[r]) for r in range(world)], [int(rbytes[r]) for r in range(world)], stream)
                return 0
            except Exception:          # noqa: BLE001
                import traceback
                traceback.print_exc()
                return -1

        class Transport(C.Structure):
            _fields_ = [("user", C.c_void_p), ("allgather_u64", AG), ("alltoallv_dev", AV)]
        self._transport = Transport(None, AG(ag), AV(av))          # keeps the callbacks alive as long as the handle
        self._call("exchange_init_transport", self.h, C.cast(C.pointer(self._transport), C.c_void_p))

    def exchange_add_reads(self, bases_ptr, quals_ptr, offsets_ptr, n_reads, total_bases, first_read_idx=0, discarded_ptr=None):
        """collective: one batch of this rank's reads through extract -> all-to-all -> insert at the owner (kmr_exchange_add_reads_dev)"""
        self._call("exchange_add_reads_dev", self.h, bases_ptr, quals_ptr, offsets_ptr, n_reads, total_bases, first_read_idx, discarded_ptr)

    def exchange_stats(self):
        b = C.c_uint64()
        ms = C.c_double()
        self._call("exchange_stats", self.h, C.byref(b), C.byref(ms))
        return {"bytes_to_peers": b.value, "alltoall_ms": ms.value}

    def set_stream_origin(self, ordinal):
        """position in the whole input of the next base this handle is fed (kmr_set_stream_origin)"""
        self._call("set_stream_origin", self.h, int(ordinal))

    # owner exchange of super-k-mer lists (build_mode 3; kmernator_amd.distributed.build_partitioned_superkmers)
    def sk_exchange_begin(self):
        """from here on the handle's reads go through the list exchange: every owner's k-mers are kept (kmr_sk_exchange_begin)"""
        self._call("sk_exchange_begin", self.h)

    def sk_exchange_counts(self):
        world = self.cfg.world_size
        chunks = np.zeros(world, dtype=np.uint64)
        granules = np.zeros(world, dtype=np.uint64)
        self._call("sk_exchange_counts", self.h, chunks.ctypes.data_as(C.POINTER(C.c_uint64)), granules.ctypes.data_as(C.POINTER(C.c_uint64)))
        return chunks, granules

    def sk_exchange_pack(self, data_ptr, meta_ptr, granule_offset, chunk_offset):
        go = np.ascontiguousarray(granule_offset, dtype=np.uint64)
        co = np.ascontiguousarray(chunk_offset, dtype=np.uint64)
        self._call("sk_exchange_pack_dev", self.h, data_ptr, meta_ptr, go.ctypes.data_as(C.POINTER(C.c_uint64)), co.ctypes.data_as(C.POINTER(C.c_uint64)))

    def sk_exchange_uniform(self):
        """kmr_sk_exchange_uniform: kind << 32 | weight bits of this rank's records (what a sender tells the owners)"""
        v = C.c_uint64()
        self._call("sk_exchange_uniform", self.h, C.byref(v))
        return v.value

    def sk_exchange_peer_uniform(self, state):
        """kmr_sk_exchange_peer_uniform: fold a sender's state in, before adopting its chunks"""
        self._call("sk_exchange_peer_uniform", self.h, int(state))

    def sk_exchange_range(self, list_lo=0, list_hi=0xFFFFFFFFFFFFFFFF):
        """kmr_sk_exchange_range: the lists the next sk_exchange_counts / sk_exchange_pack are about"""
        self._call("sk_exchange_range", self.h, int(list_lo), int(list_hi))

    def count_lists_prefix(self, min_depth, list_hi):
        """kmr_count_lists_prefix: count this handle's lists below list_hi now (asynchronously); finalize(min_depth) does the rest"""
        self._call("count_lists_prefix", self.h, int(min_depth), int(list_hi))

    def sk_exchange_adopt(self, data_ptr, meta_ptr, n_chunks, n_granules):
        self._call("sk_exchange_adopt_dev", self.h, data_ptr, meta_ptr, n_chunks, n_granules)

    # the device steps of the distributed scoreAndTrimReads (kmernator_amd.distributed.score_partitioned); tensors are torch
    # tensors on this handle's device
    def lookup_requests(self, bases, offsets, lo, hi, total_bases, keys, pos, seg_capacity, seg_counts):
        self._call("lookup_requests_dev", self.h, bases.data_ptr(), offsets.data_ptr() + 8 * lo, hi - lo, total_bases,
                   keys.data_ptr(), pos.data_ptr(), seg_capacity, seg_counts.data_ptr())

    def lookup_keys(self, keys, n, counts):
        self._call("lookup_keys_dev", self.h, keys.data_ptr(), n, counts.data_ptr())

    def scatter_counts(self, counts, pos, n, position_counts):
        self._call("scatter_counts_dev", self.h, counts.data_ptr(), pos.data_ptr(), n, position_counts.data_ptr())

    def score_counts(self, bases, offsets, n_reads, position_counts, minimum_kmer_score, scoring_type="MEDIAN"):
        to = np.zeros(n_reads, dtype=np.uint32)
        tl = np.zeros(n_reads, dtype=np.uint32)
        sc = np.zeros(n_reads, dtype=np.float32)
        wt = np.zeros(n_reads, dtype=np.uint8)
        if n_reads:
            self._call("score_counts_dev", self.h, bases.data_ptr(), offsets.data_ptr(), n_reads, position_counts.data_ptr(), float(minimum_kmer_score),
                       self.SCORING[scoring_type], to.ctypes.data_as(C.POINTER(C.c_uint32)), tl.ctypes.data_as(C.POINTER(C.c_uint32)),
                       sc.ctypes.data_as(C.POINTER(C.c_float)), wt.ctypes.data_as(C.POINTER(C.c_uint8)))
        return to, tl, sc, wt.astype(bool)

    def stream(self):
        return self.lib.kmr_stream(self.h)

    # -- queries
    def stats(self):
        s = KmrStats()
        self._call("get_stats", self.h, C.byref(s))
        return s.as_dict()

    def getRawKmers(self):
        return self.stats()["raw_kmers"]

    def getRawGoodKmers(self):
        return self.stats()["raw_good_kmers"]

    def getUniqueKmers(self):
        return self.stats()["unique_kmers"]

    def getSingletonKmers(self):
        return self.stats()["singleton_kmers"]

    def num_buckets(self, which):
        v = C.c_uint64()
        self._call("num_buckets", self.h, which, C.byref(v))
        return v.value

    def getCount(self, packed_kmers):
        """KmerSpectrum::getCount(kmer, false) for packed canonical k-mers [n, kb]."""
        keys = _u8(packed_kmers)
        n = keys.size // self.kb
        out = np.zeros(n, dtype=np.uint32)
        if n:
            self._call("lookup", self.h, keys.ctypes.data_as(C.POINTER(C.c_uint8)), n, out.ctypes.data_as(C.POINTER(C.c_uint32)))
        return out

    lookup = getCount

    def getCountsForReads(self, bases, offsets):
        """Per-position counts of every read (ReadSelector::setKmerValues)."""
        bases = _u8(bases)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        lens = (offsets[1:] - offsets[:-1]).astype(np.int64)
        nk = np.maximum(lens - self.k + 1, 0).astype(np.uint64)
        out_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(nk, out=out_off[1:])
        out = np.zeros(int(out_off[-1]), dtype=np.uint32)
        if n:
            self._call("lookup_reads", self.h, bases.ctypes.data_as(C.c_void_p), offsets.ctypes.data_as(C.POINTER(C.c_uint64)), n,
                       out.ctypes.data_as(C.POINTER(C.c_uint32)), out_off.ctypes.data_as(C.POINTER(C.c_uint64)))
        return out, out_off

    SCORING = {"SUM": 0, "MEDIAN": 1, "MIN": 2, "MAX": 3, "AVG": 4}

    def scoreAndTrimReads(self, bases, offsets, minimum_kmer_score, scoring_type="MEDIAN"):
        """ReadSelector::scoreAndTrimReads: (trim_offset, trim_length, score, was_trimmed) per read."""
        bases = _u8(bases)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        to = np.zeros(n, dtype=np.uint32)
        tl = np.zeros(n, dtype=np.uint32)
        sc = np.zeros(n, dtype=np.float32)
        wt = np.zeros(n, dtype=np.uint8)
        if n:
            self._call("score_reads", self.h, bases.ctypes.data_as(C.c_void_p), offsets.ctypes.data_as(C.POINTER(C.c_uint64)), n,
                       float(minimum_kmer_score), self.SCORING[scoring_type], to.ctypes.data_as(C.POINTER(C.c_uint32)),
                       tl.ctypes.data_as(C.POINTER(C.c_uint32)), sc.ctypes.data_as(C.POINTER(C.c_float)), wt.ctypes.data_as(C.POINTER(C.c_uint8)))
        return to, tl, sc, wt.astype(bool)

    def scoreAndTrimReadSet(self, read_set, minimum_kmer_score, scoring_type="MEDIAN"):
        """scoreAndTrimReads on a device-resident ReadSet (kmr_score_read_batch)"""
        n = read_set.n
        to = np.zeros(n, dtype=np.uint32)
        tl = np.zeros(n, dtype=np.uint32)
        sc = np.zeros(n, dtype=np.float32)
        wt = np.zeros(n, dtype=np.uint8)
        if n:
            self._call("score_read_batch", self.h, read_set.r, float(minimum_kmer_score), self.SCORING[scoring_type], to.ctypes.data_as(C.POINTER(C.c_uint32)),
                       tl.ctypes.data_as(C.POINTER(C.c_uint32)), sc.ctypes.data_as(C.POINTER(C.c_float)), wt.ctypes.data_as(C.POINTER(C.c_uint8)))
        return to, tl, sc, wt.astype(bool)

    def histogram(self, nbins=256):
        counts = np.zeros(nbins, dtype=np.uint64)
        weights = np.zeros(nbins, dtype=np.float64)
        self._call("count_histogram", self.h, counts.ctypes.data_as(C.POINTER(C.c_uint64)),
                   weights.ctypes.data_as(C.POINTER(C.c_double)), nbins)
        return counts, weights

    def digest(self, which=KMR_MAP_WEAK):
        """kmr_map_digest: order-independent digest of a finalized map (sums of rank / part digests = the whole spectrum's)"""
        d = _lib.KmrDigest()
        self._call("map_digest", self.h, which, C.byref(d))
        return d.as_dict()

    def getHistogram(self, zoom_max=256, log_base=2.0):
        """KmerSpectrum::getHistogram (src/KmerSpectrum.h:1066-1071): Histogram(256).set(*this) -> Histogram"""
        nb = self.lib.kmr_histogram_bins(zoom_max)
        visits = np.zeros(nb, dtype=np.uint64)
        vcount = np.zeros(nb, dtype=np.uint64)
        vweight = np.zeros(nb, dtype=np.float64)
        self._call("histogram", self.h, zoom_max, log_base, visits.ctypes.data_as(C.POINTER(C.c_uint64)),
                   vcount.ctypes.data_as(C.POINTER(C.c_uint64)), vweight.ctypes.data_as(C.POINTER(C.c_double)), nb)
        return Histogram(zoom_max, log_base, visits, vcount, vweight)

    def getSizeTracker(self, force_last=True):
        """KmerSpectrum::getSizeTracker (src/KmerSpectrum.h:902-904) after trackSpectrum(force_last): the size history, sampled at
        read boundaries (kmr_size_tracker; needs kmr_config.size_tracker = 1)"""
        n = C.c_uint64()
        self._call("size_tracker", self.h, 1 if force_last else 0, None, 0, C.byref(n))
        el = np.zeros((n.value, 4), dtype=np.uint64)
        if n.value:
            self._call("size_tracker", self.h, 1 if force_last else 0, el.ctypes.data_as(C.POINTER(C.c_uint64)), n.value, C.byref(n))
        return SizeTracker(el)

    # -- export / restore in the reference's mmap format
    def image(self, which=KMR_MAP_WEAK):
        sz = C.c_uint64()
        self._call("image_size", self.h, which, C.byref(sz))
        buf = np.zeros(sz.value, dtype=np.uint8)
        self._call("write_image", self.h, which, buf.ctypes.data_as(C.c_void_p), sz.value)
        return buf

    def load_image(self, which, buf):
        buf = _u8(buf)
        self._call("load_image", self.h, which, buf.ctypes.data_as(C.c_void_p), buf.size)

    def merge_image(self, which, buf):
        """restore-and-merge of one stored part (KmerSpectrum::buildKmerSpectrumInParts, src/KmerSpectrum.h:1871-1884)"""
        buf = _u8(buf)
        self._call("merge_image", self.h, which, buf.ctypes.data_as(C.c_void_p), buf.size)

    def storeMmap(self, filename, min_depth=2):
        """KmerSpectrum::storeMmap: <filename> (weak) and <filename>-singleton when min_depth <= 1."""
        self.image(KMR_MAP_WEAK).tofile(filename)
        if min_depth <= 1:
            self.image(KMR_MAP_SINGLETON).tofile(filename + "-singleton")

    def restoreMmap(self, filename):
        """KmerSpectrum::restoreMmap"""
        loaded = False
        if os.path.exists(filename) and os.path.getsize(filename) > 0:
            self.load_image(KMR_MAP_WEAK, np.fromfile(filename, dtype=np.uint8))
            loaded = True
        s = filename + "-singleton"
        if os.path.exists(s) and os.path.getsize(s) > 0:
            self.load_image(KMR_MAP_SINGLETON, np.fromfile(s, dtype=np.uint8))
            loaded = True
        if not loaded:
            raise KmerSpectrumError("Terribly sorry but there were no kmer spectrum mmap files at: %s*" % filename)

    def dumpCounts(self, filename, min_depth):
        self._call("dump_mercount", self.h, filename.encode(), min_depth)

    def dumpGraphs(self, filename, min_depth):
        self._call("dump_mergraph", self.h, filename.encode(), min_depth)

    def kernel_time(self, which=0):
        ms, n = C.c_double(), C.c_uint64()
        self._call("kernel_time", self.h, which, C.byref(ms), C.byref(n))
        return ms.value, n.value

    def kernel_time_reset(self):
        self._call("kernel_time_reset", self.h)


class Histogram:
    """KmerSpectrum::Histogram (src/KmerSpectrum.h:909-1057): buckets filled on the device, finish()/toString() here."""

    def __init__(self, zoom_max, log_base, visits, visited_count, visited_weight):
        import math
        self.zoomMax, self.logBase = zoom_max, log_base
        self.logFactor = math.log(log_base)
        self.zoomLogSkip = int(math.log(zoom_max + 1.0) / self.logFactor - 1.0)
        self.visits, self.visitedCount, self.visitedWeight = visits, visited_count, visited_weight
        self.finish()

    def getBucketValue(self, idx):
        return idx if idx <= self.zoomMax else int(self.logBase ** float(idx + self.zoomLogSkip - self.zoomMax))

    def finish(self):
        """:986-1001"""
        self.cumulativeVisits = np.cumsum(self.visits[::-1])[::-1].copy()
        self.count = int(self.visits.sum())
        nz = np.nonzero(self.visits)[0]
        self.lastBucket = int(nz[-1]) if nz.size else 0
        self.totalCount = float(self.visitedCount[nz].astype(np.float64).sum()) if nz.size else 0.0
        # the reference sums from the last bucket down
        tw = 0.0
        for i in nz[::-1]:
            tw += float(self.visitedWeight[i])
        self.totalWeightedCount = tw

    def toString(self):
        """:1002-1035, std::fixed << std::setprecision(3)"""
        f = lambda x: "%.3f" % x
        div = lambda a, b: (a / b) if b else float("nan")
        out = ["Counts, Weights and Directions",
               "Counts:\t%d\t%s\t%s\t" % (self.count, f(self.totalCount), f(div(self.totalCount, self.count))),
               "Weights:\t%d\t%s\t%s\t%s" % (self.count, f(self.totalWeightedCount), f(div(self.totalWeightedCount, self.count)),
                                                f(div(self.totalWeightedCount, self.totalCount))),
               "",
               "Bucket\tCumulative\tUnique\t%Unique\tCount\t%Count\tWeight\tQualProb\t%Weight"]
        for i in range(1, self.lastBucket + 1):
            v, c, w = int(self.visits[i]), int(self.visitedCount[i]), float(self.visitedWeight[i])
            out.append("%d\t%d\t%d\t%s\t%d\t%s\t\t%s\t%s\t%s\t" % (
                self.getBucketValue(i), int(self.cumulativeVisits[i]), v, f(div(100.0 * v, self.count)), c,
                f(div(100.0 * c, self.totalCount)), f(w), f(div(w, c)), f(div(100.0 * w, self.totalWeightedCount))))
        return "\n".join(out) + "\n"


class ReadSet:
    """Device-resident reads parsed from FASTQ text on the GPU: the reference's ReadSet::appendFastq path
    (src/ReadSet.cpp:311-345 over FastqStreamParser, src/ReadFileReader.h:768-835) incl. the Casava-1.8 filter and the
    quality-base detection of validateFastqStart (src/ReadSet.h:171-209).  Bound to the device of `spectrum`."""

    def __init__(self, spectrum, text, input_quality_base=0, store_comment=True):
        self.sp = spectrum
        self.text = bytes(text)
        r = C.c_void_p()
        buf = np.frombuffer(self.text, dtype=np.uint8)
        spectrum._call("ingest_fastq", spectrum.h, buf.ctypes.data_as(C.c_void_p) if buf.size else None, buf.size,
                       input_quality_base, 1 if store_comment else 0, C.byref(r))
        self.r = r
        n, tot, qb, nf = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint64()
        spectrum.lib.kmr_reads_info(self.r, C.byref(n), C.byref(tot), C.byref(qb), C.byref(nf))
        self.n, self.total_bases, self.input_quality_base, self.filtered = n.value, tot.value, qb.value, nf.value

    @classmethod
    def _adopt(cls, spectrum, text, handle):
        """wrap a kmr_reads the library produced from another batch (names still point into `text`)"""
        self = cls.__new__(cls)
        self.sp, self.text, self.r = spectrum, text, handle
        n, tot, qb, nf = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint64()
        spectrum.lib.kmr_reads_info(self.r, C.byref(n), C.byref(tot), C.byref(qb), C.byref(nf))
        self.n, self.total_bases, self.input_quality_base, self.filtered = n.value, tot.value, qb.value, nf.value
        return self

    @classmethod
    def from_arrays(cls, spectrum, bases, quals, offsets):
        """reads the host already holds (uint8 bases / quals scaled to the spectrum's quality base, uint64 offsets[n+1])"""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        r = C.c_void_p()
        spectrum._call("reads_from_host", spectrum.h, bases.ctypes.data_as(C.c_void_p), quals.ctypes.data_as(C.c_void_p),
                       offsets.ctypes.data_as(C.POINTER(C.c_uint64)), offsets.size - 1, C.byref(r))
        return cls._adopt(spectrum, b"", r)

    @classmethod
    def from_twobit(cls, spectrum, twobit, twobit_offsets, offsets, quals=None, uniform_quality=0, markups=None):
        """reads the host keeps as the reference's Read does (2-bit packed bases, every read on bytes of its own; markups = (positions, chars,
        offsets[n+1]); qualities an array, one character for all, or none): kmr_reads_from_twobit"""
        tw = np.ascontiguousarray(twobit, dtype=np.uint8)
        to = np.ascontiguousarray(twobit_offsets, dtype=np.uint64)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        q = None if quals is None else np.ascontiguousarray(quals, dtype=np.uint8)
        mp = mc = mo = None
        if markups is not None:
            mp = np.ascontiguousarray(markups[0], dtype=np.uint32); mc = np.ascontiguousarray(markups[1], dtype=np.uint8); mo = np.ascontiguousarray(markups[2], dtype=np.uint64)
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        r = C.c_void_p()
        spectrum._call("reads_from_twobit", spectrum.h, vp(tw), vp(to), vp(offsets), vp(mo), vp(mp), vp(mc), vp(q), int(uniform_quality), offsets.size - 1, C.byref(r))
        return cls._adopt(spectrum, b"", r)

    def twobit(self):
        """(packed bases, offsets[n+1], markup positions, markup chars, markup offsets[n+1]): every read as
        TwoBitSequence::compressSequence packs it (src/TwoBitSequence.cpp:242-269), on the device"""
        nb, nm = C.c_uint64(), C.c_uint64()
        u64 = C.POINTER(C.c_uint64)
        self.sp._call("reads_twobit", self.sp.h, self.r, None, 0, None, None, None, 0, None, C.byref(nb), C.byref(nm))
        tw = np.zeros(max(1, nb.value), dtype=np.uint8)
        to = np.zeros(self.n + 1, dtype=np.uint64)
        mp = np.zeros(max(1, nm.value), dtype=np.uint32)
        mc = np.zeros(max(1, nm.value), dtype=np.uint8)
        mo = np.zeros(self.n + 1, dtype=np.uint64)
        self.sp._call("reads_twobit", self.sp.h, self.r, tw.ctypes.data_as(C.c_void_p), tw.size, to.ctypes.data_as(u64), mp.ctypes.data_as(C.POINTER(C.c_uint32)),
                      mc.ctypes.data_as(C.c_void_p), mp.size, mo.ctypes.data_as(u64), C.byref(nb), C.byref(nm))
        return tw[:nb.value], to, mp[:nm.value], mc[:nm.value], mo

    def getSize(self):
        return self.n

    def arrays(self):
        """(bases, quals, offsets, names) on the host"""
        b = np.zeros(self.total_bases, dtype=np.uint8)
        q = np.zeros(self.total_bases, dtype=np.uint8)
        o = np.zeros(self.n + 1, dtype=np.uint64)
        no = np.zeros(max(1, self.n), dtype=np.uint64)
        nl = np.zeros(max(1, self.n), dtype=np.uint32)
        rc = self.sp.lib.kmr_reads_copy(self.r, b.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.POINTER(C.c_uint64)),
                                        no.ctypes.data_as(C.POINTER(C.c_uint64)), nl.ctypes.data_as(C.POINTER(C.c_uint32)))
        if rc != 0:
            raise KmerSpectrumError("kmr_reads_copy: %s" % _lib.STATUS.get(rc, rc))
        names = [self.text[int(no[i]):int(no[i]) + int(nl[i])] for i in range(self.n)]
        return b, q, o, names

    def close(self):
        if getattr(self, "r", None):
            self.sp.lib.kmr_reads_free(self.r)
            self.r = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FilterKnownOddities:
    """The artifact filter FilterReads runs before the spectrum build (src/FilterKnownOddities.h, apps/FilterReads.cpp:107-118):
    quality-run trimming plus a screen of every 4th 24-mer of a read against the artifact sequences (and their substitution
    neighbours), on the device.  `fasta` = the sequences as FASTA text (the reference embeds its own table); keyword
    arguments are the fields of kmr_artifact_config (edit_distance, min_quality, fastq_start_char, min_read_length, ...)."""

    def __init__(self, spectrum, fasta, **kw):
        self.sp = spectrum
        cfg = _lib.KmrArtifactConfig()
        spectrum.lib.kmr_artifact_config_init(C.byref(cfg))
        cfg.fastq_start_char = spectrum.cfg.fastq_start_char
        cfg.min_quality = spectrum.cfg.min_quality_score
        for name, v in kw.items():
            if not hasattr(cfg, name):
                raise TypeError("unknown artifact filter option %r" % name)
            setattr(cfg, name, v)
        self.cfg = cfg
        fasta = bytes(fasta)
        f = C.c_void_p()
        spectrum._call("artifact_filter_create", spectrum.h, C.byref(cfg), fasta, len(fasta), C.byref(f))
        self.f = f
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint32()
        spectrum.lib.kmr_artifact_filter_info(self.f, C.byref(a), C.byref(b), C.byref(c))
        self.n_sequences, self.n_filter_kmers, self.remaining_edits = a.value, b.value, c.value

    def entries(self):
        keys = np.zeros(self.n_filter_kmers, dtype=np.uint64)
        vals = np.zeros(self.n_filter_kmers, dtype=np.uint32)
        rc = self.sp.lib.kmr_artifact_filter_entries(self.f, keys.ctypes.data_as(C.POINTER(C.c_uint64)), vals.ctypes.data_as(C.POINTER(C.c_uint32)), keys.size)
        if rc != 0:
            raise KmerSpectrumError("kmr_artifact_filter_entries: %s" % _lib.STATUS.get(rc, rc))
        return keys, vals

    def applyFilter(self, reads, mate=None, want_reads=True):
        """applyFilter(ReadSet&) (:663-733).  Returns (results, filtered ReadSet): results = dict of per-read arrays value,
        min_pass, max_pass, action (0 untouched / 1 trimmed / 2 discarded), remnant_off, remnant_len; the new ReadSet holds the
        trimmed reads in place and the rescued remnants behind them."""
        n = reads.n
        res = {k: np.zeros(max(1, n), dtype=np.uint32) for k in ("value", "min_pass", "max_pass", "remnant_off", "remnant_len")}
        res["action"] = np.zeros(max(1, n), dtype=np.uint8)
        u32 = C.POINTER(C.c_uint32)
        m = None
        if mate is not None:
            mate = np.ascontiguousarray(mate, dtype=np.int64)
            assert mate.size == n
            m = mate.ctypes.data_as(C.POINTER(C.c_int64))
        out = C.c_void_p()
        self.sp._call("artifact_filter_apply", self.sp.h, self.f, reads.r, m, res["value"].ctypes.data_as(u32), res["min_pass"].ctypes.data_as(u32),
                      res["max_pass"].ctypes.data_as(u32), res["action"].ctypes.data_as(C.POINTER(C.c_uint8)), res["remnant_off"].ctypes.data_as(u32),
                      res["remnant_len"].ctypes.data_as(u32), C.byref(out) if want_reads else None)
        res = {k: v[:n] for k, v in res.items()}
        return res, (ReadSet._adopt(self.sp, reads.text, out) if want_reads else None)

    def close(self):
        if getattr(self, "f", None):
            self.sp.lib.kmr_artifact_filter_free(self.f)
            self.f = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def synth_reads_device(torch, seed, first_read, n_reads, read_len, genome_len, noisy, device):
    """kmr_synth_reads_dev into torch tensors on `device`: (bases u8, quals u8, offsets i64).  The buffers carry 64 spare bytes behind
    the last read, as the device entry points of the build ask for."""
    lib = _lib.load()
    with torch.cuda.device(device):
        bases = torch.zeros(n_reads * read_len + 64, dtype=torch.uint8, device=device)
        quals = torch.zeros(n_reads * read_len + 64, dtype=torch.uint8, device=device)
        offsets = torch.empty(n_reads + 1, dtype=torch.int64, device=device)
        torch.cuda.synchronize()
        rc = lib.kmr_synth_reads_dev(seed, first_read, n_reads, read_len, genome_len, 1 if noisy else 0, bases.data_ptr(), quals.data_ptr(), offsets.data_ptr())
        if rc:
            raise KmerSpectrumError("kmr_synth_reads_dev: %s" % _lib.STATUS.get(rc, rc))
    return bases, quals, offsets
