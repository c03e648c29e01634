"""world_size-2 (and 3) gloo runs of the owner-partitioned exchange
(kmernator_amd/distributed.py::exchange_records) on the CPU.  The oracle stands in for
the two device steps (extract-by-owner, insert-records); the exchange code under test is
the product's.  Each rank owns a slice of the reads; afterwards the union of the per-rank
spectra must equal the single-partition spectrum and every k-mer must sit on its lookup3
owner (src/Kmer.h:2284-2295)."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import (GOLDEN, KMR_VALUE_EXT, OracleSpectrum, default_config, oracle_extract_by_owner, oracle_lib, read_fastq)


def _worker(rank, world, port, tmp, k, ext):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kmernator_amd.distributed import all_ranks_chunk_count, exchange_records
        from kmernator_amd import record_bytes
        rb_all = read_fastq(os.path.join(GOLDEN, "1000.fastq"))
        per = (rb_all.n + world - 1) // world
        lo, hi = rank * per, min(rb_all.n, (rank + 1) * per)
        kw = dict(value_kind=KMR_VALUE_EXT, min_weight=0.0, min_quality_score=2) if ext else {}
        cfg = default_config(k, fastq_start_char=64, estimated_raw_kmers=56000, rank=rank, world_size=world, **kw)
        spec = OracleSpectrum(cfg)
        recb = record_bytes(k, cfg.value_kind)
        chunk = 130                                   # uneven chunk counts across ranks on purpose
        n_chunks = (hi - lo + chunk - 1) // chunk if rank != world - 1 else 2
        total = all_ranks_chunk_count(n_chunks)
        seg_cap = 20000
        for c in range(total):
            a = lo + c * chunk
            b = min(hi, a + chunk) if c < n_chunks - 1 or rank != world - 1 else hi
            if c >= n_chunks or a >= hi:
                recs = np.zeros(world * seg_cap * recb, dtype=np.uint8)
                counts = np.zeros(world, dtype=np.uint64)
            else:
                recs, counts = oracle_extract_by_owner(cfg, rb_all.slice(a, b), seg_cap)
            recv, n = exchange_records(torch.from_numpy(recs), torch.from_numpy(counts.astype(np.int64)), seg_cap, recb)
            if n:
                spec.insert_records(recv.numpy(), n)
        spec.finalize(2)
        keys, cnt, dirb, w, extv = spec.entries()
        lib = oracle_lib()
        for kk in keys[:200]:
            assert lib.orc_distributed_thread_id(lib.orc_hash(kk.tobytes(), len(kk)), world) == rank
        spec.dump(os.path.join(tmp, "counts.%d" % rank), 2, False)
        if ext:
            spec.dump(os.path.join(tmp, "graph.%d" % rank), 2, True)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k,ext", [(2, 21, True), (3, 21, True), (2, 31, False)])
def test_exchange_union_equals_single(world, k, ext):
    port = 29500 + (os.getpid() % 2000) + world
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(world, port, tmp, k, ext), nprocs=world, join=True)
        got = []
        for r in range(world):
            got += open(os.path.join(tmp, "counts.%d" % r)).read().splitlines()
        if ext:
            exp = open(os.path.join(GOLDEN, "phix.mercount.m21")).read().splitlines()
            assert sorted(got) == sorted(exp)
            gg = []
            for r in range(world):
                gg += open(os.path.join(tmp, "graph.%d" % r)).read().splitlines()
            assert sorted(gg) == sorted(open(os.path.join(GOLDEN, "phix.mergraph.m21.D2")).read().splitlines())
        else:
            rb = read_fastq(os.path.join(GOLDEN, "1000.fastq"))
            s = OracleSpectrum(default_config(k, fastq_start_char=64, estimated_raw_kmers=46000))
            s.add_reads(rb)
            s.finalize(2)
            s.dump(os.path.join(tmp, "single"), 2, False)
            assert sorted(got) == sorted(open(os.path.join(tmp, "single")).read().splitlines())


# ---------------------------------------------------------------- distributed scoreAndTrimReads (f1, request / response form)

class _OracleScoringRank:
    """Stands in for the four device steps of kmernator_amd.distributed.score_partitioned on the CPU; the exchange logic under
    test is the product's.  Keys travel as the product sends them: KMR_KEY_WORDS u64 words, most significant first."""

    def __init__(self, cfg, spec):
        self.cfg, self.spec, self.k = cfg, spec, cfg.k
        self.kb = (cfg.k + 3) // 4
        self.words = (self.kb + 7) // 8
        self.lib = oracle_lib()

    def sync(self):
        pass

    def _words(self, keys):                              # [n, kb] bytes -> [n, words] int64
        pad = np.zeros((keys.shape[0], 8 * self.words), dtype=np.uint8)
        pad[:, :self.kb] = keys
        return pad.view(">u8").astype(np.uint64).view(np.int64)

    def lookup_requests(self, bases, offsets, lo, hi, total_bases, keys, pos, seg_capacity, seg_counts):
        from helpers import oracle_weighted_kmers
        b, off = bases.numpy(), offsets.numpy()
        fill = [0] * keys.shape[0]
        for r in range(lo, hi):
            seq = b[int(off[r]):int(off[r + 1])].tobytes()
            kk, w, _ = oracle_weighted_kmers(self.cfg, seq, None)
            ww = self._words(kk)
            for i in range(kk.shape[0]):
                if w[i] == 0:                            # a k-mer over a markup: never asked for
                    continue
                o = self.lib.orc_distributed_thread_id(self.lib.orc_hash(kk[i].tobytes(), self.kb), keys.shape[0])
                keys[o, fill[o]] = torch.from_numpy(ww[i].copy())
                pos[o, fill[o]] = int(off[r]) + i
                fill[o] += 1
        seg_counts.copy_(torch.tensor(fill, dtype=torch.int64))

    def lookup_keys(self, keys, n, counts):
        kw = keys.numpy().reshape(n, self.words).view(np.uint64).astype(">u8").view(np.uint8).reshape(n, 8 * self.words)[:, :self.kb]
        counts.copy_(torch.from_numpy(self.spec.lookup(np.ascontiguousarray(kw)).astype(np.int32)).reshape(n, 1))

    def scatter_counts(self, counts, pos, n, position_counts):
        position_counts[pos[:n].long()] = counts.reshape(-1)[:n]

    def score_counts(self, bases, offsets, n_reads, position_counts, minimum_kmer_score, scoring_type="MEDIAN"):
        from refsemantics import score_and_trim
        b, off, pc = bases.numpy(), offsets.numpy(), position_counts.numpy()
        out = []
        for r in range(n_reads):
            seq = b[int(off[r]):int(off[r + 1])].tobytes()
            nk = max(0, len(seq) - self.k + 1)
            out.append(score_and_trim(pc[int(off[r]):int(off[r]) + nk], seq, self.k, minimum_kmer_score, scoring_type))
        return out


def _score_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kmernator_amd.distributed import score_partitioned
        k = 31
        rb_all = read_fastq(os.path.join(GOLDEN, "1000.fastq"))
        cfg = default_config(k, fastq_start_char=64, estimated_raw_kmers=46000, rank=rank, world_size=world)
        spec = OracleSpectrum(cfg)
        spec.add_reads(rb_all)                               # the oracle keeps the k-mers this rank owns
        spec.finalize(2)
        per = (rb_all.n + world - 1) // world
        lo, hi = rank * per, min(rb_all.n, (rank + 1) * per) if rank != world - 1 else rb_all.n - 7 * (world - 1)
        lo = lo if rank == 0 else lo - 7 * rank                # uneven slices: different chunk counts per rank
        mine = rb_all.slice(lo, hi)
        res = score_partitioned(_OracleScoringRank(cfg, spec), torch.from_numpy(mine.bases), torch.from_numpy(mine.offsets.astype(np.int64)), 2, "MEDIAN",
                                chunk_reads=90 + 40 * rank)
        with open(os.path.join(tmp, "labels.%d" % rank), "w") as f:
            for i, (to, tl, sc, wt) in enumerate(res):
                f.write("%d %s%s\n" % (lo + i, "Trim:%d+%d " % (to, tl) if wt else "", "MedianScore:%d" % int(sc + 0.5)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_score_partitioned_reproduces_golden_labels(world):
    """every rank scores its slice of test/1000.fastq against a spectrum split over `world` owners: the labels of the 949 reads
    without AFTrim in test/1000-Filtered.fastq, as from one spectrum"""
    port = 31500 + (os.getpid() % 2000) + world
    gold = read_fastq(os.path.join(GOLDEN, "1000-Filtered.fastq"))
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_score_worker, args=(world, port, tmp), nprocs=world, join=True)
        seen = set()
        for r in range(world):
            for line in open(os.path.join(tmp, "labels.%d" % r)).read().splitlines():
                idx, label = line.split(" ", 1)
                i = int(idx)
                seen.add(i)
                if b"AFTrim" not in gold.names[i]:
                    assert label.encode() == gold.names[i].split(b" ", 1)[1], (i, label, gold.names[i])
        assert len(seen) >= 1000 - 7 * world


def _sliced_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kmernator_amd.distributed import _all_to_all_sliced, reduce_histogram, reduce_stats
        # rank r sends (r + 1) * (d + 2) rows of 4 words to rank d, nothing to itself; row content names source, destination and index
        send_split = [0 if d == rank else (rank + 1) * (d + 2) * 7 for d in range(world)]
        recv_split = [0 if s == rank else (s + 1) * (rank + 2) * 7 for s in range(world)]
        rows = []
        for d in range(world):
            for i in range(send_split[d]):
                rows.append([rank, d, i, 1000 * rank + d])
        send = torch.tensor(rows, dtype=torch.int32).reshape(-1, 4)
        for max_rows in (None, 5, 64):      # one message, many slices, a few slices
            got = _all_to_all_sliced(send, send_split, recv_split, max_rows=max_rows)
            assert got.shape == (sum(recv_split), 4)
            at = 0
            for s in range(world):
                part = got[at:at + recv_split[s]]
                assert bool((part[:, 0] == s).all()) and bool((part[:, 1] == rank).all())
                assert part[:, 2].tolist() == list(range(recv_split[s]))
                at += recv_split[s]

        class _Spec:        # what reduce_stats / reduce_histogram ask of a spectrum
            def stats(self):
                return {"raw_kmers": 100 + rank, "unique_kmers": 10 * (rank + 1), "weak_entries": rank}

            def getHistogram(self, zoom_max, log_base):
                import kmernator_amd.spectrum as sp
                n = (1 << 16) + 2 + zoom_max
                v = np.zeros(n, dtype=np.uint64); c = np.zeros(n, dtype=np.uint64); w = np.zeros(n, dtype=np.float64)
                v[2 + rank] = 5; c[2 + rank] = 5 * (2 + rank); w[2 + rank] = 4.5
                v[7] = 1; c[7] = 7; w[7] = 6.5
                return sp.Histogram(zoom_max, log_base, v, c, w)

        st = reduce_stats(_Spec())
        assert st == {"raw_kmers": sum(100 + r for r in range(world)), "unique_kmers": sum(10 * (r + 1) for r in range(world)), "weak_entries": sum(range(world))}
        h = reduce_histogram(_Spec(), 255, 2.0)
        assert int(h.visits[7]) == world and int(h.visitedCount[7]) == 7 * world and abs(float(h.visitedWeight[7]) - 6.5 * world) < 1e-9
        assert all(int(h.visits[2 + r]) == 5 for r in range(world)) and h.count == 6 * world
        open(os.path.join(tmp, "ok.%d" % rank), "w").write(h.toString())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sliced_all_to_all_and_the_reduces(world):
    """The wire of the super-k-mer exchange (rows grouped by destination, cut into slices below RCCL's message limit, every rank
    running the same number of slices) and the job-wide sums of statistics and histograms (MPIHistogram::reduce,
    src/DistributedFunctions.h:513-535), over gloo; every rank must print the same histogram table."""
    port = 33300 + (os.getpid() % 1500) + world
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_sliced_worker, args=(world, port, tmp), nprocs=world, join=True)
        texts = [open(os.path.join(tmp, "ok.%d" % r)).read() for r in range(world)]
        assert all(t == texts[0] for t in texts) and texts[0].startswith("Counts, Weights and Directions")


# ---------------------------------------------------------------- the list-chunk exchange driver (build_mode 3, N > 1) on the CPU
class _FakeListSpectrum:
    """Stands in for a build_mode 3 handle in build_partitioned_superkmers: `extracting` a piece makes, for every other owner, a
    known number of chunks with a known fill whose granules carry (sender, owner, piece, chunk, granule); pack lays them out as the
    library does (owner after owner, data and meta in the same order), adopt records what arrives.  What is tested is the driver:
    pieces, offsets, the two counts exchanges, the sliced all-to-alls, and that meta and data still belong together at the owner."""

    def __init__(self, rank, world):
        self.rank, self.world, self.piece, self.pending, self.adopted, self.origin, self.begun = rank, world, -1, None, [], [], False
        self.peer_states = []
        self.calls = []

    def sk_exchange_begin(self):
        self.begun = True

    def set_stream_origin(self, o):
        self.origin.append(int(o))

    def buildKmerSpectrumDevice(self, b, q, o, n, total, first):
        self.piece += 1
        self.pending = {r: [((self.rank * 7 + r * 3 + self.piece * 5 + c) % 64) + 1 for c in range((self.rank + 2 * r + self.piece) % 4 + (1 if n else 0))]
                        for r in range(self.world)}

    def sk_exchange_counts(self):
        if self.pending is None:
            self.pending = {r: [] for r in range(self.world)}
        ch = np.array([len(self.pending[r]) for r in range(self.world)], dtype=np.uint64)
        gr = np.array([sum(self.pending[r]) for r in range(self.world)], dtype=np.uint64)
        return ch, gr

    def sk_exchange_pack(self, data_ptr, meta_ptr, goff, coff):
        import ctypes as C
        for r in range(self.world):
            if r == self.rank:
                continue
            g, c = goff[r], coff[r]
            for ci, fill in enumerate(self.pending[r]):
                meta = np.array([1000 * r + ci, fill], dtype=np.int32)
                C.memmove(meta_ptr + 8 * c, meta.ctypes.data, 8)
                gran = np.array([[self.rank, r, self.piece * 100 + ci, gi] for gi in range(fill)], dtype=np.int32)
                C.memmove(data_ptr + 16 * g, gran.ctypes.data, 16 * fill)
                g += fill
                c += 1
        self.pending = None

    def build_info(self, what):
        assert what == "lists"
        return 1000.0

    def sk_exchange_range(self, lo=0, hi=0xFFFFFFFFFFFFFFFF):
        self.calls.append(("range", int(lo), int(hi)))
        # a step of the list space: what the fake "holds" for the others is made anew per step, as if the step's lists were packed
        if (lo, hi) != (0, 0xFFFFFFFFFFFFFFFF) and self.calls.count(("range", int(lo), int(hi))) == 1 and len([c for c in self.calls if c[0] == "range"]) > 1:
            self.piece += 1
            self.pending = {r: [((self.rank * 7 + r * 3 + self.piece * 5 + c) % 64) + 1 for c in range((self.rank + 2 * r + self.piece) % 4 + 1)] for r in range(self.world)}

    def count_lists_prefix(self, min_depth, hi):
        self.calls.append(("count_prefix", int(min_depth), int(hi), len(self.adopted)))

    def sk_exchange_uniform(self):
        return (1 << 32) | (0x3f000000 + self.rank)      # "one weight", a different one on every rank, so that the owner can tell whose state it got

    def sk_exchange_peer_uniform(self, state):
        self.peer_states.append(int(state))

    def sk_exchange_adopt(self, data_ptr, meta_ptr, n_chunks, n_granules):
        import ctypes as C
        meta = np.zeros((n_chunks, 2), dtype=np.int32)
        data = np.zeros((n_granules, 4), dtype=np.int32)
        C.memmove(meta.ctypes.data, meta_ptr, 8 * n_chunks)
        C.memmove(data.ctypes.data, data_ptr, 16 * n_granules)
        self.adopted.append((meta, data))


def _list_worker(rank, world, port, tmp, pieces):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kmernator_amd.distributed import build_partitioned_superkmers
        n = 500 + 300 * rank
        offsets = torch.arange(n + 1, dtype=torch.int64) * 100
        bases = torch.zeros(int(offsets[-1]) + 64, dtype=torch.uint8)
        sp = _FakeListSpectrum(rank, world)
        stats = {}
        build_partitioned_superkmers(sp, bases, bases, offsets, first_read_idx=0, stats=stats, pieces=pieces)
        assert sp.begun and sp.piece == pieces - 1
        # global ordinals: this rank starts behind the bases of the lower ranks, and the origin does not move from piece to piece
        assert sp.origin == [sum(100 * (500 + 300 * r) for r in range(rank))] * pieces
        # what arrived: from every other rank and every piece exactly the chunks that rank made for this owner, data behind meta
        got = {}
        for meta, data in sp.adopted:
            at = 0
            for lid, fill in meta:
                g = data[at:at + fill]
                at += fill
                assert lid // 1000 == rank and len(g) == fill and np.all(g[:, 1] == rank) and np.array_equal(g[:, 3], np.arange(fill))
                sender, tag = int(g[0, 0]), int(g[0, 2])
                assert np.all(g[:, 0] == sender) and np.all(g[:, 2] == tag) and tag % 100 == lid % 1000
                got[(sender, tag)] = int(fill)
            assert at == len(data)
        want = {}
        for s in range(world):
            if s == rank:
                continue
            for p in range(pieces):
                fake = _FakeListSpectrum(s, world)
                fake.piece = p - 1
                fake.buildKmerSpectrumDevice(0, 0, 0, 1, 0, 0)
                for ci, fill in enumerate(fake.pending[rank]):
                    want[(s, p * 100 + ci)] = fill
        assert got == want
        # every sender that sent something told this owner its uniform-weight state (it rides in the upper bits of the chunk counts)
        senders = {s_ for (s_, _) in want}
        assert set(sp.peer_states) == {(1 << 32) | (0x3f000000 + s_) for s_ in senders}
        assert stats["bytes_to_peers"] > 0
        open(os.path.join(tmp, "ok.%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def _steps_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kmernator_amd.distributed import build_partitioned_superkmers
        n = 400 + 100 * rank
        offsets = torch.arange(n + 1, dtype=torch.int64) * 100
        bases = torch.zeros(int(offsets[-1]) + 64, dtype=torch.uint8)
        sp = _FakeListSpectrum(rank, world)
        build_partitioned_superkmers(sp, bases, bases, offsets, first_read_idx=0, list_steps=2, early_min_depth=2)
        # one extraction of the whole batch; the list space in two steps; the lower half counted after it was adopted and before the upper
        # half is; the range restored at the end
        assert sp.piece == 1                      # build (piece 0) + the second step's pack
        kinds = [c[0] for c in sp.calls]
        assert kinds == ["range", "range", "count_prefix", "range"]
        assert sp.calls[0][1:] == (0, 500) and sp.calls[1][1] == 500 and sp.calls[3][1:] == (0, 0xFFFFFFFFFFFFFFFF)
        assert sp.calls[2][1:3] == (2, 500) and sp.calls[2][3] == 1      # exactly one adopt (the lower half's) had happened
        assert len(sp.adopted) == 2
        open(os.path.join(tmp, "ok.%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_list_exchange_in_steps_driver(world):
    """build_partitioned_superkmers(list_steps=2, early_min_depth=2): the call sequence an owner sees -- the lower half of the list space
    adopted, counted early, then the upper half -- over gloo with a stand-in spectrum"""
    port = 31100 + (os.getpid() % 500) + 10 * world
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_steps_worker, args=(world, port, tmp), nprocs=world, join=True)
        assert all(os.path.exists(os.path.join(tmp, "ok.%d" % r)) for r in range(world))


@pytest.mark.parametrize("world,pieces", [(2, 1), (3, 3), (2, 4)])
def test_list_chunk_exchange_driver(world, pieces):
    port = 29900 + (os.getpid() % 500) + 10 * world + pieces
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_list_worker, args=(world, port, tmp, pieces), nprocs=world, join=True)
        assert all(os.path.exists(os.path.join(tmp, "ok.%d" % r)) for r in range(world))
