"""kmr_exchange_*: the owner exchange driven from inside the library (the C / C++ host's path to N GPUs, no Python driver).

* one rank over RCCL (the library dlopens librccl and makes its own communicator): the exchange path must give the plain build's
  maps byte for byte -- lists (build_mode 0 -> 3), k-mer records (build_mode 2) and extension values;
* 2 and 3 ranks sharing this box's one GPU: RCCL refuses two ranks on a device, so the ranks hand the library a transport
  (kmr_exchange_init_transport) made of gloo collectives staged through host memory -- the driver's own logic (global ordinals,
  counts matrix, offsets, slices, adopt / insert) is the code an N-GPU job runs, only the two collectives differ."""
import ctypes as C
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import KMR_MAP_WEAK, KMR_VALUE_EXT, parse_image, synth_reads

pytestmark = pytest.mark.gpu

N_READS, READ_LEN = 60000, 150


SKEW = False          # set by the skewed-input test in the parent AND (through mp.spawn's pickled args) in the workers


def _reads(skew=False):
    rb = synth_reads(N_READS, read_len=READ_LEN, genome_len=5 * N_READS, seed=21, quality="noisy", n_rate=0.001)
    if skew:
        # half of the reads are poly-A: one k-mer, one owner, which then gets ~3/4 of every batch's records -- more than the
        # segment a rank sets aside for an owner (its share + 25 %), so the sender has to grow its segments and extract again
        bases = rb.bases.copy().reshape(N_READS, READ_LEN)
        bases[::2] = ord("A")
        quals = rb.quals.copy().reshape(N_READS, READ_LEN)
        quals[::2] = ord("I")
        rb = type(rb).from_arrays(bases.reshape(-1), quals.reshape(-1), rb.offsets)
    return rb


def _dev(rb, dev):
    tb = torch.from_numpy(np.concatenate([rb.bases, np.zeros(64, np.uint8)])).to(dev)
    tq = torch.from_numpy(np.concatenate([rb.quals, np.zeros(64, np.uint8)])).to(dev)
    to = torch.from_numpy(rb.offsets.astype(np.int64)).to(dev)
    return tb, tq, to


@pytest.mark.parametrize("mode,value_kind,k", [(0, None, 31), (2, None, 31), (0, KMR_VALUE_EXT, 21), (0, None, 51)])
def test_one_rank_over_rccl_equals_the_plain_build(mode, value_kind, k):
    import kmernator_amd as ka
    dev = torch.device("cuda", 0)
    rb = _reads()
    kw = dict(estimated_raw_kmers=N_READS * (READ_LEN - k + 1), device=0, build_mode=mode)
    if value_kind is not None:
        kw["value_kind"] = value_kind
    plain = ka.KmerSpectrum(ka.default_config(k, **kw))
    xch = ka.KmerSpectrum(ka.default_config(k, rank=0, world_size=1, **kw))
    xch.exchange_init(ka.KmerSpectrum.exchange_unique_id())
    cut = N_READS // 3
    for lo, hi in ((0, cut), (cut, N_READS)):
        part = rb.slice(lo, hi)
        plain.buildKmerSpectrum(part.bases, part.quals, part.offsets, first_read_idx=lo)
        tb, tq, to = _dev(part, dev)
        xch.exchange_add_reads(tb.data_ptr(), tq.data_ptr(), to.data_ptr(), hi - lo, int(part.offsets[-1]), lo)
    xch.exchange_add_reads(None, None, None, 0, 0, N_READS)          # a rank that ran out of reads still takes part
    plain.finalize(2)
    xch.finalize(2)
    assert plain.stats() == xch.stats()
    if mode == 0 and value_kind is None:
        assert np.array_equal(plain.image(KMR_MAP_WEAK), xch.image(KMR_MAP_WEAK))          # lists: global ordinals, nothing depends on arrival
    else:
        # k-mer records are stamped in the order they arrive at the owner (as in the reference's MPI build): keys and counts are
        # those of the plain build, the first sighting's direction may be another occurrence's
        vbytes = 60 if value_kind == KMR_VALUE_EXT else 12
        _, _, a = parse_image(plain.image(KMR_MAP_WEAK), plain.kb, vbytes)
        _, _, b = parse_image(xch.image(KMR_MAP_WEAK), xch.kb, vbytes)
        for (ka_, va), (kb_, vb) in zip(a, b):
            assert np.array_equal(ka_, kb_)
            if len(va):
                assert np.array_equal(va[:, :2], vb[:, :2])
    assert xch.exchange_stats()["bytes_to_peers"] == 0


def _slice(rank, world):
    per = (N_READS + world - 1) // world
    lo = rank * per
    hi = min(N_READS, (rank + 1) * per + (3000 if rank == 0 else 0))          # rank 0 also takes some of rank 1's reads: uneven shares
    return lo, hi


def _batches(rank, world):
    lo, hi = _slice(rank, world)
    n = hi - lo
    cuts = [0, n // 2, n] if rank else [0, n // 3, n]          # two batches per rank, of different sizes
    return list(zip(cuts[:-1], cuts[1:]))


def _worker(rank, world, port, tmp, mode, k, skew=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kmernator_amd as ka
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        dev = torch.device("cuda", 0)

        def allgather(mine):
            rows = [None] * world
            dist.all_gather_object(rows, list(mine))
            return rows

        def alltoallv(send, soff, sbytes, recv, roff, rbytes, stream):
            assert hip.hipStreamSynchronize(stream) == 0
            assert sbytes[rank] == 0 and rbytes[rank] == 0 and max(sbytes + rbytes) <= 1 << 30
            outs = []
            for r in range(world):
                buf = np.empty(sbytes[r], dtype=np.uint8)
                if sbytes[r]:
                    assert hip.hipMemcpy(buf.ctypes.data, send + soff[r], sbytes[r], 2) == 0
                outs.append(torch.from_numpy(buf))
            ins = [torch.empty(rbytes[r], dtype=torch.uint8) for r in range(world)]
            # pairwise sends (gloo's all_to_all wants equal splits)
            reqs = []
            for r in range(world):
                if r != rank and sbytes[r]:
                    reqs.append(dist.isend(outs[r], r))
            for r in range(world):
                if r != rank and rbytes[r]:
                    dist.recv(ins[r], r)
            for q in reqs:
                q.wait()
            for r in range(world):
                if rbytes[r]:
                    assert hip.hipMemcpy(recv + roff[r], ins[r].numpy().ctypes.data, rbytes[r], 1) == 0

        lo, hi = _slice(rank, world)
        rb = _reads(skew).slice(lo, hi)
        sp = ka.KmerSpectrum(ka.default_config(k, estimated_raw_kmers=N_READS * (READ_LEN - k + 1), device=0, rank=rank, world_size=world, build_mode=mode))
        sp.exchange_init_transport(allgather, alltoallv)
        for a, b in _batches(rank, world):
            part = rb.slice(a, b)
            tb, tq, to = _dev(part, dev)
            sp.exchange_add_reads(tb.data_ptr(), tq.data_ptr(), to.data_ptr(), b - a, int(part.offsets[-1]), lo + a)
        sp.finalize(2)
        np.save(os.path.join(tmp, "image.%d.npy" % rank), sp.image(KMR_MAP_WEAK))
        st = sp.stats()
        np.save(os.path.join(tmp, "stats.%d.npy" % rank), np.array([st["raw_kmers"], st["raw_good_kmers"], st["weak_entries"], st["unique_kmers"], sp.exchange_stats()["bytes_to_peers"]], dtype=np.int64))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,k,skew", [(2, 0, 31, False), (3, 0, 51, False), (2, 2, 31, False), (2, 2, 31, True), (2, 0, 31, True)])
def test_ranks_sharing_one_gpu_through_a_host_transport(world, mode, k, skew):
    """skew: half of the reads are poly-A, so one owner receives far more than its share of every batch -- the k-mer record path
    has to grow its owner segments and extract again (no error, no hang: the decision is the sender's own), the lists take it as
    it comes"""
    import kmernator_amd as ka
    port = 33300 + (os.getpid() % 1500) + world + 7 * mode + (13 if skew else 0)
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(world, port, tmp, mode, k, skew), nprocs=world, join=True)
        rb = _reads(skew)
        multi = ka.KmerSpectrum(ka.default_config(k, estimated_raw_kmers=N_READS * (READ_LEN - k + 1), device=0))
        # the job's input order: round by round, and inside a round rank by rank (what kmr_exchange_add_reads_dev's ordinals say)
        for i in range(2):
            for r in range(world):
                lo, _ = _slice(r, world)
                a, b = _batches(r, world)[i]
                part = rb.slice(lo + a, lo + b)
                multi.buildKmerSpectrum(part.bases, part.quals, part.offsets)
        multi.finalize(2)
        ms = multi.stats()
        stats = sum(np.load(os.path.join(tmp, "stats.%d.npy" % r)) for r in range(world))
        assert (int(stats[1]), int(stats[2]), int(stats[3])) == (ms["raw_good_kmers"], ms["weak_entries"], ms["unique_kmers"]), (stats, ms)
        assert int(stats[4]) > 0
        _, _, whole = parse_image(multi.image(KMR_MAP_WEAK), multi.kb, 12)
        wk = np.concatenate([kk for kk, _ in whole if len(kk)])
        wv = np.concatenate([v for _, v in whole if len(v)])
        # counts everywhere; direction biases where the exchange carries global ordinals (the lists) -- through k-mer records the
        # first sighting is the first to arrive at the owner, as in the reference's MPI build
        cut = (lambda v: bytes(v[:2]) + bytes(v[8:10])) if mode != 2 else (lambda v: bytes(v[:2]))
        want = {bytes(kk): cut(v) for kk, v in zip(wk, wv)}
        seen = 0
        for r in range(world):
            _, _, buckets = parse_image(np.load(os.path.join(tmp, "image.%d.npy" % r)), multi.kb, 12)
            keys = np.concatenate([kk for kk, _ in buckets if len(kk)])
            vals = np.concatenate([v for _, v in buckets if len(v)])
            for kk, v in zip(keys, vals):
                assert want.pop(bytes(kk)) == cut(v)
            seen += len(keys)
        assert seen == ms["weak_entries"] and not want


def _failing_worker(rank, world, port, tmp, mode, k):
    """as _worker, but rank 1's second batch fails on that rank alone (kmr_tune exchange_fail_once): every rank has to come back
    from the collective step with an error -- the failing one with its own, the others naming it -- instead of waiting for ever"""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kmernator_amd as ka
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        dev = torch.device("cuda", 0)
        moved = []

        def allgather(mine):
            rows = [None] * world
            dist.all_gather_object(rows, list(mine))
            return rows

        def alltoallv(send, soff, sbytes, recv, roff, rbytes, stream):
            assert hip.hipStreamSynchronize(stream) == 0
            moved.append(sum(sbytes))
            outs = []
            for r in range(world):
                buf = np.empty(sbytes[r], dtype=np.uint8)
                if sbytes[r]:
                    assert hip.hipMemcpy(buf.ctypes.data, send + soff[r], sbytes[r], 2) == 0
                outs.append(torch.from_numpy(buf))
            ins = [torch.empty(rbytes[r], dtype=torch.uint8) for r in range(world)]
            reqs = [dist.isend(outs[r], r) for r in range(world) if r != rank and sbytes[r]]
            for r in range(world):
                if r != rank and rbytes[r]:
                    dist.recv(ins[r], r)
            for q in reqs:
                q.wait()
            for r in range(world):
                if rbytes[r]:
                    assert hip.hipMemcpy(recv + roff[r], ins[r].numpy().ctypes.data, rbytes[r], 1) == 0

        lo, hi = _slice(rank, world)
        rb = _reads(False).slice(lo, hi)
        sp = ka.KmerSpectrum(ka.default_config(k, estimated_raw_kmers=N_READS * (READ_LEN - k + 1), device=0, rank=rank, world_size=world, build_mode=mode))
        sp.exchange_init_transport(allgather, alltoallv)
        outcome = []
        for i, (a, b) in enumerate(_batches(rank, world)):
            part = rb.slice(a, b)
            tb, tq, to = _dev(part, dev)
            if i == 1 and rank == 1:
                sp.tune(exchange_fail_once=1)
            before = len(moved)
            try:
                sp.exchange_add_reads(tb.data_ptr(), tq.data_ptr(), to.data_ptr(), b - a, int(part.offsets[-1]), lo + a)
                outcome.append("ok")
            except ka.KmerSpectrumError as e:
                outcome.append(str(e))
                assert len(moved) == before          # nothing was sent in the failed step
        open(os.path.join(tmp, "outcome.%d.txt" % rank), "w").write("\n".join(outcome))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", [0, 2])
def test_one_failing_rank_fails_the_step_on_every_rank(mode):
    """kmr_exchange_add_reads_dev agrees on success before every exchange: a rank-local failure between two collectives (an
    allocation, the extraction, the packing) reaches all ranks through the status word of the gathered rows."""
    world, k = 3, 31
    port = 35100 + (os.getpid() % 1500) + mode
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_failing_worker, args=(world, port, tmp, mode, k), nprocs=world, join=True)
        for r in range(world):
            first, second = open(os.path.join(tmp, "outcome.%d.txt" % r)).read().split("\n")
            assert first == "ok"
            if r == 1:
                assert "injected failure" in second
            else:
                assert "rank 1 failed" in second
