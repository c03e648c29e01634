"""Rehearsal of the N>1 code path with several ranks on ONE GPU: one process per rank as under torch.distributed.run, gloo as
the transport (RCCL refuses two ranks on one device; kmernator_amd.distributed stages device tensors through the host for
gloo), everything else -- extract-by-owner, the chunked and pipelined exchange driver, wire-format inserts, finalize, the
request / response scoring -- is the code bench.py --gpus N runs.  The union of the ranks' spectra must be the single-GPU
spectrum, every k-mer on its lookup3 owner, and every rank's read scores those of the whole spectrum."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import KMR_MAP_WEAK, parse_image, synth_reads

pytestmark = pytest.mark.gpu

N_READS, READ_LEN, K = 90000, 150, 31


def _reads():
    return synth_reads(N_READS, read_len=READ_LEN, genome_len=6 * N_READS, seed=12, quality="noisy", n_rate=0.001)


def _slice(rank, world):
    per = (N_READS + world - 1) // world
    lo = rank * per - (5000 * rank if rank else 0)            # uneven shares: the ranks run different numbers of chunks
    hi = N_READS if rank == world - 1 else (rank + 1) * per - 5000 * (rank + 1)
    return lo, hi


def _worker(rank, world, port, tmp, pipeline):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kmernator_amd as ka
        from kmernator_amd.distributed import build_partitioned, score_partitioned
        dev = torch.device("cuda", 0)
        lo, hi = _slice(rank, world)
        rb = _reads().slice(lo, hi)
        tb = torch.from_numpy(np.concatenate([rb.bases, np.zeros(64, np.uint8)])).to(dev)
        tq = torch.from_numpy(np.concatenate([rb.quals, np.zeros(64, np.uint8)])).to(dev)
        to = torch.from_numpy(rb.offsets.astype(np.int64)).to(dev)
        sp = ka.KmerSpectrum(ka.default_config(K, estimated_raw_kmers=N_READS * (READ_LEN - K + 1), device=0, rank=rank, world_size=world))
        build_partitioned(sp, tb, tq, to, first_read_idx=lo, chunk_reads=7000 + 2000 * rank, pipeline=pipeline)
        sp.finalize(2)
        np.save(os.path.join(tmp, "image.%d.npy" % rank), sp.image(KMR_MAP_WEAK))
        st = sp.stats()
        np.save(os.path.join(tmp, "stats.%d.npy" % rank), np.array([st["raw_kmers"], st["raw_good_kmers"], st["weak_entries"], st["unique_kmers"]], dtype=np.int64))
        res = score_partitioned(sp, tb, to, 2, "MEDIAN", chunk_reads=9000 + 3000 * rank)
        np.savez(os.path.join(tmp, "score.%d.npz" % rank), to=res[0], tl=res[1], sc=res[2], wt=res[3])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,pipeline", [(2, True), (3, False)])
def test_ranks_sharing_one_gpu(world, pipeline):
    import kmernator_amd as ka
    port = 30500 + (os.getpid() % 1500) + world
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(world, port, tmp, pipeline), nprocs=world, join=True)
        rb = _reads()
        whole = ka.KmerSpectrum(ka.default_config(K, estimated_raw_kmers=N_READS * (READ_LEN - K + 1), device=0))
        whole.buildKmerSpectrum(rb.bases, rb.quals, rb.offsets)
        whole.finalize(2)
        ws = whole.stats()
        stats = sum(np.load(os.path.join(tmp, "stats.%d.npy" % r)) for r in range(world))
        # the reads overlap between neighbouring slices (uneven shares), so compare against a spectrum of the same multiset
        both = []
        for r in range(world):
            lo, hi = _slice(r, world)
            both.append(rb.slice(lo, hi))
        multi = ka.KmerSpectrum(ka.default_config(K, estimated_raw_kmers=N_READS * (READ_LEN - K + 1), device=0))
        for part in both:
            multi.buildKmerSpectrum(part.bases, part.quals, part.offsets)
        multi.finalize(2)
        ms = multi.stats()
        assert (int(stats[1]), int(stats[2]), int(stats[3])) == (ms["raw_good_kmers"], ms["weak_entries"], ms["unique_kmers"]), (stats, ms, ws)
        lib = ka.load()
        total = 0
        for r in range(world):
            nb, mask, buckets = parse_image(np.load(os.path.join(tmp, "image.%d.npy" % r)), multi.kb, 12)
            keys = np.concatenate([k for k, _ in buckets if len(k)])
            vals = np.concatenate([v for _, v in buckets if len(v)])
            counts = np.ascontiguousarray(vals[:, :2]).view(np.uint16).reshape(-1).astype(np.uint32)
            assert np.array_equal(multi.getCount(keys), counts)
            for kk in keys[::997]:
                assert lib.kmr_distributed_thread_id(lib.kmr_hash(kk.tobytes(), multi.kb), world) == r
            total += len(keys)
        assert total == ms["weak_entries"]
        for r in range(world):
            part = both[r]
            want = multi.scoreAndTrimReads(part.bases, part.offsets, 2, "MEDIAN")
            got = np.load(os.path.join(tmp, "score.%d.npz" % r))
            for a, b in zip((got["to"], got["tl"], got["sc"], got["wt"]), want):
                assert np.array_equal(a, b)


def _worker_sk(rank, world, port, tmp, k, coarse=0, ext=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kmernator_amd as ka
        from kmernator_amd.distributed import build_partitioned_superkmers, score_partitioned
        dev = torch.device("cuda", 0)
        lo, hi = _slice(rank, world)
        rb = _reads().slice(lo, hi)
        tb = torch.from_numpy(np.concatenate([rb.bases, np.zeros(64, np.uint8)])).to(dev)
        tq = torch.from_numpy(np.concatenate([rb.quals, np.zeros(64, np.uint8)])).to(dev)
        to = torch.from_numpy(rb.offsets.astype(np.int64)).to(dev)
        xkw = dict(value_kind=1, min_weight=0.0, min_quality_score=2) if ext else {}
        sp = ka.KmerSpectrum(ka.default_config(k, estimated_raw_kmers=N_READS * (READ_LEN - k + 1), device=0, rank=rank, world_size=world, build_mode=3, **xkw))
        sp.tune(coarse_lists=coarse)
        xs = {}
        build_partitioned_superkmers(sp, tb, tq, to, first_read_idx=lo, stats=xs, pieces=1 if world == 2 else 3)      # three ranks: in three pieces
        sp.finalize(2)
        np.save(os.path.join(tmp, "image.%d.npy" % rank), sp.image(KMR_MAP_WEAK))
        st = sp.stats()
        np.save(os.path.join(tmp, "stats.%d.npy" % rank), np.array([st["raw_kmers"], st["raw_good_kmers"], st["weak_entries"], st["unique_kmers"], xs.get("bytes_to_peers", 0)], dtype=np.int64))
        res = score_partitioned(sp, tb, to, 2, "MEDIAN", chunk_reads=9000 + 3000 * rank)      # requests go to the owner of the k-mer's list
        np.savez(os.path.join(tmp, "score.%d.npz" % rank), to=res[0], tl=res[1], sc=res[2], wt=res[3])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k,coarse,ext", [(2, 31, 0, 0), (3, 51, 0, 0), (3, 31, 1, 0), (2, 51, 1, 0), (2, 21, 0, 1), (3, 31, 0, 1)])
def test_superkmer_exchange_ranks_sharing_one_gpu(world, k, coarse, ext):
    """The N > 1 build of build_mode 3 (every rank scatters its reads' super-k-mers into the job's lists, the chunks of other
    owners travel, the owner appends them to its lists) with 2 and 3 ranks on this GPU over gloo: the union of the ranks' weak
    maps is the weak map of one spectrum over the same reads -- same keys, counts, direction biases -- every k-mer lives on
    exactly one rank, the statistics add up, and every rank's reads score as on the whole spectrum (a lookup goes to the owner of
    the k-mer's list, not to its lookup3 owner).  coarse = 1: the ranks scatter into, exchange and adopt coarse lists (2^ceil(log2 world)
    fine lists each) and the owner splits them before the count pass (kmr_tune "coarse_lists").  ext = 1: extension values -- the records
    that travel carry their neighbour bases and qualities, the twelve tallies of every k-mer must be the whole spectrum's too."""
    import kmernator_amd as ka
    port = 32100 + (os.getpid() % 1500) + world
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker_sk, args=(world, port + 17 * coarse + 31 * ext, tmp, k, coarse, ext), nprocs=world, join=True)
        rb = _reads()
        xkw = dict(value_kind=1, min_weight=0.0, min_quality_score=2) if ext else {}
        vbytes = 60 if ext else 12
        multi = ka.KmerSpectrum(ka.default_config(k, estimated_raw_kmers=N_READS * (READ_LEN - k + 1), device=0, **xkw))
        for r in range(world):
            lo, hi = _slice(r, world)
            part = rb.slice(lo, hi)
            multi.buildKmerSpectrum(part.bases, part.quals, part.offsets)
        multi.finalize(2)
        ms = multi.stats()
        stats = sum(np.load(os.path.join(tmp, "stats.%d.npy" % r)) for r in range(world))
        assert (int(stats[0]), int(stats[1]), int(stats[2]), int(stats[3])) == (ms["raw_kmers"], ms["raw_good_kmers"], ms["weak_entries"], ms["unique_kmers"]), (stats, ms)
        assert int(stats[4]) > 0
        _, _, whole = parse_image(multi.image(KMR_MAP_WEAK), multi.kb, vbytes)
        wk = np.concatenate([kk for kk, _ in whole if len(kk)])
        wv = np.concatenate([v for _, v in whole if len(v)])
        want = {bytes(kk): bytes(v[:2]) + bytes(v[8:]) for kk, v in zip(wk, wv)}       # count, directionBias (the first sighting through an exchange is still the first in the stream: ordinals travel) and the extension tallies
        seen = 0
        for r in range(world):
            _, _, buckets = parse_image(np.load(os.path.join(tmp, "image.%d.npy" % r)), multi.kb, vbytes)
            keys = np.concatenate([kk for kk, _ in buckets if len(kk)])
            vals = np.concatenate([v for _, v in buckets if len(v)])
            for kk, v in zip(keys, vals):
                assert want.pop(bytes(kk)) == bytes(v[:2]) + bytes(v[8:])
            seen += len(keys)
        assert seen == ms["weak_entries"] and not want
        for r in range(world):
            lo, hi = _slice(r, world)
            part = rb.slice(lo, hi)
            wanted = multi.scoreAndTrimReads(part.bases, part.offsets, 2, "MEDIAN")
            got = np.load(os.path.join(tmp, "score.%d.npz" % r))
            for a, b in zip((got["to"], got["tl"], got["sc"], got["wt"]), wanted):
                assert np.array_equal(a, b)


@pytest.mark.parametrize("how", ["external launcher", "bench.py --gpus 2"])
def test_bench_n2_code_path_on_one_gpu(how):
    """bench.py for N = 2, both ranks on this GPU over gloo (RCCL refuses two ranks on one device): once under
    torch.distributed.run as the driver launches it, once as plain `bench.py --gpus 2`, which has to start its two ranks itself.
    One JSON line must come out, say n_gpus 2, and account for every k-mer of both ranks; its roofline is priced on the whole
    step (exchange included) and the exchange's bytes and time are reported."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 31800 + (os.getpid() % 1000)
    tail = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--reads", "300000", "--no-cpu", "--rehearse-on-one-gpu"]
    if how == "external launcher":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)] + tail
    else:
        cmd = [sys.executable] + tail
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["total_kmers"] == 2 * 300000 * 120
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
    assert abs(d["roofline"]["achieved"] * 1e9 - d["roofline"]["algorithmic_bytes_per_step"] / (d["ms_per_step"] / 1e3)) < 1e-3 * d["roofline"]["achieved"] * 1e9
    assert d["exchange"]["bytes_to_peers"] > 0 and d["exchange"]["alltoall_ms"] > 0
