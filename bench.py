#!/usr/bin/env python3
"""bench.py -- k-mer spectrum build throughput on MI355X.

A step = one full pass of the hot path over one batch of synthetic reads that is already
resident in HBM: reset (empty table) -> extract + canonicalise + weight + lookup3 + insert
-> finalize (purge, bucket, sort: the queryable spectrum in the reference's map layout).

N=1 runs BASELINE.json configs[1] ("C2": k=31, 10M synthetic 150 bp reads, 1 GPU, single hash
partition).  N>1 (launched by torch.distributed.run, one rank per GPU) is weak scaling: every
rank brings its own 10M reads of one shared genome and k-mers are exchanged to their
lookup3 owner with an RCCL all-to-all.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

K = 31
READ_LEN = 150
ERR = 0.01
HBM_PEAK = 8.0e12      # B/s, /opt/skills/guides/MI355X_MICROARCH.md


def gen_reads(n_reads, genome_len, seed, rank, dev, chunk=1 << 20):
    """SURVEY.md 8(d) generator on the GPU: uniform genome, uniform starts, random strand,
    1 % substitutions, flat Q40 ('I'), no N.  The genome depends on `seed` only, the reads on
    (seed, rank)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    genome = torch.randint(0, 4, (genome_len,), generator=g, device=dev, dtype=torch.uint8)
    g.manual_seed(seed * 1000003 + 17 * (rank + 1))
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    bases = torch.empty(n_reads * READ_LEN, dtype=torch.uint8, device=dev)
    ar = torch.arange(READ_LEN, device=dev, dtype=torch.int64)
    for lo in range(0, n_reads, chunk):
        m = min(chunk, n_reads - lo)
        starts = torch.randint(0, genome_len - READ_LEN + 1, (m,), generator=g, device=dev, dtype=torch.int64)
        codes = genome[starts[:, None] + ar[None, :]]
        strand = torch.randint(0, 2, (m,), generator=g, device=dev, dtype=torch.uint8).bool()
        rc = (3 - codes).flip(1)
        codes = torch.where(strand[:, None], rc, codes)
        errs = torch.rand((m, READ_LEN), generator=g, device=dev) < ERR
        shift = torch.randint(1, 4, (m, READ_LEN), generator=g, device=dev, dtype=torch.uint8)
        codes = torch.where(errs, (codes + shift) & 3, codes)
        bases[lo * READ_LEN:(lo + m) * READ_LEN] = lut[codes.long()].reshape(-1)
    quals = torch.full((n_reads * READ_LEN,), ord("I"), dtype=torch.uint8, device=dev)
    offsets = torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * READ_LEN
    return bases, quals, offsets


def cpu_baseline(bases, quals, n_sample):
    """The oracle (CPU restatement of _buildKmerSpectrumParallel, kind 'port') on a bounded
    sample of the same workload, all host cores."""
    from helpers import OracleSpectrum, ReadBatch, default_config, oracle_lib
    lib = oracle_lib()
    cores = lib.orc_max_threads()
    b = bases[:n_sample * READ_LEN].cpu().numpy()
    q = quals[:n_sample * READ_LEN].cpu().numpy()
    off = (np.arange(n_sample + 1, dtype=np.uint64) * np.uint64(READ_LEN))
    rb = ReadBatch.from_arrays(b, q, off)
    cfg = default_config(K, estimated_raw_kmers=n_sample * (READ_LEN - K + 1))
    # the reference's build takes global atomics per occurrence (src/KmerSpectrum.h:1589-1603) and does
    # not scale to many cores, so the thread count is swept and the fastest one is reported
    best, best_t, sweep = None, 1, {}
    for t in sorted(set([1, 4, 8, 16, 32, cores])):
        if t > cores:
            continue
        s = OracleSpectrum(cfg)
        t0 = time.perf_counter()
        s.add_reads(rb, threads=t)
        s.finalize(2)
        dt = time.perf_counter() - t0
        st = s.stats()
        s.close()
        sweep[t] = st["raw_kmers"] / dt
        if best is None or dt < best:
            best, best_t = dt, t
    return {"value": st["raw_kmers"] / best, "unit": "kmers/s", "cores": best_t, "kind": "port",
            "sample": "first %d reads of the same synthetic batch (%d k-mers), OpenMP T x T bucket-ownership build + purge; "
                      "fastest of a thread sweep on a %d-thread host" % (n_sample, st["raw_kmers"], cores),
            "seconds": best, "kmers_per_s_by_threads": sweep}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU (C2 = 10M)")
    ap.add_argument("--cpu-sample", type=int, default=50_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--force-exchange", action="store_true", help="run the owner-exchange path even on one GPU (sanity/timing of the N>1 code)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help="development: all ranks of a torch.distributed.run launch share GPU 0 and talk "
                    "over gloo (RCCL refuses two ranks on one device); exercises the N>1 code, its timings mean nothing")
    ap.add_argument("--tune", action="append", default=[], metavar="KNOB=VALUE", help="kmr_tune knob of the handle (measurement sweeps), e.g. partition_blocks=192")
    ap.add_argument("--no-check", action="store_true", help="development: skip the conservation assert (ablation runs of a debug build)")
    ap.add_argument("--build-mode", type=int, default=0, help="kmr_config.build_mode: 0 auto (streaming partition), 1 device table")
    args = ap.parse_args()
    # stdout carries exactly one JSON line: libraries that print there (RCCL writes a version banner when its first communicator
    # comes up) are pointed at stderr, and the line goes out through the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rehearse_on_one_gpu:
        local_rank = 0
    if world > 1 or args.force_exchange:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        elif world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    import kmernator_amd as ka
    from kmernator_amd.distributed import build_partitioned

    n_reads = args.reads
    kmers_per_read = READ_LEN - K + 1
    genome_len = 5 * n_reads * world               # 30x coverage
    bases, quals, offsets = gen_reads(n_reads, genome_len, 1, rank, dev)
    total_bases = n_reads * READ_LEN
    torch.cuda.synchronize()

    cfg = ka.default_config(K, estimated_raw_kmers=n_reads * kmers_per_read * world, device=dev.index,
                            rank=rank, world_size=world, build_mode=args.build_mode)
    sp = ka.KmerSpectrum(cfg)
    sp.tune(**{kv.split("=")[0]: float(kv.split("=")[1]) for kv in args.tune})

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        sp.reset()
        if world == 1 and not args.force_exchange:
            sp.buildKmerSpectrumDevice(bases.data_ptr(), quals.data_ptr(), offsets.data_ptr(), n_reads, total_bases, 0)
        else:
            build_partitioned(sp, bases, quals, offsets, first_read_idx=rank * n_reads)
        sp.finalize(2)

    for _ in range(args.warmup):
        step()
    barrier()
    sp.kernel_time_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    red_dev = "cpu" if args.rehearse_on_one_gpu else dev      # gloo reduces host tensors
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    st = sp.stats()
    build_ms, build_launches = sp.kernel_time(0)
    fin_ms, fin_launches = sp.kernel_time(1)
    raw_local = st["raw_kmers"]
    uniq_local = st["unique_kmers"]
    if dist is not None:
        t = torch.tensor([raw_local, uniq_local], dtype=torch.int64, device=red_dev)
        dist.all_reduce(t)
        raw_total, uniq_total = int(t[0].item()), int(t[1].item())
    else:
        raw_total, uniq_total = raw_local, uniq_local
    total_kmers = n_reads * kmers_per_read * world
    assert args.no_check or raw_total == total_kmers, (raw_total, total_kmers)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_kmers / (dt / args.steps)
        kb = (K + 3) // 4
        # algorithmic bytes: K * (2L/(L-k+1) + kb + 24) + D * kb for the k-mers rank 0 handled.
        # The hot path of one step is a short chain of launches (extract, partition x2, count, bucket/sort),
        # all timed with HIP events on the handle's stream; achieved = algorithmic bytes of the step / their sum.
        k_local = raw_local
        alg_bytes = k_local * (2.0 * READ_LEN / kmers_per_read + kb + 24) + uniq_local * kb
        hot_ms = (build_ms + fin_ms) / max(1, args.steps)
        achieved = alg_bytes / (hot_ms / 1e3) if hot_ms > 0 else 0.0
        mode = "device-table" if args.build_mode == 1 else "streaming-partition"
        out = {
            "metric": "total k-mers/sec at k=31, 150 bp reads (spectrum build, inputs resident in HBM)",
            "value": value, "unit": "kmers/s",
            "distinct_kmers_per_sec": uniq_total / (dt / args.steps),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "C2: k=31, %d synthetic 150 bp reads per GPU, genome %d bp (30x), 1%% substitutions, flat Q40, "
                                   "min-depth 2, %s" % (n_reads, genome_len, "single hash partition" if world == 1 else
                                                        "owner-partitioned (lookup3) RCCL all-to-all over %d GPUs" % world),
                       "k": K, "read_len": READ_LEN, "reads_per_gpu": n_reads, "total_kmers": total_kmers,
                       "distinct_kmers": uniq_total, "build_mode": mode,
                       "parallelism": "1 process per GPU, owner partition x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": None,
                         "kernel": ("extract_kernel<1,false,InsertOp>" if args.build_mode == 1 else
                                    "hot path of one step: extract_kernel<LinearOp> + partition_direct_kernel<1,1> per sub-batch, "
                                    "then partition_direct_kernel<1,2> + count_kernel + entry_scatter/sort_buckets"),
                         "algorithmic_bytes_per_step": alg_bytes, "hot_path_ms_per_step": hot_ms,
                         "build_ms_per_step": build_ms / max(1, args.steps), "build_launch_groups_per_step": build_launches // max(1, args.steps),
                         "finalize_ms_per_step": fin_ms / max(1, args.steps)},
        }
        if args.build_mode != 1:
            # per-kernel view of the same step: every kernel against the bytes its own role has to move (16-byte
            # records at k=31; weak entries are 8-byte key + 12-byte value), each timed with HIP events on the handle's
            # stream; rocprofv3 averages of the same command are in profiles/ (kernel_stats csv)
            rec = 8 * ((kb + 7) // 8) + 8
            weak = st["weak_entries"]
            per = [("extract_kernel<LinearOp>", 2, k_local * (2.0 * READ_LEN / kmers_per_read + rec)),
                   ("partition_direct_kernel<level 1>", 3, k_local * 2.0 * rec),
                   ("partition_direct_kernel<level 2>", 4, k_local * 2.0 * rec),
                   ("count_kernel", 5, k_local * rec + weak * 20.0),
                   ("bucket scan + entry_scatter_kernel + sort_buckets_kernel", 6, weak * 60.0)]
            kernels = []
            for name, grp, nbytes in per:
                ms, launches = sp.kernel_time(grp)
                if launches == 0:
                    continue
                ms_step = ms / max(1, args.steps)
                kernels.append({"name": name, "ms_per_step": ms_step, "launches_per_step": launches // max(1, args.steps),
                                "ms_per_launch": ms / launches, "bytes_per_step": nbytes,
                                "achieved_GBps": nbytes / (ms_step / 1e3) / 1e9, "frac": nbytes / (ms_step / 1e3) / HBM_PEAK})
            out["roofline"]["kernels"] = kernels
            if kernels:
                out["roofline"]["dominant_kernel"] = max(kernels, key=lambda k: k["ms_per_step"])["name"]
        # HBM bytes from the PMC counters: bench.py cannot run rocprofv3 on itself, so the committed summary of
        # the same command (profiles/r01_pmc_traffic.json, tools/pmc.sh) is quoted when it describes this workload
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if args.build_mode != 1 and n_reads == 10_000_000 and world == 1:
                out["roofline"]["traffic"] = tj["hot_path_total_GB_per_step"] * 1e9
                out["roofline"]["traffic_source"] = "profiles/r01_pmc_traffic.json (bytes per step, FETCH_SIZE x2 corrected)"
        except Exception:
            pass
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(bases, quals, min(args.cpu_sample, n_reads))
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
