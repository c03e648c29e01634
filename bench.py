#!/usr/bin/env python3
"""bench.py -- k-mer spectrum build throughput on MI355X.

A step = one full pass of the hot path over one batch of synthetic reads that is already
resident in HBM: reset (empty maps) -> extract + canonicalise + weight + count -> finalize
(purge, bucket, sort: the queryable spectrum in the reference's map layout).

N=1 runs BASELINE.json configs[1] ("C2": k=31, 10M synthetic 150 bp reads, 1 GPU, single hash
partition).  N>1 is weak scaling, one rank per GPU: every rank brings its own 10M reads of one
shared genome and the k-mers travel to their owner over RCCL.  `python bench.py --gpus N` starts
the N ranks itself (python -m torch.distributed.run as a child process, before this process has
touched a GPU); under an external launcher (WORLD_SIZE set) it is one of the ranks.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

K = 31
READ_LEN = 150
ERR = 0.01
HBM_PEAK = 8.0e12      # B/s, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: C2 = 10M; with --gpus 8 C3 = 12.5M per GPU)")
    ap.add_argument("--quality", choices=["flat", "noisy"], default="flat", help="SURVEY 8(d): flat Q40 (the headline), or noisy "
                    "(Q in {40,30,20,10,2} with p = {.80,.10,.05,.04,.01}, errors at Q10: the divide chain and the discard path are live)")
    ap.add_argument("--cpu-reads", type=int, default=1_000_000, help="reads of the CPU baseline's sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the PCIe-inclusive leg")
    ap.add_argument("--h2d-pieces", type=int, default=4, help="pieces the packed PCIe-inclusive leg sends a batch in")
    ap.add_argument("--force-exchange", action="store_true", help="run the owner-exchange path even on one GPU (sanity/timing of the N>1 code)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help="development: all ranks share GPU 0 and talk over gloo (RCCL refuses two "
                    "ranks on one device); exercises the N>1 code, its timings mean nothing")
    ap.add_argument("--exchange-pieces", type=int, default=1, help="N > 1: the batch goes through the list exchange in this many pieces, the all-to-all of one "
                    "running while the next is extracted.  Default 1: with 10 M reads per rank a rank holds half a chunk for each of the job's "
                    "lists at 8 ranks, so every further piece sends (and the owner adopts) that many more partly filled chunks "
                    "(tools/two_rank_step.py: 43.8 ms of compute per step in one piece, 48.4 in two)")
    ap.add_argument("--exchange-steps", type=int, default=1, help="N > 1: the exchange goes over the list space in this many steps and every owner counts the lists that "
                    "have arrived (kmr_count_lists_prefix) while the next step is on the wire.  Default 1: nothing on a one-GPU box can say what it buys")
    ap.add_argument("--tune", action="append", default=[], metavar="KNOB=VALUE", help="kmr_tune knob of the handle (measurement sweeps), e.g. partition_blocks=192")
    ap.add_argument("--no-check", action="store_true", help="development: skip the conservation assert (ablation runs of a debug build)")
    ap.add_argument("--build-mode", type=int, default=0, help="kmr_config.build_mode: 0 auto, 1 device table, 2 two-level k-mer partition, 3 super-k-mer lists")
    return ap.parse_args()


def launch_ranks(args):
    """`bench.py --gpus N` without a launcher: start N ranks as a child job and pass its JSON line through.  Nothing in this
    process has touched a GPU (torch is not even imported yet)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    sys.exit(p.returncode if p.returncode else (0 if lines else 1))


def gen_reads(torch, n_reads, genome_len, seed, rank, dev, quality="flat", read_len=READ_LEN):
    """SURVEY.md 8(d)'s generator on the GPU (kmr_synth_reads_dev: xorshift64*, integer only -- the oracle's orc_synth_reads gives the
    same bytes on a CPU, which is how tests/golden/full_size_digests.json pins C2 and C4 to the oracle): uniform genome, uniform
    starts, random strand, 1 % substitutions, no N.  The genome depends on `seed` only; rank r holds reads r*n_reads .. (r+1)*n_reads
    of the job."""
    import kmernator_amd as ka
    return ka.synth_reads_device(torch, seed, rank * n_reads, n_reads, read_len, genome_len, quality == "noisy", dev)


def cpu_baseline(torch, n_reads, quality, dev):
    """The oracle (CPU restatement of _buildKmerSpectrumParallel, kind 'port') on a sample at the SAME coverage as the GPU
    workload (its own genome of 5 bases per read: 30x): the thread count is picked on a 100 000-read sweep (the reference's
    build takes global atomics per occurrence, src/KmerSpectrum.h:1589-1603, and does not scale to many cores), then the full
    sample is built three times at that count and the median reported."""
    import numpy as np
    from helpers import OracleSpectrum, ReadBatch, default_config, oracle_lib
    lib = oracle_lib()
    cores = lib.orc_max_threads()
    from helpers import synth_reads_8d
    rb_all = synth_reads_8d(7, 0, n_reads, READ_LEN, 5 * n_reads, quality == "noisy", threads=cores)
    b, q = rb_all.bases, rb_all.quals

    def run(n, threads):
        off = np.arange(n + 1, dtype=np.uint64) * np.uint64(READ_LEN)
        rb = ReadBatch.from_arrays(b[:n * READ_LEN], q[:n * READ_LEN], off)
        s = OracleSpectrum(default_config(K, estimated_raw_kmers=n * (READ_LEN - K + 1)))
        t0 = time.perf_counter()
        s.add_reads(rb, threads=threads)
        s.finalize(2)
        dt = time.perf_counter() - t0
        raw = s.stats()["raw_kmers"]
        s.close()
        return raw, dt

    sweep = {}
    n_sweep = min(100_000, n_reads)
    for t in sorted(set([1, 4, 8, 16, 32, cores])):
        if t <= cores:
            raw, dt = run(n_sweep, t)
            sweep[t] = raw / dt
    best_t = max(sweep, key=sweep.get)
    runs = [run(n_reads, best_t) for _ in range(3)]
    raw = runs[0][0]
    secs = statistics.median(r[1] for r in runs)
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": raw / secs, "unit": "kmers/s", "cores": best_t, "kind": "port",
            "sample": "%d synthetic reads (%d k-mers) of a %d bp genome -- the GPU workload's 30x coverage, error rate and quality mode -- "
                      "through the OpenMP T x T bucket-ownership build + purge of the oracle; %d threads (best of a sweep on %d reads), "
                      "median of 3 runs" % (n_reads, raw, 5 * n_reads, best_t, n_sweep),
            "seconds": secs, "run_seconds": [r[1] for r in runs], "kmers_per_s_by_threads_on_sweep": sweep,
            "host": {"nproc": os.cpu_count(), "threads_available": cores, "cpu_model": model}}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
    # stdout carries exactly one JSON line: libraries that print there (RCCL writes a version banner when its first communicator
    # comes up) are pointed at stderr, and the line goes out through the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rehearse_on_one_gpu:
        local_rank = 0
    exchange = world > 1 or args.force_exchange
    if exchange:
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
    from kmernator_amd.distributed import build_partitioned, build_partitioned_superkmers

    # BASELINE.json configs: 1 GPU = C2 (10 M reads, seed 1, 50 Mbp); 8 GPUs = C3 exactly (100 M reads = 12.5 M per GPU, seed 2, 500 Mbp);
    # other rank counts keep C2's per-GPU batch (weak scaling: 10 M reads per GPU of one shared genome at 30x)
    c3 = world == 8 and args.reads in (0, 12_500_000)
    n_reads = args.reads or (12_500_000 if c3 else 10_000_000)
    seed = 2 if c3 else 1
    kmers_per_read = READ_LEN - K + 1
    genome_len = 500_000_000 if c3 else 5 * n_reads * world               # 30x coverage
    bases, quals, offsets = gen_reads(torch, n_reads, genome_len, seed, rank, dev, args.quality)
    total_bases = n_reads * READ_LEN
    torch.cuda.synchronize()

    # N > 1 (or --force-exchange): build_mode 0 means the super-k-mer lists and their chunk exchange; 2 the k-mer record exchange
    mode_num = args.build_mode or 3
    cfg = ka.default_config(K, estimated_raw_kmers=n_reads * kmers_per_read * world, device=dev.index,
                            rank=rank, world_size=world, build_mode=(mode_num if exchange else args.build_mode))
    sp = ka.KmerSpectrum(cfg)
    sp.tune(**{kv.split("=")[0]: float(kv.split("=")[1]) for kv in args.tune})
    xstats = {}

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        sp.reset()
        if not exchange:
            sp.buildKmerSpectrumDevice(bases.data_ptr(), quals.data_ptr(), offsets.data_ptr(), n_reads, total_bases, 0)
        elif mode_num == 3:
            build_partitioned_superkmers(sp, bases[:total_bases + 64], quals[:total_bases + 64], offsets, first_read_idx=rank * n_reads, stats=xstats,
                                         stream_origin=rank * total_bases, pieces=args.exchange_pieces if world > 1 else 1,
                                         list_steps=args.exchange_steps, early_min_depth=2 if args.exchange_steps > 1 else None)
        else:
            build_partitioned(sp, bases, quals, offsets, first_read_idx=rank * n_reads, stats=xstats)
        sp.finalize(2)

    for _ in range(args.warmup):
        step()
    barrier()
    sp.kernel_time_reset()
    xstats.clear()
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
    ktimes = {g: sp.kernel_time(g) for g in range(7)}       # HIP-event times of the timed steps (the PCIe leg below runs more builds)
    raw_local, uniq_local = st["raw_kmers"], st["unique_kmers"]
    if dist is not None:
        t = torch.tensor([raw_local, uniq_local], dtype=torch.int64, device=red_dev)
        dist.all_reduce(t)
        raw_total, uniq_total = int(t[0].item()), int(t[1].item())
    else:
        raw_total, uniq_total = raw_local, uniq_local
    total_kmers = n_reads * kmers_per_read * world
    # conservation: every k-mer of every rank's reads is accounted for somewhere (through the exchange the owners count the good
    # ones they receive, so the flat-quality case -- nothing discarded -- is the one that can be checked there)
    assert args.no_check or raw_total == total_kmers or (exchange and mode_num != 3 and args.quality != "flat"), (raw_total, total_kmers)

    # The line checks itself against the ORACLE (outside the timed region): the ranks' statistics and map digests (kmr_map_digest: additive over
    # owners) must add up to what the serial CPU oracle made of the same job (tests/golden/full_size_digests.json, written by
    # tests/golden/make_full_size_digests.py in the build container) -- C2 flat / noisy on one GPU, the weak-scaling jobs at N = 2 / 4, and
    # BASELINE config 3 at N = 8.  Configurations without a committed digest report null.
    oracle_check = None
    try:
        from helpers import add_digests, digests_agree, full_size_golden
        gname = None
        if args.build_mode in (0, 3) and not args.tune:
            if world == 1 and n_reads == 10_000_000:
                gname = "c2_flat" if args.quality == "flat" else "c2_noisy"
            elif c3 and args.quality == "flat":
                gname = "c3_flat"
            elif world in (2, 4) and n_reads == 10_000_000 and args.quality == "flat":
                gname = "scale_n%d" % world
        try:
            gold = full_size_golden(gname) if gname else None
        except KeyError:
            gold = None
        if gold is not None:
            dg = sp.digest(0)
            keys = ("raw_kmers", "raw_good_kmers", "unique_kmers", "singleton_kmers", "discarded", "weak_entries", "reads")
            ints = [st[k_] for k_ in keys] + [dg[k_] for k_ in ("entries", "count_sum", "dir_sum", "hash_sum", "hash_xor")]
            if dist is not None:
                t = torch.tensor([v - (1 << 64) if v >= (1 << 63) else v for v in ints], dtype=torch.int64, device=red_dev)
                every = [torch.zeros_like(t) for _ in range(world)]
                dist.all_gather(every, t)
                w = torch.tensor([dg["weighted_sum"]], dtype=torch.float64, device=red_dev)
                dist.all_reduce(w)
                rows = [[int(x) & 0xFFFFFFFFFFFFFFFF for x in e.cpu().tolist()] for e in every]
                wsum = float(w.item())
            else:
                rows, wsum = [ints], dg["weighted_sum"]
            stats_sum = {k_: sum(r[i] for r in rows) for i, k_ in enumerate(keys)}
            dsum = None
            for r in rows:
                dsum = add_digests(dsum, dict(entries=r[7], count_sum=r[8], dir_sum=r[9], hash_sum=r[10], hash_xor=r[11], weighted_sum=0.0))
            dsum["weighted_sum"] = wsum
            oracle_check = {"config": gname, "statistics_equal": all(stats_sum[k_] == gold["stats"][k_] for k_ in keys),
                            "weak_map_digest_equal": bool(digests_agree(dsum, gold["weak_digest"], 1e-6)),
                            "what": "sum over ranks of kmr_get_stats and kmr_map_digest against the serial CPU oracle's (keys, counts, direction biases bit for bit; weightedCount sum within 1e-6)"}
    except Exception as e:      # the check must never cost a bench line
        oracle_check = {"error": repr(e)}

    # PCIe-inclusive legs (t_build from "first byte of in-memory reads available"): the reads start in pinned host memory and
    # go to the device in eight pieces on a copy stream while the library's stream builds the pieces that have arrived.  Twice: as
    # text (kmr_add_reads_dev: a byte per base and per quality), and as the reference's Read keeps them (kmr_add_reads_twobit_dev:
    # bases 2-bit packed, every read on bytes of its own; one quality character for all bases where the input has just one)
    h2d = None
    if not args.no_h2d and not exchange and rank == 0:
        lib_stream = torch.cuda.ExternalStream(sp.stream(), device=dev)
        copy_stream = torch.cuda.Stream(device=dev)
        uniform_q = args.quality == "flat"
        PB = (READ_LEN + 3) // 4                                      # packed bytes per read

        def leg(packed, pieces):
            per = (n_reads + pieces - 1) // pieces
            hq = dq = None
            if packed:
                hb = torch.empty(n_reads * PB, dtype=torch.uint8).pin_memory()
                for lo in range(0, n_reads, 1 << 20):                    # TwoBitSequence::compressSequence of every read (A C G T -> 0 1 2 3, first base in bits 7-6)
                    m = min(1 << 20, n_reads - lo)
                    c = bases[lo * READ_LEN:(lo + m) * READ_LEN].view(m, READ_LEN)
                    c = ((c >> 1) & 3) ^ ((c >> 2) & 1)
                    c = torch.nn.functional.pad(c, (0, PB * 4 - READ_LEN)).view(m, PB, 4)
                    hb[lo * PB:(lo + m) * PB].copy_((c[:, :, 0] << 6 | c[:, :, 1] << 4 | c[:, :, 2] << 2 | c[:, :, 3]).reshape(-1))
                db = torch.empty(n_reads * PB + 64, dtype=torch.uint8, device=dev)
                tb_off = torch.arange(n_reads + 1, device=dev, dtype=torch.int64) * PB
                unit = PB
            else:
                hb = torch.empty(total_bases, dtype=torch.uint8).pin_memory()
                hb.copy_(bases[:total_bases])
                db = torch.empty_like(bases)
                db[total_bases:] = 0
                unit = READ_LEN
            if not (packed and uniform_q):
                hq = torch.empty(total_bases, dtype=torch.uint8).pin_memory()
                hq.copy_(quals[:total_bases])
                dq = torch.empty_like(quals)
                dq[total_bases:] = 0
            nbytes = hb.numel() + (hq.numel() if hq is not None else 0)

            def step(copy_only=False):
                sp.reset()
                torch.cuda.synchronize()
                evs = []
                with torch.cuda.stream(copy_stream):
                    for c in range(pieces):
                        r0, r1 = c * per, min(n_reads, (c + 1) * per)
                        db[r0 * unit:r1 * unit].copy_(hb[r0 * unit:r1 * unit], non_blocking=True)
                        if hq is not None:
                            dq[r0 * READ_LEN:r1 * READ_LEN].copy_(hq[r0 * READ_LEN:r1 * READ_LEN], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(copy_stream)
                        evs.append(ev)
                if not copy_only:
                    for c in range(pieces):
                        r0, r1 = c * per, min(n_reads, (c + 1) * per)
                        lib_stream.wait_event(evs[c])
                        if packed:
                            sp.buildKmerSpectrumTwoBitDevice(db.data_ptr(), tb_off.data_ptr() + 8 * r0, offsets.data_ptr() + 8 * r0, r1 - r0, (r1 - r0) * READ_LEN,
                                                             quals_ptr=None if dq is None else dq.data_ptr() + r0 * READ_LEN, uniform_quality=(33 + 40) if dq is None else 0, first_read_idx=r0)
                        else:
                            sp.buildKmerSpectrumDevice(db.data_ptr(), dq.data_ptr(), offsets.data_ptr() + 8 * r0, r1 - r0, (r1 - r0) * READ_LEN, r0)
                    sp.finalize(2)
                torch.cuda.synchronize()

            step()
            t1 = time.perf_counter()
            for _ in range(max(1, args.steps)):
                step()
            t_incl = (time.perf_counter() - t1) / max(1, args.steps)
            st = sp.stats()
            assert args.no_check or (st["raw_kmers"] == n_reads * kmers_per_read and st["unique_kmers"] == uniq_local), (st, uniq_local)
            t1 = time.perf_counter()
            step(copy_only=True)
            t_copy = time.perf_counter() - t1
            return {"pieces": pieces, "value_incl_h2d": n_reads * kmers_per_read / t_incl, "ms_per_step_incl_h2d": t_incl * 1e3, "h2d_ms": t_copy * 1e3, "h2d_GB": nbytes / 1e9,
                    "h2d_GBps": nbytes / t_copy / 1e9}

        text = leg(False, 8)
        text["how"] = "bases + quals as text (%.1f GB) from pinned host memory in %d pieces on a copy stream, each piece built as it arrives (kmr_add_reads_dev)" % (text["h2d_GB"], text["pieces"])
        h2d = leg(True, args.h2d_pieces)
        h2d["how"] = ("bases 2-bit packed as the reference's Read keeps them%s (%.2f GB) from pinned host memory in %d pieces on a copy stream, each piece unpacked and built "
                      "on the device as it arrives (kmr_add_reads_twobit_dev)" % (", one quality character for all bases" if uniform_q else " + quals as text", h2d["h2d_GB"], h2d["pieces"]))
        h2d["as_text"] = text

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_kmers / (dt / args.steps)
        kb = (K + 3) // 4
        # algorithmic bytes: K * (2L/(L-k+1) + kb + 24) + D * kb for the k-mers this rank handled; achieved = those
        # bytes / the WHOLE step (wall clock between the barriers: exchange, host gaps and finalize included)
        alg_bytes = raw_local * (2.0 * READ_LEN / kmers_per_read + kb + 24) + uniq_local * kb
        achieved = alg_bytes / (ms_per_step / 1e3)
        build_ms, fin_ms = ktimes[0][0], ktimes[1][0]
        mode = {1: "device-table", 2: "two-level k-mer partition", 3: "super-k-mer lists"}[mode_num]
        out = {
            "metric": "total k-mers/sec at k=31, 150 bp reads (spectrum build, inputs resident in HBM)",
            "value": value, "unit": "kmers/s",
            "distinct_kmers_per_sec": uniq_total / (dt / args.steps),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%s: k=31, %d synthetic 150 bp reads per GPU, seed %d, genome %d bp (30x), 1%% substitutions, %s, "
                                   "min-depth 2, %s" % ("C3 (BASELINE.json configs[2]: 100 M reads over 8 GPUs)" if c3 else ("C2" if world == 1 else "C2's batch per GPU, weak scaling"), n_reads, seed, genome_len, "flat Q40" if args.quality == "flat" else "noisy qualities (SURVEY 8d)",
                                                        "single hash partition" if world == 1 else "owner-partitioned RCCL all-to-all over %d GPUs" % world),
                       "k": K, "read_len": READ_LEN, "reads_per_gpu": n_reads, "total_kmers": total_kmers, "good_kmers_rank0": st["raw_good_kmers"],
                       "distinct_kmers": uniq_total, "build_mode": mode, "quality": args.quality,
                       "parallelism": "1 process per GPU, owner partition x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": None,
                         "algorithmic_bytes_per_step": alg_bytes, "step_ms": ms_per_step,
                         "hip_event_ms_per_step": {"build": build_ms / max(1, args.steps), "finalize": fin_ms / max(1, args.steps)}},
        }
        if args.build_mode != 1:
            # per-kernel view of the same step: every kernel against the bytes its own role has to move, each timed with HIP
            # events on the handle's stream; rocprofv3 averages of the same command are in profiles/ (kernel_stats csv)
            weak = st["weak_entries"]
            if mode == "super-k-mer lists":
                sk_bytes = (3.8 if args.quality == "flat" else 9.4) * raw_local      # a 32-byte record per ~8.5 k-mers (+ 4 bytes per k-mer when the weights differ)
                lean = args.quality == "flat"      # one quality character: the launch goes to sk_extract_lean_kernel (bases only are read)
                per = [("sk_extract_lean_kernel (bases only: minimizer + runs + list scatter)" if lean else "sk_extract_kernel (extract + weight chain + minimizer + list scatter)", 2,
                        raw_local * (1.0 if lean else 2.0) * READ_LEN / kmers_per_read + sk_bytes,
                        "scattered device atomics: one returning 64-bit add per record, ~1.8e10/s chip-wide" if lean else "vector-instruction issue at 1.5 wavefronts per SIMD (LDS: 24 KB per wavefront)"),
                       ("sk_count_kernel (expand + count in LDS%s)" % (", one-weight form" if lean else ""), 5, sk_bytes + weak * 16.0,
                        "vector-instruction issue (VALU busy ~70 % of a SIMD's cycles at 4 wavefronts per SIMD; LDS caps the occupancy)"),
                       ("bb_hist + bb_scatter (x levels) + bb_group_kernel (radix partition by bucket, per-group sort)", 6, weak * (16.0 * 5 + 20.0),
                        "HBM / L2 transactions")]
            else:
                rec = 8 * ((kb + 7) // 8) + 8
                per = [("extract_kernel<LinearOp>", 2, raw_local * (2.0 * READ_LEN / kmers_per_read + rec), "HBM write stream"),
                       ("partition_direct_kernel<level 1>", 3, raw_local * 2.0 * rec, "scattered 16-byte stores"),
                       ("partition_direct_kernel<level 2>", 4, raw_local * 2.0 * rec, "scattered 16-byte stores"),
                       ("count_kernel", 5, raw_local * rec + weak * 20.0, "LDS latency"),
                       ("bucket scan + entry_scatter_kernel + sort_buckets_kernel", 6, weak * 60.0, "HBM")]
            kernels = []
            for name, grp, nbytes, limiter in per:
                ms, launches = ktimes[grp]
                if launches == 0:
                    continue
                ms_step = ms / max(1, args.steps)
                kernels.append({"name": name, "ms_per_step": ms_step, "launches_per_step": launches // max(1, args.steps),
                                "ms_per_launch": ms / launches, "bytes_per_step": nbytes,
                                "achieved_GBps": nbytes / (ms_step / 1e3) / 1e9, "frac": nbytes / (ms_step / 1e3) / HBM_PEAK, "limiter": limiter})
            out["roofline"]["limiter_source"] = "profiles/r04_sq_counters.json (SQ / LDS / L2 counters per kernel, tools/sq_counters.sh)"
            out["roofline"]["kernels"] = kernels
            if kernels:
                dom = max(kernels, key=lambda k: k["ms_per_step"])
                out["roofline"]["dominant_kernel"] = {"name": dom["name"], "ms_per_launch": dom["ms_per_launch"],
                                                      "bytes_per_launch": dom["bytes_per_step"] / max(1, dom["launches_per_step"]),
                                                      "achieved_GBps": dom["achieved_GBps"], "frac": dom["frac"]}
        out["oracle_check"] = oracle_check
        if exchange:
            out["exchange"] = {k: (v / max(1, args.steps) if isinstance(v, (int, float)) else v) for k, v in xstats.items()}
        # HBM bytes from the PMC counters: bench.py cannot run rocprofv3 on itself, so the committed summary of the same command
        # (tools/pmc.sh -> profiles/r04_pmc_traffic_<quality>.json) is quoted when it describes this workload, quality mode and build mode
        # (exactly one file may describe a line: the newest round's summary whose workload, quality mode and build mode are this line's)
        for tf in ("r04_pmc_traffic_%s.json" % args.quality, "r03_pmc_traffic.json" if args.quality == "flat" else "r03_pmc_traffic_noisy.json"):
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", tf)))
            except Exception:
                continue
            if tj.get("build_mode") == mode and tj.get("quality") == args.quality and tj.get("reads", 10_000_000) == n_reads and n_reads == 10_000_000 and world == 1:
                out["roofline"]["traffic"] = tj["hot_path_total_GB_per_step"] * 1e9
                out["roofline"]["traffic_source"] = "profiles/%s (bytes per step, FETCH_SIZE x2 corrected; measured at commit %s)" % (tf, tj.get("commit", "?"))
                break
        if h2d:
            # SURVEY 8(d)'s t_build runs from "first byte of in-memory reads available" to the queryable table, the host-to-device copy
            # included: that figure, beside `value` (inputs resident in HBM, as the bench contract asks)
            out.update({"value_incl_h2d": h2d["value_incl_h2d"], "t_build_ms_incl_h2d": h2d["ms_per_step_incl_h2d"], "h2d_ms": h2d["h2d_ms"]})
            out["h2d"] = h2d
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(torch, min(args.cpu_reads, n_reads), args.quality, dev)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
