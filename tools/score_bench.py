"""development tool: scoreAndTrimReads of a C2-size batch (10 M x 150 bp, the bench's generator) on a device-resident read batch:
k-mer counts by the streaming pass over minimizer lists against the per-k-mer probes of the lookup table.
usage: tools/score_bench.py [reads] [stream_lookups 0|1] [quality flat|noisy]"""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import kmernator_amd as ka
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
stream = int(sys.argv[2]) if len(sys.argv) > 2 else 1
quality = sys.argv[3] if len(sys.argv) > 3 else "flat"
dev = torch.device("cuda", 0)
bases, quals, offsets = bench.gen_reads(torch, n, 5 * n, 1234, 0, dev, quality)
total = n * bench.READ_LEN
sp = ka.KmerSpectrum(ka.default_config(bench.K, estimated_raw_kmers=n * (bench.READ_LEN - bench.K + 1), device=0)).tune(stream_lookups=stream)
sp.buildKmerSpectrumDevice(bases.data_ptr(), quals.data_ptr(), offsets.data_ptr(), n, total)
sp.finalize(2)
print("spectrum:", {k: v for k, v in sp.stats().items() if k in ("raw_good_kmers", "unique_kmers", "weak_entries")}, flush=True)
lib = sp.lib
hb, hq, ho = bases[:total].cpu().numpy(), quals[:total].cpu().numpy(), offsets.cpu().numpy().astype(np.uint64)
r = C.c_void_p()
lib.kmr_reads_from_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_void_p)]
rc = lib.kmr_reads_from_host(sp.h, hb.ctypes.data_as(C.c_void_p), hq.ctypes.data_as(C.c_void_p), ho.ctypes.data_as(C.POINTER(C.c_uint64)), n, C.byref(r))
assert rc == 0, lib.kmr_last_error(sp.h)
to = np.zeros(n, np.uint32); tl = np.zeros(n, np.uint32); sc = np.zeros(n, np.float32); wt = np.zeros(n, np.uint8)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    rc = lib.kmr_score_read_batch(sp.h, r, 2.0, 1, to.ctypes.data_as(C.POINTER(C.c_uint32)), tl.ctypes.data_as(C.POINTER(C.c_uint32)), sc.ctypes.data_as(C.POINTER(C.c_float)), wt.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert rc == 0, lib.kmr_last_error(sp.h)
    dt = time.time() - t0
    print("rep %d: scoreAndTrimReads of %d reads (%d k-mers) %.1f ms -> %.2f G lookups/s; trimmed %d, median of medians %.0f, checksum %d" % (
        rep, n, n * (bench.READ_LEN - bench.K + 1), dt * 1e3, n * (bench.READ_LEN - bench.K + 1) / dt / 1e9, int(wt.sum()), float(np.median(sc)), int(sc.astype(np.float64).sum())), flush=True)
