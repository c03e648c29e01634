"""development tool: the C2 batch with a share of its reads replaced by one homopolymer read (a single k-mer, a single super-k-mer
list that every lane of the chip appends to and one block has to count): what a hot minimizer costs.
usage: tools/hot_list.py [reads] [every n-th read is poly-A] [build_mode]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import kmernator_amd as ka
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 50
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda", 0)
bases, quals, offsets = bench.gen_reads(torch, n, 5 * n, 1, 0, dev, "flat")
L = bench.READ_LEN
if every:
    b2 = bases[:n * L].view(n, L)
    b2[::every] = ord("A")
total = n * L
sp = ka.KmerSpectrum(ka.default_config(bench.K, estimated_raw_kmers=n * (L - bench.K + 1), device=0, build_mode=mode))
for rep in range(3):
    sp.reset(); sp.kernel_time_reset(); torch.cuda.synchronize(); t0 = time.time()
    sp.buildKmerSpectrumDevice(bases.data_ptr(), quals.data_ptr(), offsets.data_ptr(), n, total)
    sp.sync(); t1 = time.time()
    sp.finalize(2); torch.cuda.synchronize(); t2 = time.time()
    print("rep %d: every %d-th read poly-A, mode %d: build %.1f ms, finalize %.1f ms; extract %.1f count %.1f buckets %.1f" % (
        rep, every, mode, (t1 - t0) * 1e3, (t2 - t1) * 1e3, sp.kernel_time(2)[0], sp.kernel_time(5)[0], sp.kernel_time(6)[0]), flush=True)
