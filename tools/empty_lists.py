import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, kmernator_amd as ka, bench
dev = torch.device("cuda", 0)
n = 100000
bases, quals, offsets = bench.gen_reads(torch, n, 5 * n, 1, 0, dev, "flat")
for est in (1.2e9, 9.6e9):
    sp = ka.KmerSpectrum(ka.default_config(31, estimated_raw_kmers=int(est), device=0, build_mode=3))
    for rep in range(3):
        sp.reset(); sp.kernel_time_reset(); torch.cuda.synchronize(); t0 = time.time()
        sp.buildKmerSpectrumDevice(bases.data_ptr(), quals.data_ptr(), offsets.data_ptr(), n, n * 150); sp.sync(); t1 = time.time()
        sp.finalize(2); torch.cuda.synchronize(); t2 = time.time()
    print("lists for %.1e k-mers, 100 K reads: build %.2f ms, finalize %.2f ms (count kernel %.2f, buckets %.2f)" % (est, (t1 - t0) * 1e3, (t2 - t1) * 1e3, sp.kernel_time(5)[0], sp.kernel_time(6)[0]), flush=True)
