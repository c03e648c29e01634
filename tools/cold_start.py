"""development tool: what the first build of a handle costs beside a warm one (C2 batch)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import kmernator_amd as ka
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda", 0)
bases, quals, offsets = bench.gen_reads(torch, n, 5 * n, 1, 0, dev)
torch.cuda.synchronize()
t0 = time.time()
sp = ka.KmerSpectrum(ka.default_config(31, estimated_raw_kmers=n * 120, device=0))
t1 = time.time()
print("create %.1f ms" % ((t1 - t0) * 1e3))
for rep in range(3):
    t0 = time.time(); sp.reset(); t1 = time.time()
    sp.buildKmerSpectrumDevice(bases.data_ptr(), quals.data_ptr(), offsets.data_ptr(), n, n * 150, 0); sp.sync(); t2 = time.time()
    sp.finalize(2); t3 = time.time()
    print("build %d: reset %.1f ms, add_reads %.1f ms, finalize %.1f ms" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
from kmernator_amd import KMR_MAP_WEAK
for rep in range(2):
    t0 = time.time(); img = sp.image(KMR_MAP_WEAK); t1 = time.time()
    print("weak image export %d: %.1f MB in %.1f ms (device pack + copy to pageable host memory)" % (rep, img.size / 1e6, (t1 - t0) * 1e3), flush=True)
