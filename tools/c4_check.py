"""development tool: k=51 (two-word keys) at scale, both build modes"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, numpy as np
import bench, kmernator_amd as ka
from helpers import KMR_MAP_WEAK
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 51
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
bases, quals, offsets = bench.gen_reads(n, 5 * n, 3, 0, dev)
torch.cuda.synchronize()
imgs = {}
for mode in ((2, 1) if len(sys.argv) < 4 else (2,)):
    sp = ka.KmerSpectrum(ka.default_config(k, estimated_raw_kmers=n * (150 - k + 1), device=0, build_mode=mode))
    for rep in range(2):
        sp.reset(); torch.cuda.synchronize(); t0 = time.time()
        sp.buildKmerSpectrumDevice(bases.data_ptr(), quals.data_ptr(), offsets.data_ptr(), n, n * 150, 0)
        t1 = time.time(); sp.finalize(2); dt = time.time() - t0
        print("  rep", rep, "build %.1f ms finalize %.1f ms" % ((t1 - t0) * 1e3, (time.time() - t1) * 1e3), "kernel groups", [round(sp.kernel_time(g)[0], 1) for g in range(7)], flush=True)
        sp.kernel_time_reset()
    st = sp.stats(); print("mode", mode, "k", k, "%.1f ms" % (dt * 1e3), "%.2f G kmers/s" % (st["raw_kmers"] / dt / 1e9), st)
    imgs[mode] = sp.image(KMR_MAP_WEAK); del sp
print("images identical:", np.array_equal(imgs[1], imgs[2]))
