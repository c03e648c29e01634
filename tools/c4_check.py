"""development tool: k = 51 (two-word keys) at scale: the default build (super-k-mer lists) and, for the image comparison, the
device-table build.  usage: tools/c4_check.py [reads] [k] [read_len] [modes e.g. 3,2,1] [knob=value ...]
With 50000000 51 150 the input is BASELINE config 4 (seed 3, 250 Mbp genome) and the result is held to the oracle's
digest under tests/golden/full_size_digests.json."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
import bench, kmernator_amd as ka
from helpers import KMR_MAP_WEAK
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 51
L = int(sys.argv[3]) if len(sys.argv) > 3 else 100
modes = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [3, 1]
tune = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in sys.argv[5:]}
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
bases, quals, offsets = bench.gen_reads(torch, n, 5 * n, 3, 0, dev, "flat", read_len=L)
torch.cuda.synchronize()
imgs = {}
for mode in modes:
    sp = ka.KmerSpectrum(ka.default_config(k, estimated_raw_kmers=n * (L - k + 1), device=0, build_mode=mode)).tune(**tune)
    for rep in range(2):
        sp.reset(); sp.kernel_time_reset(); torch.cuda.synchronize(); t0 = time.time()
        sp.buildKmerSpectrumDevice(bases.data_ptr(), quals.data_ptr(), offsets.data_ptr(), n, n * L, 0)
        sp.sync(); t1 = time.time(); sp.finalize(2); dt = time.time() - t0
        print("  rep", rep, "build %.1f ms finalize %.1f ms" % ((t1 - t0) * 1e3, (time.time() - t1) * 1e3), "kernel groups", [round(sp.kernel_time(g)[0], 1) for g in range(7)], "launches", [sp.kernel_time(g)[1] for g in range(7)], flush=True)
    st = sp.stats(); print("mode", mode, "k", k, "%d x %d bp: %.1f ms" % (n, L, dt * 1e3), "%.2f G kmers/s" % (st["raw_kmers"] / dt / 1e9), st, flush=True)
    if (n, k, L) == (50_000_000, 51, 150):
        from helpers import digests_agree, full_size_golden
        g = full_size_golden("c4_flat")
        print("  statistics == oracle:", st == g["stats"], " weak digest == oracle:", digests_agree(sp.digest(KMR_MAP_WEAK), g["weak_digest"], 1e-6), flush=True)
    if len(modes) > 1:
        imgs[mode] = sp.image(KMR_MAP_WEAK)
    del sp
if len(imgs) > 1:
    a = list(imgs.values())
    print("images identical:", all(np.array_equal(a[0], x) for x in a[1:]))
