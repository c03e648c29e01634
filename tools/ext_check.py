"""development tool: extension values (Meraculous) at scale, streaming vs device-table mode"""
import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
import bench, kmernator_amd as ka
from helpers import KMR_MAP_WEAK, KMR_VALUE_EXT
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 21
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
quality = sys.argv[3] if len(sys.argv) > 3 else "flat"
bases, quals, offsets = bench.gen_reads(torch, n, 5 * n, 3, 0, dev, quality=quality)
torch.cuda.synchronize()
imgs = {}
modes = tuple(int(x) for x in os.environ.get("EXT_MODES", "3,2,1").split(","))
tune = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in sys.argv[4:]}      # knob=value ...
for mode in modes:
    sp = ka.KmerSpectrum(ka.default_config(k, estimated_raw_kmers=n * (150 - k + 1), device=0, build_mode=mode, value_kind=KMR_VALUE_EXT, min_weight=0.0, min_quality_score=2)).tune(**tune)
    for rep in range(2):
        sp.reset(); sp.kernel_time_reset(); torch.cuda.synchronize(); t0 = time.time()
        sp.buildKmerSpectrumDevice(bases.data_ptr(), quals.data_ptr(), offsets.data_ptr(), n, n * 150, 0)
        sp.finalize(2); dt = time.time() - t0
    print("  kernel groups", [round(sp.kernel_time(g)[0], 1) for g in range(7)])
    st = sp.stats(); print("mode", mode, "k", k, "%.1f ms" % (dt * 1e3), "%.2f G kmers/s" % (st["raw_kmers"] / dt / 1e9), st, flush=True)
    imgs[mode] = sp.image(KMR_MAP_WEAK); del sp
if len(modes) == 3: print("images identical: 2 vs 1", np.array_equal(imgs[1], imgs[2]), " 3 vs 1", np.array_equal(imgs[1], imgs[3]))
