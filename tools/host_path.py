"""development tool: C2 through the host-pointer entry point kmr_add_reads (PCIe-inclusive rate)"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, numpy as np
import bench, kmernator_amd as ka
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
bases, quals, offsets = bench.gen_reads(torch, n, 5 * n, 1, 0, dev)
hb, hq, ho = bases.cpu().numpy(), quals.cpu().numpy(), offsets.cpu().numpy().astype(np.uint64)
sp = ka.KmerSpectrum(ka.default_config(31, estimated_raw_kmers=n * 120, device=0))
for rep in range(3):
    sp.reset(); torch.cuda.synchronize(); t0 = time.time()
    sp.buildKmerSpectrum(hb, hq, ho)
    t1 = time.time(); sp.finalize(2); t2 = time.time()
    print("rep %d: add_reads (H2D + extract + level 1) %.1f ms, finalize %.1f ms -> %.2f G k-mers/s" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, n * 120 / (t2 - t0) / 1e9), flush=True)
