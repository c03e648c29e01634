"""development tool: C2 through the host-pointer entry points (PCIe-inclusive, pageable host memory): kmr_add_reads (text) against
kmr_add_reads_twobit (packed bases, one quality character), whose pieces overlap the copy with the build inside the library"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, numpy as np
import bench, kmernator_amd as ka
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
piece = float(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
bases, quals, offsets = bench.gen_reads(torch, n, 5 * n, 1, 0, dev)
L = 150; PB = 38
c = bases[:n * L].view(n, L); c = ((c >> 1) & 3) ^ ((c >> 2) & 1)
c = torch.nn.functional.pad(c, (0, PB * 4 - L)).view(n, PB, 4)
tw = (c[:, :, 0] << 6 | c[:, :, 1] << 4 | c[:, :, 2] << 2 | c[:, :, 3]).reshape(-1).cpu().numpy()
to = (np.arange(n + 1, dtype=np.uint64) * np.uint64(PB))
hb, hq, ho = bases[:n * L].cpu().numpy(), quals[:n * L].cpu().numpy(), offsets.cpu().numpy().astype(np.uint64)
sp = ka.KmerSpectrum(ka.default_config(31, estimated_raw_kmers=n * 120, device=0))
if piece: sp.tune(twobit_piece_bases=piece)
for rep in range(3):
    sp.reset(); torch.cuda.synchronize(); t0 = time.time()
    sp.buildKmerSpectrumTwoBit(tw, to, ho, uniform_quality=33 + 40)
    t1 = time.time(); sp.finalize(2); t2 = time.time()
    print("packed rep %d: add_reads %.1f ms, finalize %.1f ms -> %.2f G k-mers/s  %s" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, n * 120 / (t2 - t0) / 1e9, sp.stats()["unique_kmers"]), flush=True)
for rep in range(2):
    sp.reset(); torch.cuda.synchronize(); t0 = time.time()
    sp.buildKmerSpectrum(hb, hq, ho)
    t1 = time.time(); sp.finalize(2); t2 = time.time()
    print("text rep %d: add_reads %.1f ms, finalize %.1f ms -> %.2f G k-mers/s  %s" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, n * 120 / (t2 - t0) / 1e9, sp.stats()["unique_kmers"]), flush=True)
