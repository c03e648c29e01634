"""development tool: BASELINE config 3's WHOLE input (100 M synthetic 150 bp reads of a 500 Mbp genome, seed 2, k = 31: 1.2e10 k-mers --
what bench.py --gpus 8 spreads over eight GPUs, rank r's reads generated as bench.py generates them) built on ONE GPU in eight calls,
each rank's batch generated, fed and dropped in turn; the time is that of the build calls and kmr_finalize (generation excluded).
The single-GPU figure the 8-GPU run is to be set against.  usage: tools/c3_one_gpu.py [ranks=8] [reads per rank=12500000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import kmernator_amd as ka
import bench
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12_500_000
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
L, K = bench.READ_LEN, bench.K
per = L - K + 1
sp = ka.KmerSpectrum(ka.default_config(K, estimated_raw_kmers=n * per * world, device=0))
for rep in range(2):
    sp.reset(); sp.kernel_time_reset(); t_build = 0.0; calls = []
    for r in range(world):
        b, q, o = bench.gen_reads(torch, n, 500_000_000 if world == 8 and n == 12_500_000 else 5 * n * world, 2, r, dev)
        torch.cuda.synchronize(); t0 = time.time()
        sp.buildKmerSpectrumDevice(b.data_ptr(), q.data_ptr(), o.data_ptr(), n, n * L, r * n); sp.sync()
        t_build += time.time() - t0; calls.append(round((time.time() - t0) * 1e3, 1))
        del b, q, o
    torch.cuda.synchronize(); t0 = time.time()
    sp.finalize(2); torch.cuda.synchronize(); t_fin = time.time() - t0
    st = sp.stats()
    hist = sp.histogram(4096)[0]
    cons = int((hist * np.arange(hist.size, dtype=np.uint64)).sum()) + st["singleton_kmers"] - st["raw_good_kmers"]
    tot = t_build + t_fin
    print("rep %d: %d reads, %.3e k-mers: build calls %.1f ms + finalize %.1f ms = %.1f ms -> %.2f G k-mers/s; kernel groups %s; raw %d good %d unique %d weak %d; conservation defect %d, hist sum - weak %d; mem %.1f GB" % (
        rep, n * world, st["raw_kmers"], t_build * 1e3, t_fin * 1e3, tot * 1e3, st["raw_kmers"] / tot / 1e9, [round(sp.kernel_time(g)[0], 1) for g in range(7)],
        st["raw_kmers"], st["raw_good_kmers"], st["unique_kmers"], st["weak_entries"], cons, int(hist.sum()) - st["weak_entries"], torch.cuda.mem_get_info()[0] / 1e9), flush=True)
    print("   ms per call:", calls, flush=True)
    assert st["raw_kmers"] == n * world * per and cons == 0 and int(hist.sum()) == st["weak_entries"]
    if world == 8 and n == 12_500_000:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import digests_agree, full_size_golden
        try:
            g = full_size_golden("c3_flat")
            print("   statistics == oracle:", st == g["stats"], " weak digest == oracle:", digests_agree(sp.digest(0), g["weak_digest"], 1e-6), flush=True)
        except KeyError:
            print("   (no oracle digest of config 3 under tests/golden yet)")
