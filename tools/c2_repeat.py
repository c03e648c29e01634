"""development tool: repeat full-size builds over changing inputs; print conservation defect + image checksum"""
import sys, os, zlib
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import bench
import kmernator_amd as ka
from helpers import KMR_MAP_WEAK
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
nseeds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0)
seen = {}
p = ka.KmerSpectrum(ka.default_config(31, estimated_raw_kmers=n * 120, device=0))
bad = 0
for rep in range(reps):
    seed = 1 + rep % nseeds
    bases, quals, offsets = bench.gen_reads(torch, n, 5 * n, seed, 0, dev)
    torch.cuda.synchronize()
    if rep % 5 == 4:
        p = ka.KmerSpectrum(ka.default_config(31, estimated_raw_kmers=n * 120, device=0))   # fresh handle, recycled memory
    p.reset()
    p.buildKmerSpectrumDevice(bases.data_ptr(), quals.data_ptr(), offsets.data_ptr(), n, n * 150, 0)
    p.finalize(2)
    st = p.stats()
    hist = p.histogram(4096)[0]
    cons = int((hist * np.arange(hist.size, dtype=np.uint64)).sum()) + st["singleton_kmers"] - st["raw_good_kmers"]
    crc = zlib.crc32(p.image(KMR_MAP_WEAK).tobytes())
    key = (st["unique_kmers"], st["weak_entries"], st["singleton_kmers"], crc)
    ok = cons == 0 and seen.setdefault(seed, key) == key
    bad += not ok
    print(rep, "seed", seed, key, "defect", cons, "OK" if ok else "MISMATCH", flush=True)
print("bad", bad, "of", reps)
