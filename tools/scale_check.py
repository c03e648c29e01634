"""development tool: compare build modes (and optionally the oracle) at scale"""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from helpers import *
import kmernator_amd as ka
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
use_oracle = len(sys.argv) > 2
rb = synth_reads(n, read_len=150, genome_len=5 * n, seed=1)
res = {}
for mode in (1, 2):
    c = ka.default_config(31, estimated_raw_kmers=n * 120, build_mode=mode)
    p = ka.KmerSpectrum(c)
    t0 = time.time()
    p.buildKmerSpectrum(rb.bases, rb.quals, rb.offsets); p.finalize(2)
    st = p.stats(); img = p.image(KMR_MAP_WEAK)
    print("mode", mode, st, "%.2fs" % (time.time() - t0)); res[mode] = (st, img)
a, b = res[1][1], res[2][1]
print("images equal size", a.size == b.size, "bytes equal", np.array_equal(a, b))
if a.size == b.size:
    d = np.nonzero(a != b)[0]
    print("differing bytes", d.size, d[:20])
if use_oracle:
    o = OracleSpectrum(default_config(31, estimated_raw_kmers=n * 120)); o.add_reads(rb, threads=8); o.finalize(2)
    print("oracle", o.stats())
