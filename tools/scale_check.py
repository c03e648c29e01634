"""development tool: compare build modes (and optionally the oracle) at scale"""
import sys, time, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from helpers import *
import kmernator_amd as ka
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
use_oracle = len(sys.argv) > 2
rb = synth_reads(n, read_len=150, genome_len=5 * n, seed=1)
res = {}
modes = (2,) * int(os.environ.get('REPEAT2', '0')) if os.environ.get('REPEAT2') else (1, 2)
for mode in modes:
    c = ka.default_config(31, estimated_raw_kmers=n * 120, build_mode=mode)
    p = ka.KmerSpectrum(c)
    t0 = time.time()
    p.buildKmerSpectrum(rb.bases, rb.quals, rb.offsets); p.finalize(2)
    st = p.stats(); img = p.image(KMR_MAP_WEAK)
    hist = p.histogram(4096)[0]
    cons = int((hist * np.arange(hist.size, dtype=np.uint64)).sum()) + st['singleton_kmers'] - st['raw_good_kmers']
    print("mode", mode, st, "%.2fs" % (time.time() - t0), "conservation defect", cons, "hist0", int(hist[0]), "hist1", int(hist[1])); res[mode] = (st, img)
if os.environ.get('REPEAT2'): sys.exit(0)
a, b = res[1][1], res[2][1]
print("images equal size", a.size == b.size, "bytes equal", np.array_equal(a, b))
if a.size == b.size:
    d = np.nonzero(a != b)[0]
    print("differing bytes", d.size, d[:20])
if use_oracle:
    o = OracleSpectrum(default_config(31, estimated_raw_kmers=n * 120)); o.add_reads(rb, threads=8); o.finalize(2)
    print("oracle", o.stats())
