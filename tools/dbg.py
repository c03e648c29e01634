import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import *
import kmernator_amd as ka
for k in [int(a) for a in sys.argv[1:]]:
    rb = synth_reads(500, read_len=170, seed=3)
    c = ka.default_config(k, build_mode=2, num_buckets_weak=64, num_buckets_singleton=64)
    p = ka.KmerSpectrum(c)
    try:
        p.buildKmerSpectrum(rb.bases, rb.quals, rb.offsets); p.finalize(2); print(k, "ok", p.stats()["weak_entries"])
    except Exception as e:
        print(k, "ERR", e)
