import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import *
import kmernator_amd as ka
for nreads, rl in ((256, 1000), (320, 1000), (192, 1000), (300, 600), (500, 300)):
    rb = synth_reads(nreads, read_len=rl, genome_len=400000, seed=41)
    out = []
    for mode in (1, 2):
        c = ka.default_config(31, build_mode=mode, num_buckets_weak=1024, num_buckets_singleton=4096)
        p = ka.KmerSpectrum(c)
        p.buildKmerSpectrum(rb.bases, rb.quals, rb.offsets); p.finalize(2); st = p.stats(); out.append((st["unique_kmers"], st["weak_entries"], st["raw_good_kmers"]))
    print(nreads, rl, out, "OK" if out[0] == out[1] else "MISMATCH")
