import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import kmernator_amd as ka
from helpers import synth_reads, default_config, OracleSpectrum, KMR_MAP_WEAK
for k, n in ((51, 300), (51, 3000), (33, 3000), (95, 3000)):
    rb = synth_reads(n, read_len=150, genome_len=n * 12, seed=5, err=0.01)
    cfg = ka.default_config(k, estimated_raw_kmers=n * (150 - k + 1), build_mode=3)
    p = ka.KmerSpectrum(cfg)
    p.buildKmerSpectrum(rb.bases, rb.quals, rb.offsets)
    try:
        p.finalize(2)
        st = p.stats()
        o = OracleSpectrum(default_config(k, estimated_raw_kmers=n * (150 - k + 1)))
        o.add_reads(rb); o.finalize(2)
        print(k, n, "ok", st["unique_kmers"], o.stats()["unique_kmers"], st["weak_entries"], o.stats()["weak_entries"], flush=True)
    except Exception as e:
        print(k, n, "FAIL", str(e)[:150], flush=True)
