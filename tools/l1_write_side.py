"""development tool: the level-1 partition pass with its input coming out of L2 (every tile re-reads one of the first N tiles, KMR_DEBUG_SAME_TILE=N)
beside the normal pass -- the write side alone, i.e. what the pass would cost if extract handed it the records in registers"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import kmernator_amd as ka
n = 10_000_000
dev = torch.device("cuda", 0)
bases, quals, offsets = bench.gen_reads(torch, n, 5 * n, 1, 0, dev)
torch.cuda.synchronize()
for same in (0, 1, 64, 192, 1024, 0):
    if same:
        os.environ["KMR_DEBUG_SAME_TILE"] = str(same)
    else:
        os.environ.pop("KMR_DEBUG_SAME_TILE", None)
    sp = ka.KmerSpectrum(ka.default_config(31, estimated_raw_kmers=n * 120, device=0))
    for rep in range(2):
        sp.kernel_time_reset()
        sp.buildKmerSpectrumDevice(bases.data_ptr(), quals.data_ptr(), offsets.data_ptr(), n, n * 150, 0)
        sp.sync()
        ex, _ = sp.kernel_time(2)
        l1, nl = sp.kernel_time(3)
        if rep == 0:
            sp.reset()
    print("same_tile=%d: extract %.2f ms, level 1 %.2f ms (%d launches)" % (same, ex, l1, nl), flush=True)
    del sp
