"""development tool: what counting half of the list space early (kmr_count_lists_prefix) costs on ONE GPU, where nothing is on a wire to
hide it behind: C2's batch, kmr_finalize alone against kmr_count_lists_prefix(half) + kmr_finalize.  usage: tools/early_count_cost.py [reads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import kmernator_amd as ka
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
b, q, o = bench.gen_reads(torch, n, 5 * n, 1, 0, dev)
sp = ka.KmerSpectrum(ka.default_config(bench.K, estimated_raw_kmers=n * 120, device=0))
def run(share):
    sp.reset(); sp.kernel_time_reset()
    sp.buildKmerSpectrumDevice(b.data_ptr(), q.data_ptr(), o.data_ptr(), n, n * 150, 0); sp.sync()
    torch.cuda.synchronize(); t0 = time.time()
    if share > 0:
        sp.count_lists_prefix(2, int(sp.build_info("lists") * share)); sp.sync()
    t1 = time.time()
    sp.finalize(2); torch.cuda.synchronize(); t2 = time.time()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3, sp.build_info("early_entries"), sp.stats()["weak_entries"]
for share in (0.0, 0.5, 0.0, 0.5, 0.25, 0.75, 1.0):
    run(share)
    e, f, ee, we = run(share)
    print("early share %.2f: early count %.2f ms + finalize %.2f ms = %.2f ms  (%d of %d weak entries counted early)" % (share, e, f, e + f, ee, we), flush=True)
