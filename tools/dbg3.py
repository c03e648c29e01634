import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, torch.distributed as dist
import bench, kmernator_amd as ka
from kmernator_amd.distributed import build_partitioned
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dev = torch.device("cuda", 0)
bases, quals, offsets = bench.gen_reads(n, 5 * n, 1, 0, dev)
torch.cuda.synchronize()
for mode in ("direct", "exchange"):
    sp = ka.KmerSpectrum(ka.default_config(31, estimated_raw_kmers=n * 120, device=0))
    t0 = time.time()
    if mode == "direct":
        sp.buildKmerSpectrumDevice(bases.data_ptr(), quals.data_ptr(), offsets.data_ptr(), n, n * 150, 0)
    else:
        build_partitioned(sp, bases, quals, offsets, chunk_reads=1 << 18)
    sp.finalize(2)
    print(mode, sp.stats(), "%.3fs" % (time.time() - t0))
dist.destroy_process_group()
