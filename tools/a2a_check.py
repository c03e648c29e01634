"""development tool: does a (1-rank) RCCL all_to_all move payloads beyond 1 GiB intact?"""
import os, sys, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for gib in (0.5, 1.0, 1.5, 1.9, 2.5, 3.5):
    n = int(gib * (1 << 30)) // 12
    src = torch.arange(n * 3, dtype=torch.int32, device="cuda").view(n, 3)
    dst = torch.zeros_like(src)
    dist.all_to_all([dst], [src])
    torch.cuda.synchronize()
    bad = int((dst != src).any(dim=1).sum().item())
    dst2 = torch.zeros_like(src)
    dist.all_to_all_single(dst2, src, output_split_sizes=[n], input_split_sizes=[n])
    torch.cuda.synchronize()
    bad2 = int((dst2 != src).any(dim=1).sum().item())
    print("%.1f GiB: list form bad rows %d, single form bad rows %d of %d" % (gib, bad, bad2, n), flush=True)
dist.destroy_process_group()
