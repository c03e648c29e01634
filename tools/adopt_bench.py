"""development tool: what pack and adopt cost, and what the lists an owner ends up with cost in the count pass.  An owner handle
(rank 0 of `world`) and a sender handle (rank 1) share one GPU: the batch is cut into `world` slices, the owner extracts slice 0
itself, the sender extracts every other slice, packs, and the owner adopts the segment meant for it -- the owner's lists are then
assembled from `world` partly filled pieces each, as in a real job (at 1 / world of its size).  usage: tools/adopt_bench.py [reads] [world]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import kmernator_amd as ka
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
bases, quals, offsets = bench.gen_reads(torch, n, 5 * n, 1, 0, dev, "flat")
L = bench.READ_LEN
mk = lambda r: ka.KmerSpectrum(ka.default_config(bench.K, estimated_raw_kmers=n * 120, device=0, rank=r, world_size=world, build_mode=3))
owner, sender = mk(0), mk(min(1, world - 1))
for rep in range(3):
    owner.reset(); owner.sk_exchange_begin(); owner.kernel_time_reset()
    tp = ta = 0.0; got_c = 0; got_gb = 0.0
    for s in range(world):
        lo, hi = n * s // world, n * (s + 1) // world
        h = owner if s == 0 else sender
        if s: sender.reset(); sender.sk_exchange_begin()
        h.set_stream_origin(lo * L)
        h.buildKmerSpectrumDevice(bases.data_ptr(), quals.data_ptr(), offsets.data_ptr() + 8 * lo, hi - lo, (hi - lo) * L); h.sync()
        chunks, granules = h.sk_exchange_counts()
        me = 0 if s == 0 else 1
        send_c = [int(c) if r != me else 0 for r, c in enumerate(chunks)]; send_g = [int(g) if r != me else 0 for r, g in enumerate(granules)]
        goff = [int(x) for x in np.concatenate([[0], np.cumsum(send_g)[:-1]])]; coff = [int(x) for x in np.concatenate([[0], np.cumsum(send_c)[:-1]])]
        data = torch.empty((max(sum(send_g), 1), 4), dtype=torch.int32, device=dev); meta = torch.empty((max(sum(send_c), 1), 2), dtype=torch.int32, device=dev)
        torch.cuda.synchronize(); t3 = time.time()
        h.sk_exchange_pack(data.data_ptr(), meta.data_ptr(), goff, coff); t4 = time.time()
        tp += t4 - t3
        if s:
            torch.cuda.synchronize(); t4 = time.time()
            owner.sk_exchange_adopt(data[goff[0]:].data_ptr(), meta[coff[0]:].data_ptr(), send_c[0], send_g[0]); owner.sync(); t5 = time.time()
            ta += t5 - t4; got_c += send_c[0]; got_gb += 16 * send_g[0] / 1e9
    torch.cuda.synchronize(); t5 = time.time()
    owner.finalize(2); torch.cuda.synchronize(); t6 = time.time()
    st = owner.stats()
    print("rep %d, owner of 1/%d of the lists: pack (all senders) %.1f ms, adopt %.1f ms (%d chunks, %.2f GB), finalize %.1f ms (count %.2f ms for %d distinct k-mers)" % (
        rep, world, tp * 1e3, ta * 1e3, got_c, got_gb, (t6 - t5) * 1e3, owner.kernel_time(5)[0], st["unique_kmers"]), flush=True)
