"""development tool: reproduce the rare full-size conservation defect (agree_at_scale followed by the C2 build)"""
import sys, os, zlib
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import bench
import kmernator_amd as ka
from helpers import *
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
os.environ["KMR_COUNT_CHECK"] = "6"
dev = torch.device("cuda", 0)
n3 = 3000000
rb = synth_reads(n3, read_len=150, genome_len=5 * n3, seed=1)
n = 10_000_000
bases, quals, offsets = bench.gen_reads(torch, n, 5 * n, 1, 0, dev)
torch.cuda.synchronize()
good = None
for it in range(iters):
    for mode in (1, 2):
        c = ka.default_config(31, estimated_raw_kmers=n3 * 120, build_mode=mode)
        q = ka.KmerSpectrum(c)
        q.buildKmerSpectrum(rb.bases, rb.quals, rb.offsets); q.finalize(2)
        img3 = q.image(KMR_MAP_WEAK)
    p = ka.KmerSpectrum(ka.default_config(31, estimated_raw_kmers=n * 120, device=0))
    for rep in range(2):
        p.reset()
        sys.stderr.write("== iter %d rep %d\n" % (it, rep)); sys.stderr.flush()
        p.buildKmerSpectrumDevice(bases.data_ptr(), quals.data_ptr(), offsets.data_ptr(), n, n * 150, 0)
        p.finalize(2)
        st = p.stats()
        hist = p.histogram(4096)[0]
        cons = int((hist * np.arange(hist.size, dtype=np.uint64)).sum()) + st["singleton_kmers"] - st["raw_good_kmers"]
        img = p.image(KMR_MAP_WEAK)
        crc = zlib.crc32(img.tobytes())
        print(it, rep, st["unique_kmers"], st["weak_entries"], st["singleton_kmers"], "defect", cons, "crc", crc, flush=True)
        if cons == 0 and good is None: good = img.copy()
        if cons != 0 and good is not None:
            nb = int(np.frombuffer(img[:8].tobytes(), dtype=np.uint64)[0])
            ob = np.frombuffer(img[16:16 + 8 * nb].tobytes(), dtype=np.uint64).astype(np.int64)
            og = np.frombuffer(good[16:16 + 8 * nb].tobytes(), dtype=np.uint64).astype(np.int64)
            def bucket(im, offs, b):
                o = int(offs[b]); cnt = int(np.frombuffer(im[o:o + 4].tobytes(), dtype=np.uint32)[0])
                keys = np.frombuffer(im[o + 4:o + 4 + 8 * cnt].tobytes(), dtype=">u8")
                vals = np.frombuffer(im[o + 4 + 8 * cnt:o + 4 + 20 * cnt].tobytes(), dtype=np.uint32).reshape(cnt, 3)
                return keys, vals
            sizes_b = np.diff(np.append(ob, img.size)); sizes_g = np.diff(np.append(og, good.size))
            diffb = np.nonzero(sizes_b != sizes_g)[0]
            print("buckets with different size:", diffb.size, diffb[:10])
            shown = 0
            for b in range(nb):
                if shown >= 6: break
                kb_, vb = bucket(img, ob, b); kg, vg = bucket(good, og, b)
                if kb_.size != kg.size or not np.array_equal(kb_, kg) or not np.array_equal(vb, vg):
                    shown += 1
                    print("bucket", b, "bad n", kb_.size, "good n", kg.size)
                    dg = dict(zip(kg.tolist(), vg[:, 0].tolist())); db = dict(zip(kb_.tolist(), vb[:, 0].tolist()))
                    for k_ in sorted(set(dg) | set(db)):
                        if dg.get(k_) != db.get(k_): print("   key %016x good %s bad %s" % (k_, dg.get(k_), db.get(k_)))
            sys.exit(1)
print("no defect in", iters, "iterations")
