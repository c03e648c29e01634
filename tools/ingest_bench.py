"""development tool: throughput of kmr_ingest_fastq (FASTQ text already in HBM) and parity of the result at scale"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, ctypes as C
import kmernator_amd as ka
from helpers import synth_reads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rb = synth_reads(n, read_len=150, seed=5, quality="noisy", n_rate=0.001)
L = 150
names = np.char.add(np.char.add("@r", np.arange(n).astype(str)), " 1:N:0:ACGT").astype("S")
rows = [b"%s\n%s\n+\n%s\n" % (names[i], rb.bases[i * L:(i + 1) * L].tobytes(), rb.quals[i * L:(i + 1) * L].tobytes()) for i in range(n)]
text = b"".join(rows)
sp = ka.KmerSpectrum(ka.default_config(31, estimated_raw_kmers=n * 120, device=0))
dtext = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
lib = sp.lib
for rep in range(3):
    r = C.c_void_p(); t0 = time.time()
    rc = lib.kmr_ingest_fastq_dev(sp.h, dtext.data_ptr(), len(text), 0, 1, C.byref(r)); assert rc == 0
    dt = time.time() - t0
    nn, tot = C.c_uint64(), C.c_uint64(); lib.kmr_reads_info(r, C.byref(nn), C.byref(tot), None, None)
    print("rep %d: %d reads, %.1f MB of FASTQ in %.2f ms -> %.1f GB/s" % (rep, nn.value, len(text) / 1e6, dt * 1e3, len(text) / dt / 1e9), flush=True)
    if rep < 2: lib.kmr_reads_free(r)
b = np.zeros(tot.value, np.uint8); q = np.zeros(tot.value, np.uint8)
lib.kmr_reads_copy(r, b.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p), None, None, None)
print("bases identical:", np.array_equal(b, rb.bases), "quals identical:", np.array_equal(q, rb.quals))
# the rest of the FilterReads flow on the same device-resident reads: spectrum, then scoreAndTrimReads
t0 = time.time(); rc = lib.kmr_add_read_batch(sp.h, r, 0); assert rc == 0; sp.finalize(2); torch.cuda.synchronize(); t1 = time.time()
to = np.zeros(n, np.uint32); tl = np.zeros(n, np.uint32); sc = np.zeros(n, np.float32); wt = np.zeros(n, np.uint8)
for rep in range(2):
    t2 = time.time()
    rc = lib.kmr_score_read_batch(sp.h, r, 2.0, 1, to.ctypes.data_as(C.POINTER(C.c_uint32)), tl.ctypes.data_as(C.POINTER(C.c_uint32)), sc.ctypes.data_as(C.POINTER(C.c_float)), wt.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert rc == 0; t3 = time.time()
print("build %.1f ms; scoreAndTrimReads of %d reads (%d k-mer lookups) %.1f ms -> %.2f G lookups/s; trimmed %d, median of medians %.0f" % (
    (t1 - t0) * 1e3, n, n * 120, (t3 - t2) * 1e3, n * 120 / (t3 - t2) / 1e9, int(wt.sum()), float(np.median(sc))))
# the artifact filter in front of the build (FilterKnownOddities): the reference's table, default settings
table = open(os.path.join(ROOT, "tests", "golden", "artifact_sequences.fa"), "rb").read()
rs = ka.ReadSet._adopt(sp, text, r)
for kw in (dict(), dict(edit_distance=3)):
    t0 = time.time(); f = ka.FilterKnownOddities(sp, table, **kw); t1 = time.time()
    for rep in range(2):
        t2 = time.time(); res, _ = f.applyFilter(rs, want_reads=False); t3 = time.time()
    t4 = time.time(); res, frs = f.applyFilter(rs); t5 = time.time()
    print("artifact filter %s: %d keys built in %.0f ms, %d edits at query time; screen of %d reads %.1f ms (%.1f GB/s of bases+quals), with the filtered batch %.1f ms; trimmed %d, discarded %d, remnants %d" % (
        kw, f.n_filter_kmers, (t1 - t0) * 1e3, f.remaining_edits, n, (t3 - t2) * 1e3, 2 * tot.value / (t3 - t2) / 1e9, (t5 - t4) * 1e3,
        int((res["action"] == 1).sum()), int((res["action"] == 2).sum()), int((res["remnant_len"] > 0).sum())), flush=True)
    frs.close(); f.close()
rs.r = None
