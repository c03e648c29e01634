"""development tool: throughput of kmr_ingest_fastq (FASTQ text already in HBM) and parity of the result at scale"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
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
