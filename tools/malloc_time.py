"""development tool: what hipMalloc / hipFree of tens of GB cost on the box (normally 0.3 ms whatever the size; seconds when the driver
still has freed memory to clear -- the pattern a pool that grows by reallocation produces, HISTORY.md section 6)"""
import torch, time, ctypes
hip = ctypes.CDLL("libamdhip64.so")
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for gb in (8, 16, 24, 32, 40, 64, 90, 24, 40):
    p = ctypes.c_void_p()
    t0 = time.time(); rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(gb << 30)); t1 = time.time()
    rc2 = hip.hipFree(p); t2 = time.time()
    print("hipMalloc %3d GiB: %8.1f ms (rc %d), hipFree %8.1f ms" % (gb, (t1 - t0) * 1e3, rc, (t2 - t1) * 1e3), flush=True)
