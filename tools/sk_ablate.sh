#!/bin/bash
# development aid (needs a -DKMR_DEBUG_HOOKS build): C2 bench of build_mode 3 with parts of the two kernels switched off
for e in 0 1 2; do for c in 0 1 2 4 6; do
  KMR_SK_EXTRACT_DBG=$e KMR_SK_COUNT_DBG=$c python bench.py --steps 3 --warmup 1 --no-cpu --build-mode 3 --no-check > gpurun_out/abl_$e$c.json 2> gpurun_out/abl_$e$c.err || { tail -2 gpurun_out/abl_$e$c.err; continue; }
  echo "extract_dbg=$e count_dbg=$c: $(python tools/kern.py gpurun_out/abl_$e$c.json | tr '\n' ' ')"
  [ $e != 0 ] && break
done; done
