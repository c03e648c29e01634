#!/bin/bash
for e in 0 4 1; do
  KMR_SK_EXTRACT_DBG=$e python bench.py --steps 3 --warmup 1 --no-cpu --build-mode 3 --no-check > gpurun_out/abl2_$e.json 2> gpurun_out/abl2_$e.err || { tail -2 gpurun_out/abl2_$e.err; continue; }
  echo "extract_dbg=$e: $(python tools/kern.py gpurun_out/abl2_$e.json | head -2 | tr '\n' ' ')"
done
