#!/bin/bash
# development aid (needs a -DKMR_DEBUG_HOOKS build: make -B -C kmernator_amd/csrc HIPFLAGS="... -DKMR_DEBUG_HOOKS"): C2 bench of
# build_mode 3 with parts of the two kernels switched off (the results are void)
# extract: 1 = no list appends, 2 = no gather, 4 = bookings but no record stores; count: 1 = no table, 2 = no emit, 4 = no insert loop
for spec in "0 0" "0 1" "0 2" "0 4" "0 6" "1 0" "2 0" "4 0"; do set -- $spec
  KMR_SK_EXTRACT_DBG=$1 KMR_SK_COUNT_DBG=$2 python bench.py --steps 3 --warmup 1 --no-cpu --build-mode 3 --no-check > gpurun_out/abl_$1$2.json 2> gpurun_out/abl_$1$2.err || { tail -2 gpurun_out/abl_$1$2.err; continue; }
  echo "extract_dbg=$1 count_dbg=$2: $(python tools/kern.py gpurun_out/abl_$1$2.json | tr '\n' ' ')"
done
