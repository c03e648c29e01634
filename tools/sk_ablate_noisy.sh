for spec in "0 0" "1 0" "2 0" "4 0"; do set -- $spec
  KMR_SK_EXTRACT_DBG=$1 KMR_SK_COUNT_DBG=$2 python bench.py --steps 3 --warmup 1 --no-cpu --no-h2d --quality noisy --build-mode 3 --no-check > gpurun_out/ablq_$1$2.json 2> gpurun_out/ablq_$1$2.err || { tail -2 gpurun_out/ablq_$1$2.err; continue; }
  echo "extract_dbg=$1 count_dbg=$2: $(python tools/kern.py gpurun_out/ablq_$1$2.json | tr '\n' ' ')"
done
