# the k-mer partition (build_mode 2) with 128 / 160 / 192 / 256 partition blocks: how its two passes scale with the CUs they run on
for pb in 128 160 192 256; do
  python bench.py --steps 3 --warmup 1 --no-cpu --no-h2d --build-mode 2 --tune partition_blocks=$pb > gpurun_out/pb_$pb.json 2> gpurun_out/pb_$pb.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/pb_$pb.json"))
print($pb, "ms/step %.2f" % d["ms_per_step"], [(k["name"][:28], round(k["ms_per_step"],2)) for k in d["roofline"]["kernels"]])
PY
done
