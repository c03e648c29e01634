#!/bin/bash
# usage: tools/evidence.sh <tag> -- the round's evidence in one call: bench lines (C2 default incl. CPU baseline and both PCIe legs, noisy), rocprofv3
# kernel stats, HBM traffic counters, SQ / LDS / L2 counters; everything lands in gpurun_out/ under <tag>
tag=$1
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 500 python3 bench.py > gpurun_out/${tag}_c2_bench.json 2> gpurun_out/${tag}_c2_bench.err || tail -3 gpurun_out/${tag}_c2_bench.err
python3 tools/kern.py gpurun_out/${tag}_c2_bench.json
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu --quality noisy > gpurun_out/${tag}_noisy_bench.json 2> gpurun_out/${tag}_noisy_bench.err || tail -3 gpurun_out/${tag}_noisy_bench.err
python3 tools/kern.py gpurun_out/${tag}_noisy_bench.json
tools/prof.sh ${tag} | tee gpurun_out/${tag}_prof.txt
tools/pmc.sh ${tag} > gpurun_out/${tag}_pmc.txt 2>&1; python3 tools/pmc_summary.py ${tag} > gpurun_out/${tag}_pmc_traffic.json 2>> gpurun_out/${tag}_pmc.txt; tail -2 gpurun_out/${tag}_pmc.txt
tools/sq_counters.sh ${tag} | tee gpurun_out/${tag}_sq.txt
