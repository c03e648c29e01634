#!/bin/bash
# usage: tools/evidence.sh <tag> [commit] -- the round's evidence in one call: bench lines (C2 default incl. CPU baseline and both PCIe legs, noisy),
# rocprofv3 kernel stats of C2 / noisy C2 / C4, HBM traffic counters (flat and noisy), SQ / LDS / L2 counters; everything lands in
# gpurun_out/ under <tag>.  A long stretch without output makes the pool think the call hangs: every step prints a line.
tag=$1; commit=${2:-?}
R=$GRAFT_REPO_ROOT
cd $R
echo "== C2 bench"; timeout -k 10 500 python3 bench.py --steps 20 --warmup 3 > gpurun_out/${tag}_c2_bench.json 2> gpurun_out/${tag}_c2_bench.err || tail -3 gpurun_out/${tag}_c2_bench.err
python3 tools/kern.py gpurun_out/${tag}_c2_bench.json
echo "== noisy bench"; timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu --quality noisy > gpurun_out/${tag}_noisy_bench.json 2> gpurun_out/${tag}_noisy_bench.err || tail -3 gpurun_out/${tag}_noisy_bench.err
python3 tools/kern.py gpurun_out/${tag}_noisy_bench.json
echo "== kernel stats C2"; tools/prof.sh ${tag} | tee gpurun_out/${tag}_prof.txt
echo "== kernel stats noisy"; tools/prof.sh ${tag}n --quality noisy | tee gpurun_out/${tag}n_prof.txt
echo "== kernel stats C4"; tools/prof_any.sh ${tag}c4 $R/tools/c4_check.py 50000000 51 150 3 | tee gpurun_out/${tag}c4_prof.txt; tail -4 gpurun_out/prof_${tag}c4.log
echo "== HBM counters flat"; tools/pmc.sh ${tag} > gpurun_out/${tag}_pmc.txt 2>&1; python3 tools/pmc_summary.py ${tag} flat $commit > gpurun_out/${tag}_pmc_traffic_flat.json 2>> gpurun_out/${tag}_pmc.txt; tail -2 gpurun_out/${tag}_pmc.txt
echo "== HBM counters noisy"; tools/pmc.sh ${tag}n --quality noisy > gpurun_out/${tag}n_pmc.txt 2>&1; python3 tools/pmc_summary.py ${tag}n noisy $commit > gpurun_out/${tag}_pmc_traffic_noisy.json 2>> gpurun_out/${tag}n_pmc.txt; tail -2 gpurun_out/${tag}n_pmc.txt
echo "== SQ counters"; tools/sq_counters.sh ${tag} | tail -40 | tee gpurun_out/${tag}_sq.txt
