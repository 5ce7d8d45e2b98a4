#!/bin/bash
# HBM traffic of the bench workload per kernel: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short
# bench.py run, summarised by tools/pmc_traffic.py into profiles/<round>/pmc_bench_traffic.json
# (bytes per launch = 2*FETCH_SIZE*1024 [gfx950 correction, MI355X_MICROARCH.md §HBM] + WRITE_SIZE*1024).
set -u
ROUND=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_bench
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 --repeats 1 --strong-global-batch 0 --strong16-global-batch 0 --no-cpu-baseline --no-eval > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --steps 2 --warmup 1 --repeats 1 --strong-global-batch 0 --strong16-global-batch 0 --no-cpu-baseline --no-eval > $OUT/write.log 2>&1
python3 $ROOT/tools/pmc_traffic.py $OUT $ROOT/gpurun_out/pmc_bench_traffic.json
# copy gpurun_out/pmc_bench_traffic.json to profiles/<round>/ (bench.py reads profiles/r3/pmc_bench_traffic.json; stamped with the kernel-source hash)
