#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3p20
mkdir -p $OUT
cd $ROOT
python3 bench.py --no-cpu-baseline > $OUT/bench_b16.json 2> $OUT/bench.err
tail -1 $OUT/bench_b16.json | cut -c1-400
DANHIP_HALO_B2=1 DANHIP_WGRAD_B2=1 python3 bench.py --no-cpu-baseline --no-serialized-roofline --no-eval 2>/dev/null | tail -1 | cut -c1-200
for m in pb dan dan_deform; do
  python3 bench.py --model $m --graph --no-cpu-baseline --no-serialized-roofline --no-eval 2>/dev/null | tail -1 | cut -c1-200
done
for b in 2 4 8; do
  python3 bench.py --batch-per-gpu $b --steps 20 --no-cpu-baseline --no-serialized-roofline --no-eval 2>/dev/null | tail -1 | cut -c1-200
done
timeout 1500 python3 -m pytest tests -q -m gpu -x > $OUT/tests.txt 2>&1
tail -5 $OUT/tests.txt
