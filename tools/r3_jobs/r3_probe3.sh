#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3p3
mkdir -p $OUT
cd $ROOT
timeout 1500 python3 -m pytest tests -q -m gpu --durations=45 > $OUT/test_all.txt 2>&1
tail -70 $OUT/test_all.txt
for b in 2 4 8 16; do
  python3 bench.py --batch-per-gpu $b --steps 20 --no-cpu-baseline --no-eval --no-serialized-roofline 2>/dev/null | tail -1 | cut -c1-520 >> $OUT/bench_lines.txt
done
python3 bench.py --batch-per-gpu 2 --eager --steps 20 --no-cpu-baseline --no-eval --no-serialized-roofline 2>/dev/null | tail -1 | cut -c1-520 >> $OUT/bench_lines.txt
cat $OUT/bench_lines.txt | cut -c60-200
