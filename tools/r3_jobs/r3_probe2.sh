#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3p2
mkdir -p $OUT
cd $ROOT
timeout 900 python3 -m pytest tests/test_conv_gpu.py -x -q -m gpu --durations=8 > $OUT/test_conv.txt 2>&1
tail -15 $OUT/test_conv.txt
python3 tools/bench_conv.py --set s3fd --batch 2 --which fwd,dgrad,wgrad > $OUT/conv_b2.txt 2>&1
python3 tools/bench_conv.py --set tail --batch 2 --which fwd,dgrad,wgrad >> $OUT/conv_b2.txt 2>&1
python3 tools/bench_conv.py --set s3fd --which wgrad > $OUT/wgrad_b16_slab.txt 2>&1
DANHIP_WGRAD_SLAB=0 python3 tools/bench_conv.py --set s3fd --which wgrad > $OUT/wgrad_b16_atomic.txt 2>&1
python3 tools/bench_conv.py --set tail --which fwd,dgrad,wgrad > $OUT/tail_b16.txt 2>&1
DANHIP_SPLITK=0 python3 tools/bench_conv.py --set tail --which fwd,dgrad > $OUT/tail_b16_nosplit.txt 2>&1
for b in 2 4 16; do
  python3 bench.py --batch-per-gpu $b --steps 20 --no-cpu-baseline --no-eval --no-serialized-roofline 2>/dev/null | tail -1 | cut -c1-420 >> $OUT/bench_lines.txt
done
DANHIP_WGRAD_STREAM=0 python3 bench.py --steps 20 --no-cpu-baseline --no-eval --no-serialized-roofline 2>/dev/null | tail -1 | cut -c1-420 >> $OUT/bench_lines.txt
