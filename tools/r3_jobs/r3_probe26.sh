#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3p26
mkdir -p $OUT
cd $ROOT
timeout 900 python3 -m pytest tests/test_conv_gpu.py tests/test_models_gpu.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do
for b in 1 0; do
echo "== general_epilogue=$b"
DANHIP_HALO_GENERAL_EPILOGUE=$b timeout 300 python3 tools/bench_conv.py --set s3fd --which dgrad_acc --only conv2_2,conv3_1,conv3_2,conv4_1,conv4_2 --check 2>&1 | grep -v amdgpu | cut -c1-110
done
done > $OUT/ab.txt 2>&1
grep "==\|TOTAL\|relerr" $OUT/ab.txt | head -30
for m in pb dan; do
  python3 bench.py --model $m --no-cpu-baseline --no-serialized-roofline --no-eval 2>/dev/null | tail -1 | cut -c1-160
  DANHIP_HALO_GENERAL_EPILOGUE=1 python3 bench.py --model $m --no-cpu-baseline --no-serialized-roofline --no-eval 2>/dev/null | tail -1 | cut -c1-160
done
