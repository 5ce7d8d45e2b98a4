#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3p10
mkdir -p $OUT
cd $ROOT
timeout 600 python3 -m pytest tests/test_conv_gpu.py -q -m gpu -x -k "halo2" > $OUT/tests.txt 2>&1
tail -5 $OUT/tests.txt
for i in 1 2; do
DANHIP_HALO2=1 timeout 300 python3 tools/bench_conv.py --set s3fd --which fwd,dgrad_nomask,dgrad_bits --only conv2_2,conv3_1,conv3_2 --check 2>&1 | grep -v amdgpu >> $OUT/halo2.txt
DANHIP_HALO2=0 timeout 300 python3 tools/bench_conv.py --set s3fd --which fwd,dgrad_nomask,dgrad_bits --only conv2_2,conv3_1,conv3_2 2>&1 | grep -v amdgpu >> $OUT/halo1.txt
done
paste $OUT/halo2.txt $OUT/halo1.txt | cut -c1-62,100-112,125-170
for ab in 1 2 4 8; do
  echo "== halo2 ablate=$ab" >> $OUT/ablate.txt
  DANHIP_HALO2=1 DANHIP_HALO2_ABLATE=$ab timeout 200 python3 tools/bench_conv.py --set s3fd --which fwd --only conv2_2,conv3_2 2>&1 | grep "fwd " | cut -c1-60 >> $OUT/ablate.txt
done
cat $OUT/ablate.txt
