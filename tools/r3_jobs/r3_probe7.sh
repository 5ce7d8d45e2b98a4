#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3p7
mkdir -p $OUT
cd $ROOT
timeout 600 python3 -m pytest tests/test_conv_gpu.py -q -m gpu -x -k "halo2" > $OUT/tests.txt 2>&1
tail -15 $OUT/tests.txt
for i in 1 2; do
timeout 300 python3 tools/bench_conv.py --set s3fd --which fwd,dgrad_nomask,dgrad_bits --only conv2_2,conv3_1,conv3_2 --check >> $OUT/halo2.txt 2>&1
DANHIP_HALO2=0 timeout 300 python3 tools/bench_conv.py --set s3fd --which fwd,dgrad_nomask,dgrad_bits --only conv2_2,conv3_1,conv3_2 --check >> $OUT/halo1.txt 2>&1
done
paste $OUT/halo2.txt $OUT/halo1.txt | cut -c1-75,140-215
