#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3p14
mkdir -p $OUT
cd $ROOT
timeout 600 python3 -m pytest tests/test_conv_gpu.py -q -m gpu -x -k "halo2" > $OUT/tests.txt 2>&1
tail -5 $OUT/tests.txt
H2_TRACE_DUMP=1 timeout 60 tools/halo2_trace fwd > $OUT/trace_fwd.txt 2>&1
H2_TRACE_DUMP=1 timeout 60 tools/halo2_trace dgrad > $OUT/trace_dgrad.txt 2>&1
H2_TRACE_DUMP=1 timeout 60 tools/halo2_trace fwd 16 320 320 128 128 > $OUT/trace_fwd_conv2_2.txt 2>&1
grep -v "^ *[0-9]* tap" $OUT/trace_fwd.txt $OUT/trace_dgrad.txt $OUT/trace_fwd_conv2_2.txt
grep "^ *[0-9]* tap" $OUT/trace_fwd.txt | sed -n 64,76p
for i in 1 2; do
DANHIP_HALO2=1 timeout 300 python3 tools/bench_conv.py --set s3fd --which fwd,dgrad_nomask,dgrad_bits --only conv2_2,conv3_1,conv3_2 --check 2>&1 | grep -v amdgpu >> $OUT/halo2.txt
DANHIP_HALO2=0 timeout 300 python3 tools/bench_conv.py --set s3fd --which fwd,dgrad_nomask,dgrad_bits --only conv2_2,conv3_1,conv3_2 2>&1 | grep -v amdgpu >> $OUT/halo1.txt
done
cut -c1-50 $OUT/halo2.txt | grep -v relerr | head -12; echo; cut -c1-50 $OUT/halo1.txt | head -12
grep relerr $OUT/halo2.txt | head -3
