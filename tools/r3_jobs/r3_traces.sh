#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3f
mkdir -p $OUT
cd $ROOT
H2_TRACE_DUMP=1 timeout 60 tools/halo1_trace fwd > $OUT/trace_conv_halo_fwd_conv3_2.txt 2>&1
H2_TRACE_DUMP=1 timeout 60 tools/halo1_trace dgrad > $OUT/trace_conv_halo_dgrad_bits_conv3_2.txt 2>&1
H2_TRACE_DUMP=1 timeout 60 tools/halo1_trace fwd 16 320 320 128 128 > $OUT/trace_conv_halo_fwd_conv2_2.txt 2>&1
H2_TRACE_DUMP=1 DANHIP_HALO_B2=1 timeout 60 tools/halo1_trace fwd > $OUT/trace_conv_halo_fwd_conv3_2_two_barriers.txt 2>&1
H2_TRACE_DUMP=1 timeout 60 tools/halo2_trace fwd > $OUT/trace_conv_halo2_fwd_conv3_2.txt 2>&1
H2_TRACE_DUMP=1 timeout 60 tools/halo2_trace dgrad > $OUT/trace_conv_halo2_dgrad_bits_conv3_2.txt 2>&1
grep -v "^ *[0-9]* tap\|first epilogue" $OUT/trace_conv_halo_fwd_conv3_2.txt $OUT/trace_conv_halo_dgrad_bits_conv3_2.txt $OUT/trace_conv_halo_fwd_conv2_2.txt
grep "^ *[0-9]* tap" $OUT/trace_conv_halo_fwd_conv3_2.txt | sed -n 34,48p
python3 bench.py > $OUT/s3fd_b16_bench_line.json 2> $OUT/bench.err
tail -1 $OUT/s3fd_b16_bench_line.json | cut -c1-200
