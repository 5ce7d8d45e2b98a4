#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3p21
mkdir -p $OUT
cd $ROOT
H2_TRACE_DUMP=1 timeout 60 tools/halo1_trace fwd > $OUT/halo1_fwd.txt 2>&1
H2_TRACE_DUMP=1 timeout 60 tools/halo1_trace dgrad > $OUT/halo1_dgrad.txt 2>&1
H2_TRACE_DUMP=1 timeout 60 tools/halo1_trace fwd 16 320 320 128 128 > $OUT/halo1_fwd_conv2_2.txt 2>&1
H2_TRACE_DUMP=1 DANHIP_HALO_B2=1 timeout 60 tools/halo1_trace fwd > $OUT/halo1_fwd_b2.txt 2>&1
grep -v "^ *[0-9]* tap\|first epilogue" $OUT/halo1_fwd.txt $OUT/halo1_dgrad.txt $OUT/halo1_fwd_conv2_2.txt $OUT/halo1_fwd_b2.txt
grep "^ *[0-9]* tap" $OUT/halo1_fwd.txt | sed -n 30,76p
