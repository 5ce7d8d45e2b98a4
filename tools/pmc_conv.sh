#!/bin/bash
# PMC passes over one conv shape (rocprofv3 counters only: no tracing domains besides kernel-trace).
# usage: tools/pmc_conv.sh <shape-name> <which> [set]
# writes gpurun_out/pmc_<shape>_<which>/passN/*.csv
set -u
SHAPE=${1:-conv3_2}; WHICH=${2:-fwd}; SET=${3:-s3fd}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_${SHAPE}_${WHICH}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INSTS_SALU"
P3="GRBM_GUI_ACTIVE GRBM_COUNT"
P4="FETCH_SIZE"
P5="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -- python3 $ROOT/tools/bench_conv.py --set $SET --only $SHAPE --which $WHICH --iters 3 > $OUT/pass$i.log 2>&1
done
python3 $ROOT/tools/pmc_summarize.py $OUT
