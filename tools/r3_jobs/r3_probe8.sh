#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3p8
mkdir -p $OUT
cd $ROOT
DANHIP_HALO2=1 bash tools/pmc_conv.sh conv2_2 fwd > $OUT/pmc_halo2_conv2_2_fwd.txt 2>&1
DANHIP_HALO2=0 bash tools/pmc_conv.sh conv2_2 fwd > $OUT/pmc_halo1_conv2_2_fwd.txt 2>&1
paste $OUT/pmc_halo2_conv2_2_fwd.txt $OUT/pmc_halo1_conv2_2_fwd.txt | cut -c1-70,118-180 | head -30
