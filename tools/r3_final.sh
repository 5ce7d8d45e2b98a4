#!/bin/bash
# End-of-round evidence job (one gpurun call).  Everything lands in gpurun_out/r3f/; copy what is kept into profiles/r3/.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash tools/collect_round_profiles.sh r3f > gpurun_out/r3f_collect.log 2>&1
OUT=$ROOT/gpurun_out/r3f
# step-phase traces of the 3x3 kernels (conv3_2 / conv2_2 shapes, batch 16)
H2_TRACE_DUMP=1 timeout 60 tools/halo1_trace fwd > $OUT/trace_conv_halo_fwd_conv3_2.txt 2>&1
H2_TRACE_DUMP=1 timeout 60 tools/halo1_trace dgrad > $OUT/trace_conv_halo_dgrad_bits_conv3_2.txt 2>&1
H2_TRACE_DUMP=1 timeout 60 tools/halo1_trace fwd 16 320 320 128 128 > $OUT/trace_conv_halo_fwd_conv2_2.txt 2>&1
H2_TRACE_DUMP=1 DANHIP_HALO_B2=1 timeout 60 tools/halo1_trace fwd > $OUT/trace_conv_halo_fwd_conv3_2_two_barriers.txt 2>&1
H2_TRACE_DUMP=1 timeout 60 tools/halo2_trace fwd > $OUT/trace_conv_halo2_fwd_conv3_2.txt 2>&1
H2_TRACE_DUMP=1 timeout 60 tools/halo2_trace dgrad > $OUT/trace_conv_halo2_dgrad_bits_conv3_2.txt 2>&1
# same-box A/B of the round's kernel switches (per-layer microbench, batch 16)
{
for i in 1 2; do
  for v in "DANHIP_HALO_B2=0 DANHIP_HALO_GENERAL_EPILOGUE=0 DANHIP_HALO2=0" "DANHIP_HALO_B2=1 DANHIP_HALO_GENERAL_EPILOGUE=0 DANHIP_HALO2=0" "DANHIP_HALO_B2=0 DANHIP_HALO_GENERAL_EPILOGUE=1 DANHIP_HALO2=0" "DANHIP_HALO_B2=0 DANHIP_HALO_GENERAL_EPILOGUE=0 DANHIP_HALO2=1"; do
    echo "== $v"
    env $v timeout 300 python3 tools/bench_conv.py --set s3fd --which fwd,dgrad_nomask,dgrad_bits --only conv2_2,conv3_1,conv3_2,conv4_1,conv4_2 2>&1 | grep -v amdgpu | cut -c1-100
  done
  for v in "DANHIP_WGRAD_B2=0" "DANHIP_WGRAD_B2=1"; do
    echo "== $v"
    env $v timeout 300 python3 tools/bench_conv.py --set s3fd --which wgrad 2>&1 | grep -v amdgpu | cut -c1-100
  done
done
} > $OUT/ab_kernel_switches.txt 2>&1
{
for i in 1 2; do
  echo "== defaults"; python3 bench.py --no-cpu-baseline --no-serialized-roofline --no-eval 2>/dev/null | tail -1
  echo "== DANHIP_HALO_B2=1 DANHIP_WGRAD_B2=1 DANHIP_HALO_GENERAL_EPILOGUE=1 (the round-2 forms)"; DANHIP_HALO_B2=1 DANHIP_WGRAD_B2=1 DANHIP_HALO_GENERAL_EPILOGUE=1 python3 bench.py --no-cpu-baseline --no-serialized-roofline --no-eval 2>/dev/null | tail -1
done
} > $OUT/ab_bench_lines.txt 2>&1
for w in fwd dgrad_bits wgrad; do
  bash tools/pmc_conv.sh conv3_2 $w > $OUT/pmc_conv3_2_$w.txt 2>&1
done
ls -la $OUT
