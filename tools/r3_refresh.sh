#!/bin/bash
# Last evidence step of the round: the PMC traffic file of the FINAL sources, then the bench line that reads it.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3g
mkdir -p $OUT
cd $ROOT
bash tools/pmc_bench.sh r3 > $OUT/pmc.log 2>&1
cp gpurun_out/pmc_bench_traffic.json $OUT/
cp gpurun_out/pmc_bench_traffic.json profiles/r3/pmc_bench_traffic.json      # (on the box: so that the bench line below prints `traffic`)
python3 bench.py > $OUT/s3fd_b16_bench_line.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o s3fd -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-serialized-roofline --no-eval > $OUT/s3fd_b16_bench_line_under_rocprof.json 2> $OUT/prof.err
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/s3fd_b16_kernel_stats.csv \;
rm -rf $OUT/prof
: > $OUT/models_bench_lines.jsonl
for m in pb dan dan_deform; do
  python3 bench.py --model $m --no-cpu-baseline --no-serialized-roofline 2>/dev/null | tail -1 >> $OUT/models_bench_lines.jsonl
  python3 bench.py --model $m --graph --no-cpu-baseline --no-serialized-roofline --no-eval 2>/dev/null | tail -1 >> $OUT/models_bench_lines.jsonl
done
DANHIP_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d $OUT/serial_sfd -o s -- python3 bench.py --eager --model sfd --steps 4 --warmup 2 --no-cpu-baseline --no-serialized-roofline --no-eval > $OUT/serial_sfd.log 2>&1
python3 tools/prof_db.py $OUT/serial_sfd/s_results.db 6 60 > $OUT/sfd_b16_serialized_kernels.txt
rm -rf $OUT/serial_sfd
tail -1 $OUT/s3fd_b16_bench_line.json | cut -c1-200
ls $OUT
