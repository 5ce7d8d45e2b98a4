mkdir -p gpurun_out/j14
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
DANHIP_WGRAD_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/j14/trace -o t -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-serialized-roofline --no-eval > gpurun_out/j14/bench.log 2>&1
f=$(find gpurun_out/j14/trace -name "*kernel_trace.csv" | head -n 1)
python3 tools/step_breakdown.py $f > gpurun_out/j14/step_breakdown.txt 2>&1
rm -rf gpurun_out/j14/trace

(head -n 3 gpurun_out/j14/step_breakdown.txt; tail -n 18 gpurun_out/j14/step_breakdown.txt) | cut -c1-150

