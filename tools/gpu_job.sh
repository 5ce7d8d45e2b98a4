mkdir -p gpurun_out/j4
python -m pytest tests/test_context_block_gpu.py -q > gpurun_out/j4/pytest1.log 2>&1; echo "pytest1 rc=$?"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in dan; do
  DANHIP_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d gpurun_out/j4/serial_$m -o s -- python3 bench.py --eager --model $m --steps 4 --warmup 2 --no-cpu-baseline --no-serialized-roofline --no-eval > gpurun_out/j4/serial_$m.log 2>&1
  python3 tools/prof_db.py gpurun_out/j4/serial_$m/s_results.db 6 70 > gpurun_out/j4/${m}_b16_serialized_kernels.txt
  rm -rf gpurun_out/j4/serial_$m
done
for s in 0.5 2.0; do for f in 1 2 3; do OFF_SCALE=$s DANHIP_DEFORM_BWD_FORM=$f python tools/bench_deform_bwd.py 2>/dev/null | tail -n 1 >> gpurun_out/j4/deform_bwd_forms.txt; done; done
python tools/bench_deform_fwd.py 160 2.0 > gpurun_out/j4/deform_fwd.txt 2>&1
python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | tail -n 1 > gpurun_out/j4/bench_sfd.json
python bench.py --model dan_deform --graph --no-cpu-baseline --no-serialized-roofline --no-eval 2>/dev/null | tail -n 1 > gpurun_out/j4/bench_dan_deform_graph.json
python bench.py --model pb --no-cpu-baseline --no-serialized-roofline --no-eval 2>/dev/null | tail -n 1 > gpurun_out/j4/bench_pb_eager.json
tail -n 6 gpurun_out/j4/pytest1.log
head -n 75 gpurun_out/j4/dan_b16_serialized_kernels.txt | cut -c1-175
cat gpurun_out/j4/deform_bwd_forms.txt gpurun_out/j4/deform_fwd.txt
cut -c1-160 gpurun_out/j4/bench_*.json
