#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3g
cd $ROOT; mkdir -p $OUT
: > $OUT/s3fd_small_batch_lines.jsonl
for b in 2 4 8; do
  python3 bench.py --batch-per-gpu $b --steps 20 --no-cpu-baseline --no-serialized-roofline --no-eval 2>/dev/null | tail -1 >> $OUT/s3fd_small_batch_lines.jsonl
done
: > $OUT/size1024_lines.jsonl
python3 bench.py --model dan --size 1024 --batch-per-gpu 8 --steps 5 --graph --no-cpu-baseline --no-serialized-roofline --no-eval 2>/dev/null | tail -1 >> $OUT/size1024_lines.jsonl
DANHIP_DTYPE=fp16 python3 bench.py --model dan_deform --size 1024 --batch-per-gpu 8 --steps 5 --graph --no-cpu-baseline --no-serialized-roofline --no-eval 2>/dev/null | tail -1 >> $OUT/size1024_lines.jsonl
python3 - <<'PY'
import json
for f in ("s3fd_small_batch_lines.jsonl","size1024_lines.jsonl"):
    for l in open("gpurun_out/r3g/"+f):
        d=json.loads(l); print(f[:12], d["config"].get("global_batch"), d["dtype"], d["value"], d["ms_per_step"])
PY
