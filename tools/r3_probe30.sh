#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for f in 0 1 2; do
DANHIP_DEFORM_BWD_FORM=$f timeout 600 python3 -m pytest tests/test_deform_gpu.py -q -m gpu -x 2>&1 | tail -1
done
for s in 0.3 0.5 1.0 2.0; do for f in 0 1 2; do
  if [ "$f" = "1" ] && [ "$s" = "2.0" ]; then continue; fi
  DANHIP_DEFORM_BWD_FORM=$f OFF_SCALE=$s timeout 200 python3 tools/bench_deform_bwd.py 2>&1 | grep -v amdgpu | tail -1
done; done
