#!/bin/bash
# One rocprofv3 --pmc pass over a command, per-kernel averages printed for kernels matching a pattern.
# Keep to the SQ_* / GRBM_* / FETCH_SIZE / WRITE_SIZE sets of tools/pmc_conv.sh: a mixed TA_* / TCP_* set made rocprofv3 abort and then hang in its
# finaliser for the rest of the job (hence the timeout).
# usage: tools/pmc_one.sh "<counters>" <kernel-substring> -- python3 script.py args...
set -u
CTRS=$1; PAT=$2; shift 3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_one
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
timeout 150 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -- "$@" > $OUT/run.log 2>&1
python3 - "$OUT" "$PAT" <<'PY'
import csv, collections, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k)
    for c, vals in sorted(v.items()):
        print("   %-36s %16.0f (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
PY
