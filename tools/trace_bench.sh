#!/bin/bash
# kernel-trace summary of the bench (GPU box): average durations of the step's kernels
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_bench; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/err.txt || exit 1
python3 - <<PY
import csv, glob, json
f = sorted(glob.glob("$O/t/*/*_kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if "asvgp" in r["Name"]:
        print("%-46s calls %4s  avg %8.1f us" % (r["Name"].split("(")[0].replace("void asvgp::", "")[:46], r["Calls"], float(r["AverageNs"]) / 1e3))
print("ms_per_step", json.loads(open("$O/bench.json").read().strip().splitlines()[-1])["ms_per_step"])
PY
