#!/bin/bash
# kernel timeline of a few steady-state bench steps (GPU box).  usage: bash tools/timeline.sh [bench args...]
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/timeline; rm -rf $O; mkdir -p $O; cd $R
export ASVGP_BENCH_NOPROF=1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --phase-events 0 "$@" > $O/bench.json 2> $O/err.txt || exit 1
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$O/t/*/*_kernel_trace.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if "asvgp" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-40:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print("%9.1f -> %9.1f  (%6.1f us)  q%s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0].replace("void asvgp::", "")[:40]))
PY
