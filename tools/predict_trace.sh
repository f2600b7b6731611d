#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/predict_trace; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 tools/predict_probe.py > $O/log.txt 2>&1 || exit 1
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$O/t/*/*_kernel_trace.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if "predict_kernel" in r["Kernel_Name"]]
for r in rows[-14:]:
    print(r["Grid_Size_X"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
