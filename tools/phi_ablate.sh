#!/bin/bash
# Phi kernel time with parts switched off (GPU box): ASVGP_PHI_ABLATE = 0 full, 1 no LDS atomics, 2 loads only
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for ab in 0 1 3 4; do
  O=$R/gpurun_out/abl$ab; rm -rf $O; mkdir -p $O
  ASVGP_PHI_ABLATE=$ab timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 tools/phi_pmc.py > $O/log.txt 2>&1 || exit 1
  python3 - <<PY
import csv, glob
f = sorted(glob.glob("$O/t/*/*_kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if "phi_moment_kernel" in r["Name"] or "phi_accumulate" in r["Name"]:
        print("ablate $ab  %-40s avg %8.1f us" % (r["Name"].split("(")[0][:40], float(r["AverageNs"]) / 1e3))
PY
done
