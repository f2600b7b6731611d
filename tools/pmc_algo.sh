#!/bin/bash
# usage: bash tools/pmc_algo.sh <algo>   (on the GPU box) - SQ counters of the Phi kernel for one algorithm
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_algo$1; rm -rf $O; mkdir -p $O; cd $R
export PHI_ALGO=$1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/a -- python3 tools/phi_pmc.py > $O/a.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES --kernel-trace --output-format csv -d $O/b -- python3 tools/phi_pmc.py > $O/b.log 2>&1 || exit 1
python3 - <<PY
import csv, glob
for sub in "ab":
    acc = {}
    for f in glob.glob("$O/%s/*/*_counter_collection.csv" % sub):
        for r in csv.DictReader(open(f)):
            if ("phi_accumulate" in r["Kernel_Name"] or "phi_moment_kernel" in r["Kernel_Name"]):
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-24s %14.0f   (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
