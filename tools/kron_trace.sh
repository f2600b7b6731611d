#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kron_trace; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 tools/kron_probe.py > $O/log.txt 2>&1 || exit 1
head -12 $O/t/*/*_kernel_stats.csv | cut -c1-160
