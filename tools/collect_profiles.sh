#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace stats of the bench command, the two separate HBM-traffic PMC passes and the SQ counter
# passes of the Phi kernel (unsorted and sorted input), the dependent-schedule timeline, the Phi ablation harness, the M-side probe,
# kernel-trace stats of the Kronecker and the posterior probes.  Outputs under gpurun_out/prof_r04/ ; summarised into profiles/ by
# tools/summarise_profiles.py (run afterwards, on CPU).  usage: bash tools/collect_profiles.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 30 --warmup 3 --repeats 5 --no-cpu-baseline --no-extras > $O/trace_bench.json 2> $O/trace.err || exit 1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 tools/phi_pmc.py > $O/fetch.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 tools/phi_pmc.py > $O/write.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/sq -- python3 tools/phi_pmc.py > $O/sq.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES --kernel-trace --output-format csv -d $O/sq2 -- python3 tools/phi_pmc.py > $O/sq2.log 2>&1 || exit 1
PHI_SORTED=1 timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_sorted -- python3 tools/phi_pmc.py > $O/fetch_sorted.log 2>&1 || exit 1
PHI_SORTED=1 timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_sorted -- python3 tools/phi_pmc.py > $O/write_sorted.log 2>&1 || exit 1
PHI_SORTED=1 timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/sq_sorted -- python3 tools/phi_pmc.py > $O/sq_sorted.log 2>&1 || exit 1
PHI_SORTED=1 timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES --kernel-trace --output-format csv -d $O/sq2_sorted -- python3 tools/phi_pmc.py > $O/sq2_sorted.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/timeline -- python3 bench.py --no-cpu-baseline --in-flight 0 --no-three-sets --repeats 3 > $O/timeline_bench.json 2> $O/timeline.err || exit 1
python3 tools/dep_timeline.py $(ls -t $(find $O/timeline -name "*kernel_trace.csv") | head -1) 60 > $O/dependent_timeline.txt 2>&1 || exit 1
timeout -k 10 300 python3 tools/phi_ablation.py $O/phi_ablation.json > $O/phi_ablation.log 2>&1 || exit 1
ASVGP_CHAIN_STAMPS=1 timeout -k 10 200 python3 tools/mside_probe.py > $O/mside_probe.txt 2>&1 || exit 1
timeout -k 10 100 tools/micro/bin/bcr_mfma_bench 2048 > $O/bcr_mfma_bench.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/phi_probe.py 10000000 > $O/phi_probe_10m.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/phi_probe.py 1250000 > $O/phi_probe_1250k.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/vjp_probe.py > $O/vjp_probe.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/prior_dd_probe.py > $O/prior_dd_probe.txt 2>&1 || exit 1
timeout -k 10 400 python3 tools/dep_probe.py > $O/dep_probe.txt 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kron -- python3 tools/kron_trace.py > $O/kron.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/predict -- python3 tools/predict_probe.py > $O/predict.log 2>&1 || exit 1
timeout -k 10 200 python3 tools/ahead_probe.py > $O/ahead_probe.txt 2>&1 || exit 1
timeout -k 10 300 python3 tools/kron_probe.py > $O/kron_probe_untraced.txt 2>&1 || exit 1
KTWIST=0 timeout -k 10 300 python3 tools/kron_probe.py > $O/kron_probe_onesided.txt 2>&1 || exit 1
timeout -k 10 100 tools/micro/bin/reduce_bench > $O/reduce_bench.txt 2>&1 || exit 1
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
timeout -k 10 300 python3 bench.py --sorted > $O/bench_sorted.json 2> $O/bench_sorted.err || exit 1
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*.csv" | head -40
