#!/bin/bash
# SQ busy / wait counters of a secondary workload of bench.py (one --pmc pass, kernel trace only):  tools/pmc_workload.sh TAG WORKLOAD
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1; W=$2
mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_sq_$W -o pmc -- python3 $R/bench.py --workload $W --steps 3 --warmup 1 --windows 1 --cpu-utts 0 > /dev/null 2> $O/pmc_sq_$W.err) || { tail -20 $O/pmc_sq_$W.err; exit 1; }
cd $R
python tools/rocpd_summary.py pmc $(find $O/pmc_sq_$W -name "*.db" | head -1) $O/pmc_sq_$W.csv
grep -E "score_sp|assign_lm" $O/pmc_sq_$W.csv | cut -c1-200
find $O/pmc_sq_$W -name "*.db" -delete
