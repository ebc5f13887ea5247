#!/bin/bash
# SQ / GRBM counters of the score kernels of the bench command (one pass), for the current build and environment:
#   tools/pmc_k1.sh TAG
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_sq -o pmc -- python3 $R/bench.py --steps 5 --warmup 3 --windows 1 --min-seconds 0 --cpu-utts 0 --no-seq-chain --no-early > /dev/null 2> $O/pmc_sq.err
cd $R
python tools/rocpd_summary.py pmc $(find $O/pmc_sq -name "*.db" | head -1) $O/pmc_sq.csv
grep -E "top2_rs|hint_exact" $O/pmc_sq.csv | cut -c1-200
