#!/bin/bash
# development: score kernels, per-kernel times
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pre
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/s1 -o stats -- python3 $R/tools/diag_score.py 1048576 100 1000 5 > $O/s1.log 2>&1
python3 $R/tools/rocpd_summary.py stats $O/s1/stats_results.db $O/s1.csv
grep "score_\|pair\|brute" $O/s1.csv | cut -c1-44,60-
