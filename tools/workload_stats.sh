#!/bin/bash
# rocprofv3 kernel statistics of a secondary workload of bench.py:   tools/workload_stats.sh TAG WORKLOAD
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1; W=$2
mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/stats_$W -o stats -- python3 $R/bench.py --workload $W --steps 10 --warmup 2 --windows 2 --cpu-utts 0 > $O/bench_${W}_under_rocprof.json 2> $O/rocprof_$W.err) || { tail -20 $O/rocprof_$W.err; exit 1; }
cd $R
python tools/rocpd_summary.py stats $(find $O/stats_$W -name "*.db" | head -1) $O/${W}_kernel_stats.csv
head -14 $O/${W}_kernel_stats.csv | cut -c1-170
find $O/stats_$W -name "*.db" -delete
