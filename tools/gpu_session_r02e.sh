#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R
for w in 12 8 6; do
  SEGK_PAIR_WAVES=$w timeout -k 10 300 python bench.py --cpu-utts 0 --no-events > $O/bench_pw$w.json 2> $O/bench_pw$w.err || { tail -20 $O/bench_pw$w.err; exit 1; }
  echo "prio pair2 waves=$w $(cut -c75-170 $O/bench_pw$w.json)"
  SEGK_SCORE_DBG=64 SEGK_PAIR_WAVES=$w timeout -k 10 300 python bench.py --cpu-utts 0 --no-events > $O/bench_noprio_pw$w.json 2> /dev/null
  echo "noprio(sp) pair2 waves=$w $(cut -c75-170 $O/bench_noprio_pw$w.json)"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -o stats -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-utts 0 --no-events > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err || { tail -20 $O/rocprof_stats.err; exit 1; }
cd $R
python tools/trace_timeline.py $(find $O/stats -name "*.db" | head -1) 15 1 | tail -20 | cut -c1-120
