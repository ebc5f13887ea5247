#!/bin/bash
# r02_d: exact pair stage, second form (means from L2 into registers, rows through LDS)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02d
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_headline_fullsize.py tests/test_gpu_kmeans.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for w in 12 8 6; do
  SEGK_PAIR_WAVES=$w timeout -k 10 300 python bench.py --cpu-utts 0 --no-events > $O/bench_pw$w.json 2> $O/bench_pw$w.err || { tail -20 $O/bench_pw$w.err; exit 1; }
  echo "pair2 waves=$w $(cut -c75-170 $O/bench_pw$w.json)"
done
SEGK_PAIR_V=1 timeout -k 10 300 python bench.py --cpu-utts 0 --no-events > $O/bench_pair1.json 2> /dev/null; echo "pair1 $(cut -c75-170 $O/bench_pair1.json)"
SEGK_SCORE_OVERLAP=0 timeout -k 10 300 python bench.py --cpu-utts 0 --no-events > $O/bench_onestream.json 2> /dev/null; echo "one stream $(cut -c75-170 $O/bench_onestream.json)"
timeout -k 10 300 python bench.py --cpu-utts 0 --no-events --utts 1250 > $O/bench_1250.json 2> /dev/null; echo "1250 $(cut -c75-170 $O/bench_1250.json)"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -o stats -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-utts 0 --no-events > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err || { tail -20 $O/rocprof_stats.err; exit 1; }
cd $R
python tools/rocpd_summary.py stats $(find $O/stats -name "*.db" | head -1) $O/kernel_stats.csv && head -14 $O/kernel_stats.csv | cut -c1-160
python tools/trace_timeline.py $(find $O/stats -name "*.db" | head -1) 15 1 | tail -20 | cut -c1-120
