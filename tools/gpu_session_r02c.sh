#!/bin/bash
# r02_c: chunked pre-filter pipeline (h1 chunk i+1 beside exact pair / second stage of chunk i).
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -30 $O/smoke.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_headline_fullsize.py tests/test_gpu_kmeans.py tests/test_gpu_dist.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for ch in 4 2 1 8; do
  SEGK_PRE_CHUNKS=$ch timeout -k 10 300 python bench.py --cpu-utts 0 --no-events > $O/bench_chunks$ch.json 2> $O/bench_chunks$ch.err || { tail -20 $O/bench_chunks$ch.err; exit 1; }
  echo "chunks=$ch $(cut -c1-170 $O/bench_chunks$ch.json)"
done
SEGK_SWEEP_GRAPH=1 timeout -k 10 300 python bench.py --cpu-utts 0 --no-events > $O/bench_graph1.json 2> /dev/null; echo "graph $(cut -c1-170 $O/bench_graph1.json)"
timeout -k 10 300 python bench.py --cpu-utts 0 --no-events --utts 1250 > $O/bench_1250.json 2> /dev/null; echo "1250 $(cut -c1-170 $O/bench_1250.json)"
timeout -k 10 300 python bench.py --cpu-utts 0 --no-events --utts 5000 > $O/bench_5000.json 2> /dev/null; echo "5000 $(cut -c1-170 $O/bench_5000.json)"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -o stats -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-utts 0 --no-events > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err || { tail -20 $O/rocprof_stats.err; exit 1; }
cd $R
python tools/rocpd_summary.py stats $(find $O/stats -name "*.db" | head -1) $O/kernel_stats.csv && head -16 $O/kernel_stats.csv | cut -c1-200
python tools/trace_timeline.py $(find $O/stats -name "*.db" | head -1) 15 4 | tail -34 | cut -c1-120
