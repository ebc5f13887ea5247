#!/bin/bash
# r02_b: fused statistics tail (partials -> finalize -> post), one collective, hipGraph replay.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -30 $O/smoke.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_errors.py tests/test_gpu_checkpoint.py tests/test_gpu_dist.py tests/test_gpu_kmeans.py tests/test_gpu_headline_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for g in 1 0; do
  SEGK_SWEEP_GRAPH=$g timeout -k 10 300 python bench.py --cpu-utts 0 --no-events > $O/bench_graph$g.json 2> $O/bench_graph$g.err || { tail -20 $O/bench_graph$g.err; exit 1; }
  cut -c1-200 $O/bench_graph$g.json
  SEGK_SWEEP_GRAPH=$g timeout -k 10 300 python bench.py --cpu-utts 0 --no-events --utts 1250 > $O/bench_graph${g}_1250.json 2> /dev/null
  cut -c1-200 $O/bench_graph${g}_1250.json
done
cd /tmp && export TMPDIR=/tmp
SEGK_SWEEP_GRAPH=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -o stats -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-utts 0 --no-events > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err || { tail -20 $O/rocprof_stats.err; exit 1; }
cd $R
python tools/rocpd_summary.py stats $(find $O/stats -name "*.db" | head -1) $O/kernel_stats.csv && cat $O/kernel_stats.csv
