#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02i
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_fbgmm_batch.py tests/test_gpu_statistical.py tests/test_gpu_dist.py tests/test_gpu_headline_fullsize.py -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|diag f32|Error|error" $O/pytest.log | tail -12 | cut -c1-300
timeout -k 10 300 python bench.py --workload fbgmm_diag_c2 --cpu-utts 0 > $O/bench_diag_f32.json 2> $O/bench_diag_f32.err; cut -c1-1500 $O/bench_diag_f32.json; tail -3 $O/bench_diag_f32.err
SEGK_FBB_PRECISION=f64 timeout -k 10 300 python bench.py --workload fbgmm_diag_c2 --cpu-utts 0 > $O/bench_diag_f64.json 2> /dev/null; cut -c1-260 $O/bench_diag_f64.json
timeout -k 10 300 python bench.py --workload bigram_c5 --cpu-utts 0 > $O/bench_bigram.json 2> /dev/null; cut -c1-300 $O/bench_bigram.json
