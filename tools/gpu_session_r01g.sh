#!/bin/bash
# One GPU session: smoke, bench, rocprofv3 kernel stats of the bench command, PMC passes,
# FBGMM / bigram timings (sequential chain and batch sampler).
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01g
mkdir -p $O
cd $R
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cat $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -o stats -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-utts 0 > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err || { tail -20 $O/rocprof_stats.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-utts 0 > $O/bench_pmc_fetch.json 2> $O/pmc_fetch.err || { tail -20 $O/pmc_fetch.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-utts 0 > $O/bench_pmc_write.json 2> $O/pmc_write.err || { tail -20 $O/pmc_write.err; exit 1; }
cd $R
SEGK_SCORE_B3=0 timeout -k 10 400 python bench.py --cpu-utts 0 > $O/bench_fp32_filter.json 2> /dev/null
SEGK_SCORE_B3=3 timeout -k 10 400 python bench.py --cpu-utts 0 > $O/bench_bf16x3_filter.json 2> /dev/null
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_sq -o pmc -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-utts 0 > /dev/null 2> $O/pmc_sq.err
cd $R
timeout -k 10 300 python tools/bench_kmeans_seq.py --utts 10000 --sweeps 2 > $O/kmeans_seq.log 2>&1
timeout -k 10 400 python bench.py --workload bigram_c5 > $O/bench_bigram_c5.json 2> /dev/null
timeout -k 10 400 python bench.py --workload fbgmm_diag_c2 > $O/bench_fbgmm_diag_c2.json 2> /dev/null
timeout -k 10 300 python tools/bench_fbgmm.py --cpu-utts 40 > $O/fbgmm_seq_c2.log 2>&1
timeout -k 10 300 python tools/bench_fbgmm.py --sync batch --cpu-utts 2 > $O/fbgmm_batch_c2.log 2>&1
timeout -k 10 300 python tools/bench_fbgmm.py --sync batch --precision f32 --cpu-utts 10 --utts 10000 --dim 100 --K 1000 --which fixed,bigram > $O/fbgmm_batch_c5.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_c5 -o stats -- python3 $R/tools/bench_fbgmm.py --sync batch --precision f32 --cpu-utts 2 --utts 10000 --dim 100 --K 1000 --which bigram > $O/fbgmm_batch_c5_rocprof.log 2>&1
grep -h -v amdgpu.ids $O/fbgmm_*.log | grep -v "^W2026"
