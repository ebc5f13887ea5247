#!/bin/bash
# One GPU session (state r01_h: one-product pre-filter + exact pair kernel + second stream): smoke, bench,
# rocprofv3 kernel stats of the bench command, PMC passes, the other filters, FBGMM / bigram batch lines.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01h
mkdir -p $O
cd $R
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cat $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -o stats -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-utts 0 > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err || { tail -20 $O/rocprof_stats.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-utts 0 > $O/bench_pmc_fetch.json 2> $O/pmc_fetch.err || { tail -20 $O/pmc_fetch.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-utts 0 > $O/bench_pmc_write.json 2> $O/pmc_write.err || { tail -20 $O/pmc_write.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_sq -o pmc -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-utts 0 > /dev/null 2> $O/pmc_sq.err
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES -d $O/pmc_sq2 -o pmc -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-utts 0 > /dev/null 2> $O/pmc_sq2.err
cd $R
SEGK_SCORE_PRE=0 timeout -k 10 400 python bench.py --cpu-utts 0 > $O/bench_fp16x2_filter.json 2> /dev/null
SEGK_SCORE_OVERLAP=0 timeout -k 10 400 python bench.py --cpu-utts 0 > $O/bench_one_stream.json 2> /dev/null
SEGK_SCORE_B3=0 timeout -k 10 400 python bench.py --cpu-utts 0 > $O/bench_fp32_filter.json 2> /dev/null
SEGK_SCORE_B3=3 timeout -k 10 400 python bench.py --cpu-utts 0 > $O/bench_bf16x3_filter.json 2> /dev/null
timeout -k 10 400 python bench.py --workload bigram_c5 > $O/bench_bigram_c5.json 2> /dev/null
timeout -k 10 400 python bench.py --workload fbgmm_diag_c2 > $O/bench_fbgmm_diag_c2.json 2> /dev/null
timeout -k 10 300 python tools/bench_kmeans_seq.py --utts 10000 --sweeps 2 > $O/kmeans_seq.log 2>&1
for f in bench_fp16x2_filter bench_one_stream bench_fp32_filter bench_bf16x3_filter bench_bigram_c5 bench_fbgmm_diag_c2; do echo $f; cut -c1-260 $O/$f.json; done
tail -3 $O/kmeans_seq.log
