#!/bin/bash
# r02_z: final state of round 2 -- smoke, bench, rocprofv3 kernel stats of the bench command, PMC passes, variants, secondary
# workloads, the sequential chain.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02z
mkdir -p $O
cd $R
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cut -c1-400 $O/bench.json
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> /dev/null; cut -c1-260 $O/bench_driver_args.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -o stats -- python3 $R/bench.py --steps 20 --warmup 3 --windows 3 --cpu-utts 0 > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err || { tail -20 $O/rocprof_stats.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc -- python3 $R/bench.py --steps 5 --warmup 2 --windows 1 --cpu-utts 0 > $O/bench_pmc_fetch.json 2> $O/pmc_fetch.err || { tail -20 $O/pmc_fetch.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc -- python3 $R/bench.py --steps 5 --warmup 2 --windows 1 --cpu-utts 0 > $O/bench_pmc_write.json 2> $O/pmc_write.err || { tail -20 $O/pmc_write.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_sq -o pmc -- python3 $R/bench.py --steps 5 --warmup 2 --windows 1 --cpu-utts 0 > /dev/null 2> $O/pmc_sq.err
timeout -k 10 400 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum -d $O/pmc_cache -o pmc -- python3 $R/bench.py --steps 5 --warmup 2 --windows 1 --cpu-utts 0 > /dev/null 2> $O/pmc_cache.err
cd $R
DB=$(find $O/stats -name "*.db" | head -1)
python tools/rocpd_summary.py stats $DB $O/kernel_stats.csv
python tools/rocpd_summary.py pmc $(find $O/pmc_fetch -name "*.db" | head -1) $O/pmc_fetch_size.csv
python tools/rocpd_summary.py pmc $(find $O/pmc_write -name "*.db" | head -1) $O/pmc_write_size.csv
python tools/rocpd_summary.py pmc $(find $O/pmc_sq -name "*.db" | head -1) $O/pmc_sq.csv
python tools/rocpd_summary.py pmc $(find $O/pmc_cache -name "*.db" | head -1) $O/pmc_cache.csv
python tools/trace_timeline.py $DB 15 1 > $O/timeline.txt
head -14 $O/kernel_stats.csv | cut -c1-140
grep -E "score_h1|exact_pair4" $O/pmc_fetch_size.csv $O/pmc_write_size.csv $O/pmc_cache.csv | cut -c1-220
grep "score_h1" $O/pmc_sq.csv | cut -c1-200
v() { echo "$1 $(cut -c75-150 $2)"; }
SEGK_SWEEP_GRAPH=1 timeout -k 10 400 python bench.py --cpu-utts 0 > $O/bench_graph.json 2> /dev/null; v graph $O/bench_graph.json
SEGK_SCORE_PRE=0 timeout -k 10 400 python bench.py --cpu-utts 0 > $O/bench_fp16x2_filter.json 2> /dev/null; v fp16x2 $O/bench_fp16x2_filter.json
SEGK_SCORE_B3=0 timeout -k 10 400 python bench.py --cpu-utts 0 > $O/bench_fp32_filter.json 2> /dev/null; v fp32 $O/bench_fp32_filter.json
SEGK_SCORE_OVERLAP=1 SEGK_PAIR4_WAVES=4 timeout -k 10 400 python bench.py --cpu-utts 0 > $O/bench_two_streams.json 2> /dev/null; v two_streams $O/bench_two_streams.json
SEGK_PAIR_V=2 timeout -k 10 400 python bench.py --cpu-utts 0 > $O/bench_pair2.json 2> /dev/null; v pair2 $O/bench_pair2.json
SEGK_PAIR_V=3 timeout -k 10 400 python bench.py --cpu-utts 0 > $O/bench_pair3.json 2> /dev/null; v pair3 $O/bench_pair3.json
SEGK_SEGMENT_X2=1 timeout -k 10 400 python bench.py --cpu-utts 0 > $O/bench_segment_x2.json 2> /dev/null; v segment_x2 $O/bench_segment_x2.json
for u in 5000 2500 1250; do timeout -k 10 300 python bench.py --cpu-utts 0 --utts $u > $O/bench_$u.json 2> /dev/null; v "utts=$u" $O/bench_$u.json; done
timeout -k 10 400 python bench.py --workload bigram_c5 > $O/bench_bigram_c5.json 2> /dev/null; cut -c1-200 $O/bench_bigram_c5.json
timeout -k 10 400 python bench.py --workload fbgmm_diag_c2 > $O/bench_fbgmm_diag_c2.json 2> /dev/null; cut -c1-200 $O/bench_fbgmm_diag_c2.json
SEGK_CHAIN_STAMP=1 timeout -k 10 300 python tools/bench_kmeans_seq.py --utts 10000 --sweeps 4 > $O/kmeans_seq.log 2>&1; grep -E "chain stamps|of dp" $O/kmeans_seq.log | tail -2; tail -1 $O/kmeans_seq.log
SEGK_SEQ_CHAIN=0 timeout -k 10 300 python tools/bench_kmeans_seq.py --utts 2000 --sweeps 3 > $O/kmeans_seq_three_launches.log 2>&1; tail -1 $O/kmeans_seq_three_launches.log
timeout -k 10 300 python tools/time_records.py > $O/time_records.log 2>&1; tail -3 $O/time_records.log
timeout -k 10 300 python tools/diag_queue_len.py > $O/stage_counts.log 2>&1; tail -2 $O/stage_counts.log
