#!/bin/bash
# One parameterised measurement session (replaces the per-state gpu_session_r0*.sh scripts of rounds 1-2; `git log` has them).
#   tools/gpu_session.sh TAG [STEP ...]      output under gpurun_out/TAG/; steps (default: bench stats timeline):
#     smoke     __graft_entry__.smoke()
#     bench     python bench.py (default arguments) and with the driver's --steps 20 --warmup 5
#     stats     rocprofv3 --kernel-trace --stats of the bench command -> kernel_stats.csv
#     timeline  start / end / duration of the kernels of one sweep from the stats trace -> timeline.txt
#     pmc       separate --pmc passes: FETCH_SIZE, WRITE_SIZE, SQ busy / wait counters -> pmc_*.csv
#     shards    bench.py --utts 5000 / 2500 / 1250 (the per-rank share at 2 / 4 / 8 GPUs, no collective)
#     variants  the other filters (SEGK_SCORE_HINT=0, SEGK_SCORE_PRE=0, SEGK_SCORE_B3=0)
#     workloads bench.py --workload bigram_c5 / fbgmm_diag_c2 / kmeans_c3_sequential
#     rehearse  bench.py --gpus 2 / 4 self-launched (4 ranks + launcher + parent: the box admits six processes), gloo, all ranks on the one card (plumbing of the multi-GPU run)
#     clean     delete the rocpd databases of this session (after stats / timeline / pmc have been summarised)
# Copy what is to be judged from gpurun_out/TAG/ into profiles/ (named rNN_TAG_*).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
STEPS=${*:-bench stats timeline}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
v() { echo "$1 $(cut -c75-170 $2)"; }
for S in $STEPS; do
case $S in
smoke) timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }; tail -1 $O/smoke.log ;;
bench)
    timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
    cut -c1-420 $O/bench.json
    timeout -k 10 500 python bench.py --steps 20 --warmup 5 --cpu-utts 0 --no-early > $O/bench_driver_args.json 2> /dev/null; cut -c1-260 $O/bench_driver_args.json ;;
stats)
    (cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/stats -o stats -- python3 $R/bench.py --steps 20 --warmup 3 --windows 3 --min-seconds 0 --no-seq-chain --no-early --cpu-utts 0 > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err) || { tail -20 $O/rocprof_stats.err; exit 1; }
    python tools/rocpd_summary.py stats $(find $O/stats -name "*.db" | head -1) $O/kernel_stats.csv
    head -16 $O/kernel_stats.csv | cut -c1-150 ;;
timeline) python tools/trace_timeline.py $(find $O/stats -name "*.db" | head -1) 15 1 > $O/timeline.txt; cat $O/timeline.txt ;;
clean) find $O -name "*.db" -delete ;;      # the rocpd databases are tens of MB each: gpurun_out/ travels back only under 64 MiB
pmc)
    for C in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$C -o pmc -- python3 $R/bench.py --steps 5 --warmup 2 --windows 1 --min-seconds 0 --no-seq-chain --no-early --cpu-utts 0 > /dev/null 2> $O/pmc_$C.err)
        python tools/rocpd_summary.py pmc $(find $O/pmc_$C -name "*.db" | head -1) $O/pmc_$(echo $C | tr A-Z a-z).csv
    done
    (cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_sq -o pmc -- python3 $R/bench.py --steps 5 --warmup 2 --windows 1 --min-seconds 0 --no-seq-chain --no-early --cpu-utts 0 > /dev/null 2> $O/pmc_sq.err)
    python tools/rocpd_summary.py pmc $(find $O/pmc_sq -name "*.db" | head -1) $O/pmc_sq.csv
    find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_sq -name "*.db" -delete
    grep -E "top2_rs|hint_exact|score_h1|exact_pair4" $O/pmc_fetch_size.csv $O/pmc_write_size.csv $O/pmc_sq.csv | cut -c1-230 ;;
shards) for u in 5000 2500 1250; do timeout -k 10 300 python bench.py --cpu-utts 0 --no-early --no-seq-chain --utts $u > $O/bench_$u.json 2> /dev/null; v "utts=$u" $O/bench_$u.json; done ;;
variants)
    SEGK_SCORE_HINT=0 timeout -k 10 400 python bench.py --cpu-utts 0 --no-early --no-seq-chain > $O/bench_no_hint.json 2> /dev/null; v no_hint $O/bench_no_hint.json
    SEGK_SCORE_HINT=0 SEGK_SCORE_PRE=0 timeout -k 10 400 python bench.py --cpu-utts 0 --no-early --no-seq-chain > $O/bench_fp16x2_filter.json 2> /dev/null; v fp16x2 $O/bench_fp16x2_filter.json
    SEGK_SCORE_B3=0 timeout -k 10 400 python bench.py --cpu-utts 0 --no-early --no-seq-chain > $O/bench_fp32_filter.json 2> /dev/null; v fp32 $O/bench_fp32_filter.json ;;
workloads)
    timeout -k 10 400 python bench.py --workload bigram_c5 > $O/bench_bigram_c5.json 2> /dev/null; cut -c1-200 $O/bench_bigram_c5.json
    timeout -k 10 400 python bench.py --workload fbgmm_diag_c2 > $O/bench_fbgmm_diag_c2.json 2> /dev/null; cut -c1-200 $O/bench_fbgmm_diag_c2.json
    timeout -k 10 400 python bench.py --workload kmeans_c3_sequential > $O/bench_kmeans_c3_sequential.json 2> /dev/null; cut -c1-200 $O/bench_kmeans_c3_sequential.json
    timeout -k 10 400 python bench.py --workload fbgmm_c2_sequential > $O/bench_fbgmm_c2_sequential.json 2> /dev/null; cut -c1-260 $O/bench_fbgmm_c2_sequential.json
    timeout -k 10 400 python bench.py --workload bigram_c2_sequential > $O/bench_bigram_c2_sequential.json 2> /dev/null; cut -c1-260 $O/bench_bigram_c2_sequential.json ;;
rehearse)   # the N > 1 launcher and transport on ONE card: `python bench.py --gpus N` starts its own torchrun child, gloo backend
    for n in 2 4; do timeout -k 10 500 python bench.py --gpus $n --cpu-utts 0 --steps 20 --warmup 5 > $O/bench_gpus${n}_gloo.json 2> $O/bench_gpus${n}_gloo.err || { tail -20 $O/bench_gpus${n}_gloo.err; exit 1; }; v "gpus=$n" $O/bench_gpus${n}_gloo.json; done ;;
*) echo "unknown step $S"; exit 2 ;;
esac
done
