set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_l; mkdir -p $O; cd $R
for u in 5000 2500 1250; do timeout -k 10 300 python bench.py --cpu-utts 0 --no-early --no-seq-chain --utts $u > $O/bench_$u.json 2> /dev/null; echo "utts=$u $(cut -c75-170 $O/bench_$u.json)"; done
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o stats -- python3 $R/bench.py --steps 20 --warmup 3 --windows 3 --min-seconds 0 --no-seq-chain --no-early --cpu-utts 0 --utts 1250 > $O/bench_under_rocprof_1250.json 2> $O/rocprof.err)
python tools/trace_timeline.py $(find $O/stats -name "*.db" | head -1) 16 1 > $O/timeline_1250.txt; cat $O/timeline_1250.txt
find $O -name "*.db" -delete
