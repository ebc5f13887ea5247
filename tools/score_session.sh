#!/bin/bash
# Headline path: parity (k-means + full-size headline tests), bench, and the timeline of one sweep under rocprofv3.
# usage (on the GPU box): bash tools/score_session.sh <tag> [quick]
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
if [ "$2" != "quick" ]; then
timeout -k 10 600 python -m pytest tests/test_gpu_kmeans.py tests/test_gpu_headline_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
fi
timeout -k 10 300 python bench.py --cpu-utts 0 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cut -c75-330 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/tr -o tr -- python3 $R/bench.py --steps 20 --warmup 3 --windows 3 --cpu-utts 0 > /dev/null 2>&1 || exit 1
cd $R
python tools/trace_timeline.py $(find $O/tr -name "*.db" | head -1) 30 1
