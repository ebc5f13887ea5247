#!/bin/bash
# Exact-stage variants of the headline sweep: bench value and the score-stage part of the timeline for each setting.
# usage: bash tools/score_variants.sh <tag> "ENV1=a ENV2=b" "ENV1=c" ...
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1))
  echo "== $v"
  ( export $v; python3 $R/bench.py --cpu-utts 0 --windows 5 2>/dev/null | cut -c75-140 )
  ( export $v; rocprofv3 --kernel-trace -d $O/v$i -o tr -- python3 $R/bench.py --steps 20 --warmup 3 --windows 2 --cpu-utts 0 > /dev/null 2>&1 )
  python3 $R/tools/trace_timeline.py $(find $O/v$i -name "*.db" | head -1) 25 1 | grep -E "score_h1|pair|score_sp|brute|segment"
done
