#!/bin/bash
# Sequential k-means chain: parity subset, then per-kernel durations of tools/bench_kmeans_seq.py under rocprofv3.
# usage (on the GPU box): bash tools/seq_session.sh <tag>
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_kmeans.py tests/test_gpu_edge.py tests/test_gpu_checkpoint.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -o stats -- python3 $R/tools/bench_kmeans_seq.py --utts 2000 --sweeps 3 > $O/seq.log 2>&1 || { tail -20 $O/seq.log; exit 1; }
cd $R
grep sample_time $O/seq.log
python - $O <<'PY'
import sqlite3, glob, sys, numpy as np
c = sqlite3.connect(glob.glob(sys.argv[1] + "/stats/**/*.db", recursive=True)[0])
for n in ("k_seq_update", "k_seq_score", "k_kmeans_segment_w8", "k_seq_chain"):
    d = np.array([r[0] for r in c.execute("select duration/1e3 from kernels where name like ? order by start", ("%" + n + "%",))])
    if len(d):
        print(n, "mean %.1f  p10/50/90 %s  n %d" % (d.mean(), np.percentile(d, [10, 50, 90]).round(1), len(d)))
PY
