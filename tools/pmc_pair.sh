#!/bin/bash
# Cache counters of the exact pair stage run alone (SEGK_SCORE_OVERLAP=0).  usage: bash tools/pmc_pair.sh <tag>
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SEGK_SCORE_OVERLAP=0
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_READ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCC_READ_SECTORS_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o pmc -- python3 $R/bench.py --steps 4 --warmup 2 --windows 1 --cpu-utts 0 > /dev/null 2> $O/p$i.err || { tail -5 $O/p$i.err; continue; }
  python $R/tools/rocpd_summary.py pmc $(find $O/p$i -name "*.db" | head -1) $O/p$i.csv
  grep -E "exact_pair2|score_sp" $O/p$i.csv | cut -c1-30,60-200
done
