#!/bin/bash
# Instruction-fetch / wait counters of the persistent FBGMM chain (one --pmc pass per counter group, kernel trace only):
#   tools/pmc_fb_chain.sh TAG
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS" \
         "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM SQ_INSTS_FLAT"; do
    i=$((i + 1))
    (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $G -d $O/pmc$i -o pmc -- python3 $R/tools/bench_fbgmm.py --which diag --cpu-utts 0 --sweeps 2 > /dev/null 2> $O/pmc$i.err) || { tail -5 $O/pmc$i.err; continue; }
    (cd $R && python tools/rocpd_summary.py pmc $(find $O/pmc$i -name "*.db" | head -1) $O/pmc$i.csv && grep -E "Kernel|k_fb_chain" $O/pmc$i.csv | cut -c1-400)
    find $O/pmc$i -name "*.db" -delete
done
