# development (make DEV=1 build): where the time of k_fbb_step_diag32 goes -- sweeps per second of fbgmm_diag_c2 with parts of the
# kernel switched off (results wrong).  bits: 1 no terms, 2 no span-score reductions, 4 no DP, 8 no draws, 16 return behind the staging
cd $GRAFT_REPO_ROOT
for dbg in 0 1 2 4 8 15 16; do
    SEGK_STEP_DBG=$dbg timeout -k 10 200 python bench.py --workload fbgmm_diag_c2 --steps 20 --warmup 3 --cpu-utts 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('dbg=$dbg ms_per_step %.4f (per Gibbs step %.1f us)' % (d['ms_per_step'], 1e3 * d['ms_per_step'] / 8))"
done
