# development (make DEV=1 build): where the time of k_fbb_score_diag32 goes -- the launch timed by the library's events with
# parts of the kernel switched off (results wrong).  bits: 1 no table fill, 2 no terms, 8 return before the reductions,
# 16 return behind the row staging, 32 return at once (the empty launch)
cd $GRAFT_REPO_ROOT
for dbg in 0 1 2 3 11 19 32; do
    SEGK_D32_DBG=$dbg timeout -k 10 200 python bench.py --workload fbgmm_diag_c2 --steps 20 --warmup 3 --cpu-utts 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('dbg=$dbg ms_per_launch %.4f  ms_per_step %.4f' % (d['roofline']['ms_per_launch'], d['ms_per_step']))"
done
