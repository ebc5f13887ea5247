# development: the 4-rank prefilter run of tests/test_gpu_dist.py, N times, with the tile-image check of tools/diag_dist_worker.py
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${1:-30}); do
  SEGK_SCORE_PRE=1 MASTER_ADDR=127.0.0.1 timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29511 tools/diag_dist_worker.py 3 2>&1 | grep -E "^TILES|^SCORES|^WRONG|^TOTALS rank 0" | sed "s/^/run $i: /"
done
