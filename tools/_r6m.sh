cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
  timeout 600 python -m pytest tests/test_gpu_dist2.py -q -m gpu -x 2>&1 > gpurun_out/dist2_rep_$i.log
  tail -1 gpurun_out/dist2_rep_$i.log
  if grep -q "failed" gpurun_out/dist2_rep_$i.log; then grep -E "^E  |Error|error" gpurun_out/dist2_rep_$i.log | head -20 | cut -c1-300; fi
done
