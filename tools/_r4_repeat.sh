# flake hunt: tests/test_gpu_round4.py a few times, the failure text kept
for i in 1 2 3 4 5; do
  timeout 400 python -m pytest tests/test_gpu_round4.py tests/test_gpu_dist2.py -q -m gpu -x 2>&1 > gpurun_out/r4_rep_$i.log
  tail -1 gpurun_out/r4_rep_$i.log
  if grep -q "failed" gpurun_out/r4_rep_$i.log; then grep -E "^E  |Error|error" gpurun_out/r4_rep_$i.log | head -30; fi
done
