# round 6, batch d: per-(kernel, grid) tables of the MMGCN / FREEDOM captured steps
cd /tmp && export TMPDIR=/tmp
for m in MMGCN FREEDOM; do
  out=$GRAFT_REPO_ROOT/gpurun_out/trace_$m
  rm -rf $out
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 $GRAFT_REPO_ROOT/bench.py --model $m --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/trace_$m.log 2>&1 < /dev/null
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/trace_$m.log | cut -c1-400
  python3 $GRAFT_REPO_ROOT/tools/kernel_table.py $out 0 0.3 > $GRAFT_REPO_ROOT/gpurun_out/r06_d_${m}_kernel_table.txt
  head -60 $GRAFT_REPO_ROOT/gpurun_out/r06_d_${m}_kernel_table.txt
  rm -rf $out
done
cd $GRAFT_REPO_ROOT
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round4.py "tests/test_gpu_epoch_parity.py::test_multimodal_baby_epochs_agree_with_the_reference_in_distribution[FREEDOM]" -q -m gpu -x 2>&1 | tail -15
echo "== build time probe (configs[4] whole)"; timeout 600 python tools/build_time_probe.py config5 2>&1 | tail -12
echo "== PMC: the D = 64 scoring kernels of the steady sports call"
EPOCH_APART=1 PMC_PASSES=0,1,2,3,7 timeout 900 python tools/pmc_kernels.py score_sweep score_select -- python3 $GRAFT_REPO_ROOT/tools/score_profile.py 3000 > gpurun_out/r06_d_score_pmc_d64.txt 2>&1
cat gpurun_out/r06_d_score_pmc_d64.txt | tail -60
