# round 6, batch aa: XCD grouping of the bf16x3 GEMM's workgroups, A/B
cd $GRAFT_REPO_ROOT
for x in 0 1; do
  echo "== CHAOREC_X3_XCD=$x gemm_wide_bench mmgcn"; CHAOREC_X3_XCD=$x timeout 300 python tools/gemm_wide_bench.py 2>&1 | grep -v Warn | cut -c1-200
  echo "== CHAOREC_X3_XCD=$x bench --model MMGCN / FREEDOM"
  for m in MMGCN FREEDOM; do CHAOREC_X3_XCD=$x timeout 600 python bench.py --model $m --steps 200 --warmup 20 2>/dev/null | tail -1 | cut -c1-400; done
done
echo "== GEMM users' tests"; timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round6.py tests/test_gpu_fused_step.py -q -m gpu -x 2>&1 | tail -5 | cut -c1-300
for x in 0 1; do
  echo "== CHAOREC_SWEEP_XCD=$x default bench (sports + hbm_regime + config5 whole)"
  CHAOREC_SWEEP_XCD=$x CHAOREC_BENCH_DETAIL=gpurun_out/r06_aa_detail_sweepxcd$x.json timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-3000
done
