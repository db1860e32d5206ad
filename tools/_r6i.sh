cd $GRAFT_REPO_ROOT
echo "== failing tests"; timeout 1500 python -m pytest "tests/test_gpu_dist2.py::test_captured_exchanges_on_a_one_rank_rccl_group" "tests/test_gpu_epoch_parity.py::test_multimodal_baby_epochs_agree_with_the_reference_in_distribution[FREEDOM]" -q -m gpu 2>&1 | grep -v "^  /\|Warning\|warnings.warn" | tail -60 | cut -c1-700
echo "== gemm shapes"; timeout 300 python tools/gemm_wide_bench.py mmgcn 2>&1 | tail -8
for m in MMGCN FREEDOM; do
  timeout 600 python bench.py --model $m --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'], d.get('roofline'))"
done
