cd $GRAFT_REPO_ROOT
echo "== PF2=1 (product)"; timeout 300 python tools/gemm_wide_bench.py 2>&1 | grep -v "Warn\|amdgpu.ids" | cut -c1-200
echo "== PF2=0 (experiment)"; CHAOREC_EXTRA_HIPCC_FLAGS="-DCHAOREC_X3_PF2=0" timeout 300 python tools/gemm_wide_bench.py 2>&1 | grep -v "Warn\|amdgpu.ids" | cut -c1-200
for m in MMGCN FREEDOM; do timeout 600 python bench.py --model $m --steps 200 --warmup 20 2>/dev/null | tail -1 | cut -c1-330; done
for m in MMGCN; do CHAOREC_EXTRA_HIPCC_FLAGS="-DCHAOREC_X3_PF2=0" timeout 600 python bench.py --model $m --steps 200 --warmup 20 2>/dev/null | tail -1 | cut -c1-330; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round6.py -q -m gpu -x 2>&1 | tail -2
