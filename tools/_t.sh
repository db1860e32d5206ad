cd $GRAFT_REPO_ROOT
for f in "" "-DCHAOREC_SPMM_DESC_DIRECT=1"; do
echo "== $f"
CHAOREC_EXTRA_HIPCC_FLAGS="$f" ADAM=1 timeout 600 python tools/gated_bench.py config5 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -3
CHAOREC_EXTRA_HIPCC_FLAGS="$f" timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-regime --no-full-config5 --no-models 2>/dev/null | tail -1 | cut -c150-330
done
