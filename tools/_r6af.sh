cd $GRAFT_REPO_ROOT
KNOCKOUT=1 timeout 900 python tools/gated_bench.py config5 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -10
