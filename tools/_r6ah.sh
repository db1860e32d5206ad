cd $GRAFT_REPO_ROOT
run() { echo "== flags: $1"; CHAOREC_EXTRA_HIPCC_FLAGS="$1" timeout 600 python tools/gated_bench.py config5 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -1; }
run ""
run "-DCHAOREC_SPMM_SP_UH=2 -DCHAOREC_SPMM_SP_MINW=7"
run "-DCHAOREC_SPMM_SP_UH=2 -DCHAOREC_SPMM_SP_MINW=8 -DCHAOREC_SPMM_SP_UNR=2"
run "-DCHAOREC_SPMM_SP_UH=2 -DCHAOREC_SPMM_SP_MINW=7 -DCHAOREC_SPMM_SP_UNR=2"
run "-DCHAOREC_SPMM_SP_UNR=2"
run "-DCHAOREC_SPMM_SP_UNR=3"
