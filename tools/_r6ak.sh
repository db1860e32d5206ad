cd $GRAFT_REPO_ROOT
for f in "" "-DCHAOREC_EXPAND_EP=2" "-DCHAOREC_EXPAND_EP=16" "-DCHAOREC_EXPAND_EP=32"; do echo "== $f"; CHAOREC_EXTRA_HIPCC_FLAGS="$f" timeout 600 python tools/rowlist_n1_bench.py config5 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -3 | head -1; done
