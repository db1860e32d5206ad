cd $GRAFT_REPO_ROOT
for t in 0 16384 32768 65536 131072; do
echo "== long-list stripe threshold $t"; CHAOREC_ROWLIST_STRIPE_T_LONGLIST=$t timeout 600 python tools/rowlist_n1_bench.py config5 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -2 | head -1
done
