cd $GRAFT_REPO_ROOT
timeout 900 python tools/rowlist_r0_bench.py config5 2>&1 | grep -v Warn | tail -9
