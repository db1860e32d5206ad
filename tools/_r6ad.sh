cd $GRAFT_REPO_ROOT
timeout 600 python tools/rowlist_r0_bench.py config5 2>&1 | grep -v Warn | tail -5
timeout 600 python tools/rowlist_r0_bench.py config5 2>&1 | grep -v Warn | tail -4
timeout 600 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round4.py tests/test_gpu_config5_full.py -q -m gpu -x 2>&1 | tail -2
