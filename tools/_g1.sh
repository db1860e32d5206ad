R=$GRAFT_REPO_ROOT
PMC_PASSES=6 timeout 400 python3 tools/pmc_kernels.py score_sweep -- python3 $R/tools/score_case.py 262144 262144 128 > gpurun_out/r05_b_score_pmc_d128_tcc.txt 2>&1
