python3 tools/collect_profiles.py r05_a config5 > gpurun_out/r05_collect_a.log 2>&1; tail -3 gpurun_out/r05_collect_a.log
CHAOREC_REUSE_STATS=0 python3 tools/collect_profiles.py r05_a sports > gpurun_out/r05_collect_b.log 2>&1; tail -3 gpurun_out/r05_collect_b.log
