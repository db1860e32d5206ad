python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r05_z_gpu_tests.log; cat gpurun_out/r05_z_gpu_tests.log
python3 tools/collect_profiles.py r05_z config5 > gpurun_out/r05_collect_z_a.log 2>&1; tail -3 gpurun_out/r05_collect_z_a.log
CHAOREC_REUSE_STATS=0 python3 tools/collect_profiles.py r05_z sports > gpurun_out/r05_collect_z_b.log 2>&1; tail -3 gpurun_out/r05_collect_z_b.log
