# the whole GPU suite once, the tail (with any failure's text) kept
timeout 2300 python -m pytest tests -q -m gpu 2>&1 | tail -60 > gpurun_out/r05_zzz_gpu_tests.log
grep -E "^FAILED|passed|failed|^E  " gpurun_out/r05_zzz_gpu_tests.log | cut -c1-300 | tail -20
