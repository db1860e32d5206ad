python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r05_o_gpu_tests.log; cat gpurun_out/r05_o_gpu_tests.log
