python bench.py --no-cpu-baseline --no-hbm-regime 2>gpurun_out/r05_n_err.log | tail -1 > gpurun_out/r05_n_line.json; tail -c 300 gpurun_out/r05_n_err.log
