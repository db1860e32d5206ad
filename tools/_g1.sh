# scratch: one gpurun call's worth of commands (edit, then `gpurun -- 'bash tools/_g1.sh'`)
timeout 1200 python -m pytest tests/test_gpu_sparse_family.py -q -m gpu -x 2>&1 | tail -5
