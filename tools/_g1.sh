python -m pytest tests/test_gpu_sparse_family.py tests/test_gpu_parity.py -k "mcln or sampler" -x -q 2>&1 | tail -15
