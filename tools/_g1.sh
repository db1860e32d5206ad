timeout 900 python -m pytest tests/test_gpu_sparse_family.py -q -m gpu -k "fkan or FKAN" 2>&1 | grep -v ' INFO ' | tail -30
