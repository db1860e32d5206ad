timeout 900 python -m pytest tests/test_gpu_sparse_family.py -q -m gpu -k "round5_members and MMGCL" 2>&1 | grep -v ' INFO ' | tail -30
