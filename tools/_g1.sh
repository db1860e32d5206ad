timeout 1500 python -m pytest tests/test_gpu_sparse_family.py -q -m gpu -x -k "round5_members and (POWERec or SMORE)" 2>&1 | grep -v ' INFO ' | tail -30
