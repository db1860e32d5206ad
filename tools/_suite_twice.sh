# stability check: the whole GPU suite twice, failures listed
for i in 1 2; do timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed" | tail -5; done
