PCS_TEST_VERBOSE=1 timeout 2000 python -m pytest tests/test_fuse.py -m gpu -x -q -s -k "cylinder" 2>&1 | grep -E "plain .* fused|passed|failed|^E  " | cut -c1-200 | head -70
