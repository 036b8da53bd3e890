cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -s -k "32_row" 2>&1 | grep -E "WINDOW32|passed|failed|Error|assert" | cut -c1-300
bash tools/r04_win32_thr.sh r04n
