python -m pytest tests -q -m gpu -x -k "failed_redo_pass or share_a_redo or two_begun" 2>&1 | tail -15
