#!/bin/bash
O=gpurun_out/${1:-qt}; mkdir -p $O
( timeout 600 python -m pytest tests/test_tennis.py -q -x -s -k bf16 ) > $O/pytest_tennis.log 2>&1; grep -E "passed|failed|rror|PSNR" $O/pytest_tennis.log | tail -5
