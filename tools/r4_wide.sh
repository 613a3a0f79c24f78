#!/bin/bash
# round 4: the wide-tile split-operand Winograd kernel (tile code 6064): parity tests, then device times next to the other shapes
O=gpurun_out/${1:-r4w}; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_x3.py -q -x -k "winograd_x3" ) > $O/pytest_w3.log 2>&1; grep -a -E "passed|failed|rror" $O/pytest_w3.log | tail -5
timeout 600 python tools/x3_bench.py encoder.10,encoder.8,encoder.16,encoder.12,decoder.0,decoder.2,decoder.4,encoder.2,encoder.6,conv_offset.6,backbone.0 99 > $O/x3_bench.txt 2>&1
cut -c1-100,0-0 $O/x3_bench.txt | head -0
python - "$O/x3_bench.txt" <<'PY'
import sys, re
for l in open(sys.argv[1]):
    if "wino-x3" in l:
        print(l.split("|")[0].strip(), "|", l.split("|")[1].strip(), "|", l.split("wino-x3")[1].strip())
    else:
        print(l.rstrip()[:300])
PY
