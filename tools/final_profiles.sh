#!/bin/bash
# rocprofv3 kernel-trace summaries + PMC HBM traffic of both benchmark configurations, in one GPU-box call (about 6 minutes)
HQ="--model e2fgvi_hq --hw 720x1296 --precision bf16"
bash tools/profile.sh r02_fp32 2>&1 | tail -3
bash tools/profile.sh r02_hq720_bf16 $HQ 2>&1 | tail -3
bash tools/pmc.sh r02 2>&1 | tail -3
bash tools/pmc.sh r02_hq720_bf16 $HQ 2>&1 | tail -3
