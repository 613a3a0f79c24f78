#!/bin/bash
# rocprofv3 kernel-trace summaries + PMC HBM traffic of the benchmark configurations, in one GPU-box call
#   bash tools/final_profiles.sh [fp32|bf16|both]
WHAT=${1:-both}
HQ="--model e2fgvi_hq --hw 720x1296 --precision bf16"
if [ $WHAT != bf16 ]; then bash tools/profile.sh r02_fp32 2>&1 | tail -3; bash tools/pmc.sh r02 2>&1 | tail -3; fi
if [ $WHAT != fp32 ]; then bash tools/profile.sh r02_hq720_bf16 $HQ 2>&1 | tail -3; bash tools/pmc.sh r02_hq720_bf16 $HQ 2>&1 | tail -3; fi
