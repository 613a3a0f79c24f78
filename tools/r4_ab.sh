#!/bin/bash
# round 4: A/B of conv_wino_x3w variants (E2FGVI_LIB builds of tools/r4_variants.py) + per-phase timing
O=gpurun_out/${1:-r4ab}; mkdir -p $O
LAYERS=${2:-encoder.10,encoder.8,decoder.0,decoder.4,encoder.16,conv_offset.6}
for v in "" ${VARIANTS:-8 16 24}; do
  if [ -z "$v" ]; then L=""; tag=prod; else L=tools/probe/libe2fgvi_x3v$v.so; tag=v$v; fi
  E2FGVI_LIB=$L E2FGVI_TUNE_FILE=0 timeout 300 python tools/x3_bench.py $LAYERS 99 > $O/x3_$tag.txt 2>&1
  echo "== $tag"; python - "$O/x3_$tag.txt" <<'PY'
import sys
for l in open(sys.argv[1]):
    if "wino-x3" in l:
        print(l.split("|")[0].strip()[:28], "|", " ".join(x for x in l.split("wino-x3")[1].split("  ") if "w164" in x or "w5132" in x or "w6064" in x))
PY
done
timeout 300 python tools/r4_variants.py timing ${TVARIANTS:-0 16 24} > $O/timing.txt 2>&1; cat $O/timing.txt | grep -v amdgpu.ids
