#!/bin/bash
# the traced replay with and without the propagation split on one box: the steady-state step of the chain
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
E2FGVI_PROP_SPLIT=0 bash tools/profile_graph.sh r05_whole > /dev/null 2>&1
bash tools/profile_graph.sh r05_split > /dev/null 2>&1
python - <<'PY'
import re
for tag in ("whole", "split"):
    rows = []
    for l in open("gpurun_out/profg_r05_%s/timeline.txt" % tag):
        m = re.match(r"\s*([\d.]+)\s+([\d.]+) us\s+(\S+)", l)
        if m:
            rows.append((float(m.group(1)), float(m.group(2)), m.group(3)))
    dcn = [s for s, d, n in rows if n.startswith("mdcn_kernel")]
    steps = [b - a for a, b in zip(dcn, dcn[1:])]
    steady = sorted(steps)[:12]
    t0 = next(s for s, d, n in rows if n.startswith("conv_wino_kernel<1,"))
    t1 = max(s + d for s, d, n in rows if n.startswith("mdcn_kernel"))
    print(tag, "deformable-conv to deformable-conv: median of the 12 shortest of %d steps %.1f us; chain %.0f us (first one-frame launch to the last deformable conv); span %s"
          % (len(steps), steady[len(steady) // 2], t1 - t0, open("gpurun_out/profg_r05_%s/timeline.txt" % tag).read().strip().splitlines()[-1]))
PY
