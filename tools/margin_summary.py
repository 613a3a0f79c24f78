"""Worst measured/allowed ratio per comparison over the margin logs of tools/gpu_suite_soak.sh."""
import collections
import sys

worst = collections.defaultdict(lambda: (0.0, 0.0, 0.0, 0))
for path in sys.argv[1:]:
    for line in open(path):
        parts = line.rstrip("\n").split("\t")
        if len(parts) != 4:
            continue
        what, m, a, r = parts[0], float(parts[1]), float(parts[2]), float(parts[3])
        w = worst[what]
        worst[what] = (max(w[0], r), m if r >= w[0] else w[1], a, w[3] + 1)
rows = sorted(worst.items(), key=lambda kv: -kv[1][0])
print("%d distinct comparisons, %d over 0.5 of their bound, %d over 0.8" %
      (len(rows), sum(1 for _, v in rows if v[0] > 0.5), sum(1 for _, v in rows if v[0] > 0.8)))
for what, (r, m, a, n) in rows[:60]:
    print("%6.3f  measured %.3e  allowed %.3e  n=%d  %s" % (r, m, a, n, what[:150]))
