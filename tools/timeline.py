"""Print the ordered kernel timeline of the last forward in a rocprofv3 kernel-trace CSV.
    python tools/timeline.py gpurun_out/prof_<tag>/prof_kernel_trace.csv [min_us]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if re.search(r"nchw_(small_)?to_nhwc", r["Kernel_Name"])]
# a forward = from the first resize before the mark (flows run on the side stream) to the next forward's start
seg = rows[marks[-2] - 1:marks[-1] - 1]
t0 = int(seg[0]["Start_Timestamp"])
def short(n):
    m = re.search(r"(conv_igemm_kernel<[^>]*>|mdcn_kernel<[^>]*>|focal_attn_kernel<[^>]*>|[a-z_0-9]+_kernel(<[^>]*>)?)", n)
    return m.group(1) if m else n[:40]
tot = 0
for r in seg:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    if d >= min_us:
        print("%9.1f  %8.1f us  %-48s grid %8s wg %4s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, short(r["Kernel_Name"]),
                                                         r["Grid_Size_X"], r["Workgroup_Size_X"]))
print("kernels %d, sum %.3f ms, span %.3f ms" % (len(seg), tot / 1e3, (int(seg[-1]["End_Timestamp"]) - t0) / 1e6))
