"""Timeline of the drop-in loop from a rocprofv3 --kernel-trace run of `bench.py --mode dropin`: per step the kernels
launched (fused_pass_kernel, tonemap_kernel), their durations and the idle gap between them on the device.
usage: dropin_timeline.py <rocprofv3 output dir>"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:28]))
rows.sort()
fused = [r for r in rows if "fused_pass" in r[2]]
print("kernels in trace: %d, fused_pass_kernel launches: %d" % (len(rows), len(fused)))
if len(fused) > 50:
    steady = fused[len(fused) // 8: len(fused) // 8 + 200]      # inside the timed trace + read-out loop of bench.py --mode dropin
    dur = sorted(e - s for s, e, _ in steady)
    period = sorted(b[0] - a[0] for a, b in zip(steady, steady[1:]))
    print("fused_pass_kernel duration: median %.1f us, p90 %.1f us" % (dur[len(dur) // 2] / 1e3, dur[int(len(dur) * 0.9)] / 1e3))
    print("launch-to-launch period (one step of the loop): median %.1f us" % (period[len(period) // 2] / 1e3))
    # what runs between two fused launches
    i0 = rows.index(steady[10]); i1 = rows.index(steady[11])
    print("one step on the device:")
    prev_end = None
    for s, e, n in rows[i0:i1 + 1]:
        gap = "" if prev_end is None else "  (idle before: %.1f us)" % ((s - prev_end) / 1e3)
        print("   %-28s %8.1f us%s" % (n, (e - s) / 1e3, gap))
        prev_end = e
