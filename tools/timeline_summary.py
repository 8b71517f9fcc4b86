"""Timeline of the LAST frame of a rocprofv3 --kernel-trace csv dir: which kernels overlap, how long no
trace kernel is resident, per-stream sequences.  usage: timeline_summary.py dir"""
import csv, glob, sys
d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last frame: from the last-but-one tonemap/copyBuffer marker
res = [i for i, r in enumerate(rows) if "resolve" in r["Kernel_Name"]]
nslices = int(sys.argv[2]) if len(sys.argv) > 2 else 3
last = res[-nslices:]                      # resolves of the last frame
first_resolve_prev = res[-nslices - 1] if len(res) > nslices else -1
frame = rows[first_resolve_prev + 1: last[-1] + 1]
frame = [r for r in frame if "mi355rt" in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in frame); t1 = max(int(r["End_Timestamp"]) for r in frame)
print("frame kernels %d  span %.2f ms" % (len(frame), (t1 - t0) / 1e6))
ev = []
for r in frame:
    k = "T" if "trace_kernel" in r["Kernel_Name"] else ("S" if "shade" in r["Kernel_Name"] else ("C" if "confirm" in r["Kernel_Name"] else "R"))
    ev.append((int(r["Start_Timestamp"]), 1, k)); ev.append((int(r["End_Timestamp"]), -1, k))
ev.sort()
cnt = {"T": 0, "S": 0, "C": 0, "R": 0}; prev = t0; acc = {}
for t, dlt, k in ev:
    key = "T%d S%d C%d R%d" % (cnt["T"], cnt["S"], cnt["C"], cnt["R"])
    acc[key] = acc.get(key, 0) + (t - prev); prev = t
    cnt[k] += dlt
for key, v in sorted(acc.items(), key=lambda x: -x[1])[:12]:
    print("  resident %-12s %7.2f ms" % (key, v / 1e6))
qcol = "Queue_Id" if "Queue_Id" in frame[0] else None
for r in frame:
    name = r["Kernel_Name"].replace("mi355rt::", "").split("(")[0].replace("void ", "")[:26]
    print("  q%-3s %-26s start %8.2f ms  dur %7.2f ms" % (r.get(qcol, "?") if qcol else "?", name, (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
