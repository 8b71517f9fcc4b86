"""Sum rocprofv3 --pmc counter CSVs per kernel name. usage: pmc_summary.py dir [substr]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "trace_round"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub not in k: continue
        short = ("P:" if "<true" in k else "S:") + k.split("(")[0].split("::")[-1][:24]
        acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (short, r["Dispatch_Id"])
        if key not in seen: seen.add(key); calls[short] += 1
for k in sorted(acc):
    print(k, "dispatches", calls[k])
    for c in sorted(acc[k]): print("   %-36s %.4g" % (c, acc[k][c]))
