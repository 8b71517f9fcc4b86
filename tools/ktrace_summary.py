"""Per-kernel durations from a rocprofv3 --kernel-trace csv dir: prints the last N dispatches."""
import csv, glob, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[-n]["Start_Timestamp"]) if len(rows) >= n else int(rows[0]["Start_Timestamp"])
for r in rows[-n:]:
    name = r["Kernel_Name"].replace("mi355rt::", "").split("(")[0][:40]
    print("%-42s start %9.1f us  dur %9.1f us  grid %8s wg %4s vgpr %3s lds %6s" % (
        name, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
        r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size")), r.get("VGPR_Count"), r.get("LDS_Block_Size")))
