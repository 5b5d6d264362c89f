"""Print the kernel timeline (ms) of the last N dispatches of a rocprofv3 --kernel-trace CSV directory."""
import csv, glob, sys

d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-n:]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e6
    e = (int(r["End_Timestamp"]) - t0) / 1e6
    print("%10.3f %10.3f %8.3f q%s %-48s grid=%s lds=%s" % (s, e, e - s, r["Queue_Id"], r["Kernel_Name"][:48], r["Grid_Size_X"], r["LDS_Block_Size"]))
