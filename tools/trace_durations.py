"""Durations of the launches whose kernel name contains a string, from a rocprofv3 --kernel-trace CSV, in launch order."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if sys.argv[2] in r["Kernel_Name"]]
per = int(sys.argv[3]) if len(sys.argv) > 3 else 25
print(len(d), "launches, us:")
for i in range(0, len(d), per):
    print(" ".join(f"{x:.0f}" for x in d[i:i + per]))
