"""Print the GPU timeline of ONE planning cycle from a rocprofv3 --kernel-trace (+ --memory-copy-trace) csv pair:
start offset, duration and the idle gap before every dispatch / copy of the last complete cycle (a cycle = from one
k_ilqr dispatch's end to the next one's end)."""
import csv, sys, glob, os
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48]))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[:24]))
rows.sort()
il = [i for i, r in enumerate(rows) if "k_ilqr" in r[2]]
if len(il) < 3:
    print("too few cycles"); sys.exit(0)
a, b = il[-3], il[-2]
t0 = rows[a][1]
prev = t0
busy = 0
print("%9s %8s %8s  %s" % ("start_us", "dur_us", "gap_us", "what"))
for s, e, n in rows[a + 1:b + 1]:
    print("%9.1f %8.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, n))
    busy += e - s
    prev = max(prev, e)
print("cycle %.1f us, GPU busy %.1f us" % ((rows[b][1] - t0) / 1e3, busy / 1e3))
