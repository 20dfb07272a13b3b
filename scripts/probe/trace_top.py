"""Aggregate a rocprofv3 kernel trace: for kernels whose name contains a pattern, the distribution of durations."""
import csv, sys, glob, collections
pat = sys.argv[2]
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = []
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
d.sort(reverse=True)
print("calls", len(d), "total us", sum(d))
b = collections.Counter()
for x in d:
    b[int(x // 10) * 10] += x
for k in sorted(b, reverse=True)[:25]:
    print(f"  {k:5d}-{k+10:5d} us: total {b[k]:9.1f} us, calls {sum(1 for x in d if k <= x < k + 10)}")
