"""Print the kernel timeline of the last denoiser forward found in a rocprofv3 kernel_trace csv (development aid)."""
import csv, glob, sys, collections
f = (glob.glob(sys.argv[1] + '/*/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*kernel_trace.csv'))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'graph_scan' in r['Kernel_Name']]
s, e = idx[-2], idx[-1]
t0 = int(rows[s]['Start_Timestamp']); prev = t0
agg = collections.OrderedDict()
for r in rows[s:e]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('void ', '').replace('scgg::', '').split('(')[0][:40]
    if len(sys.argv) > 2:
        print(f"{(st-t0)/1e3:8.1f} gap {(st-prev)/1e3:6.1f} dur {(en-st)/1e3:7.1f} {name}")
    d = agg.setdefault(name, [0, 0.0]); d[0] += 1; d[1] += (en - st) / 1e3
    prev = en
print(f"forward span {(prev-t0)/1e3:.1f} us")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:9.1f} us {c:4d} x {k}")
