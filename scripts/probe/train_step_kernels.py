"""Aggregate one training step out of a rocprofv3 results database (kernel trace of scripts/bench_train.py): launches, device
time, and the top kernels.  Usage: python scripts/probe/train_step_kernels.py <dir with *_results.db> [top]"""
import sqlite3, collections, glob, sys
db = glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "eigen_features_tridiag" in r[0]]
step = rows[idx[-2]:idx[-1]]
print("launches", len(step), "device ms", round(sum(r[2] - r[1] for r in step) / 1e6, 3))
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in step:
    agg[n][0] += 1; agg[n][1] += (e - s) / 1e3
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{v[1] / 1e3:7.3f} ms {v[0]:5d} calls {v[1] / v[0]:8.1f} us  {n[:110]}")
