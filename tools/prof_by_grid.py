"""rocprofv3 kernel trace (rocpd SQLite) -> per (kernel, workgroups) table: calls, mean duration, CU-time.  usage: prof_by_grid.py <results.db> [name filter]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
gx = [c for c in cols if c.lower() in ("grid_size_x", "grid_x", "grid_size")][0]
wx = [c for c in cols if c.lower() in ("workgroup_size_x", "workgroup_x", "workgroup_size")][0]
agg = {}
for n, d, g, w in db.execute(f"select {name_col}, end-start, {gx}, {wx} from kernels"):
    n = re.sub(r"\(.*", "", n)[:56]
    if flt not in n: continue
    wgs = max(1, int(g) // max(1, int(w)))
    a = agg.setdefault((n, wgs, int(w)), [0, 0.0])
    a[0] += 1; a[1] += d
print(f"{'kernel':56s} {'wgs':>6s} {'thr':>4s} {'calls':>6s} {'mean_us':>9s} {'cu_ms':>8s}")
for (n, wgs, w), a in sorted(agg.items(), key=lambda kv: -kv[1][1] * min(kv[0][1], 256)):
    print(f"{n:56s} {wgs:6d} {w:4d} {a[0]:6d} {a[1]/a[0]/1e3:9.1f} {a[1]*min(wgs,256)/256/1e6:8.3f}")
