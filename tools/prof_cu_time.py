"""rocprofv3 kernel trace (rocpd SQLite) -> per-kernel CU-time table: duration x min(workgroups, 256) / 256, i.e. what a
kernel costs a pipeline in which other launches can use the CUs it leaves idle.  usage: prof_cu_time.py <results.db>"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
gx = [c for c in cols if c.lower() in ("grid_size_x", "grid_x", "grid_size")][0]
wx = [c for c in cols if c.lower() in ("workgroup_size_x", "workgroup_x", "workgroup_size")][0]
rows = db.execute(f"select {name_col}, end-start, {gx}, {wx} from kernels").fetchall()
agg = {}
for n, d, g, w in rows:
    n = re.sub(r"\(.*", "", n)[:64]
    wgs = max(1, int(g) // max(1, int(w)))
    a = agg.setdefault(n, [0, 0.0, 0.0, 0])
    a[0] += 1; a[1] += d; a[2] += d * min(wgs, 256) / 256.0; a[3] = max(a[3], wgs)
tot = sum(a[2] for a in agg.values())
print(f"{'kernel':64s} {'calls':>6s} {'time_ms':>9s} {'cu_time_ms':>11s} {'%cu':>6s} {'max_wgs':>8s}")
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print(f"{n:64s} {a[0]:6d} {a[1]/1e6:9.3f} {a[2]/1e6:11.3f} {100*a[2]/tot:6.1f} {a[3]:8d}")
