"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) as a per-kernel table."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
span = db.execute("select min(start), max(end) from kernels").fetchone()
print(f"# {sys.argv[1]}: {sum(r[1] for r in rows)} dispatches, kernel time {tot/1e6:.3f} ms, first->last {(span[1]-span[0])/1e6:.3f} ms")
print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
for n, c, s, a, mn, mx in rows:
    n = re.sub(r"\(.*", "", n)[:70]
    print(f"{n:70s} {c:7d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.1f}")
