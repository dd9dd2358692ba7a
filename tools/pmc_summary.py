"""Per-kernel summary of one rocprofv3 --pmc pass (rocpd SQLite)."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); name = sys.argv[2]
rows = db.execute("select kernel_name, grid_size, count(*), avg(value), min(value), max(value) from counters_collection "
                  "where counter_name=? group by kernel_name, grid_size order by avg(value) desc", (name,)).fetchall()
print(f"# {name} per dispatch (rocprofv3 units: KB), by kernel and grid size")
print(f"{'kernel':44s} {'grid':>9s} {'calls':>6s} {'avg':>12s} {'min':>12s} {'max':>12s}")
for k, g, c, a, mn, mx in rows[:16]:
    print(f"{re.sub(r'[(].*', '', k)[:44]:44s} {g:9d} {c:6d} {a:12.1f} {mn:12.1f} {mx:12.1f}")
