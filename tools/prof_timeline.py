"""Print a slice of the kernel timeline (start offset, duration, name) from a rocprofv3 rocpd SQLite database."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = "queue_id" if "queue_id" in cols else None
rows = db.execute(f"select start, end, {name_col}" + (f", {qcol}" if qcol else "") + " from kernels order by start").fetchall()
skip, n = int(sys.argv[2]), int(sys.argv[3])
t0 = rows[skip][0]
for r in rows[skip:skip + n]:
    nm = re.sub(r"\(.*", "", r[2])[:48]
    print(f"{(r[0]-t0)/1e3:9.1f} us  +{(r[1]-r[0])/1e3:7.1f}  q{r[3] if qcol else '-'}  {nm}")
