"""Kernel trace (rocpd SQLite) of tools/gpu_single_timeline.py -> the LAST rollout: launches, span, sum of kernel durations, and per
launch: start offset, gap to the previous kernel's end, duration, name.  usage: prof_rollout_gaps.py <db> [max lines]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select start, end, {name_col} from kernels order by start").fetchall()
# a rollout starts with the map PointNet: k_pointnet_mfma or k_pointnet_rt launched right after the two pose copies
starts = [i for i, r in enumerate(rows) if "k_pointnet" in r[2] and i + 1 < len(rows) and "k_pointnet" in rows[i + 1][2]]
s = starts[-1]
seg = rows[s:]
t0 = seg[0][0]
tot = sum(r[1] - r[0] for r in seg)
gaps = sum(max(0, seg[i][0] - seg[i - 1][1]) for i in range(1, len(seg)))
print(f"last rollout: {len(seg)} launches, span {(seg[-1][1]-t0)/1e3:.1f} us, kernels {tot/1e3:.1f} us, gaps {gaps/1e3:.1f} us ({gaps/len(seg)/1e3:.2f} us per launch)")
agg = {}
for r in seg:
    n = re.sub(r"\(.*", "", r[2])[:48]
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (r[1] - r[0]) / 1e3
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:48s} x{a[0]:4d} {a[1]:9.1f} us")
mx = int(sys.argv[2]) if len(sys.argv) > 2 else 0
prev = None
for i, r in enumerate(seg[:mx]):
    nm = re.sub(r"\(.*", "", r[2])[:44]
    gap = (r[0] - prev) / 1e3 if prev else 0
    print(f"{(r[0]-t0)/1e3:9.1f} us gap {gap:5.1f} +{(r[1]-r[0])/1e3:7.1f}  {nm}")
    prev = r[1]
