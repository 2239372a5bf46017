"""Per-kernel statistics (the table `rocprofv3 --kernel-trace --stats` prints) from a rocprofv3 rocpd SQLite database.
usage: rocpd_stats.py results.db [out.csv]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute("select %s, start, end from kernels" % name_col).fetchall()
agg = {}
for name, s, e in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0]
    a = agg.setdefault(name, [0, 0, 1 << 62, 0])
    d = e - s
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values()) or 1
lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append('"%s",%d,%d,%.1f,%.2f,%d,%d' % (name, a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot, a[2], a[3]))
out = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
print(out)
