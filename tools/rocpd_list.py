"""List the last N kernel dispatches of a rocprofv3 rocpd database in launch order:  python tools/rocpd_list.py db N"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
rows = c.execute(f"select {name_col}, start, end from kernels order by start").fetchall()[-n:]
t0 = rows[0][1]
for name, s, e in rows:
    name = re.sub(r"\s+", " ", name)
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {name[:100]}")
print(f"span {(rows[-1][2] - t0) / 1e3:.1f} us, sum {sum(e - s for _, s, e in rows) / 1e3:.1f} us")
