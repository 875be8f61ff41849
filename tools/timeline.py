"""Timeline of the last forward in a rocprofv3 kernel trace (rocpd sqlite): per kernel start offset, duration, gap to the previous
kernel on the same queue.  usage: timeline.py results.db [n_last]"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 70
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = c.execute(f"select {name_col}, start, end, {qcol or 0} from kernels order by start").fetchall()[-n:]
t0 = rows[0][1]
last = {}
busy = 0
for nm, s, e, q in rows:
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    busy += e - s
    nm = re.sub(r"\s+", " ", nm).replace("void pnpx::", "")[:60]
    print(f"q{q} +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {nm}")
print(f"span {(rows[-1][2] - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us over {len(rows)} kernels; columns: {cols}")
