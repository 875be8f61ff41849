"""Per-dispatch PMC table from rocprofv3 rocpd databases (pnpx kernels only, short names).
usage: rocpd_pmc.py a.db [b.db ...]   -- counters of the same dispatch order are joined across files."""
import re, sqlite3, sys
from collections import OrderedDict, defaultdict

def short(n):
    m = re.search(r"pnpx::(?:\(anonymous namespace\)::)?([A-Za-z0-9_]+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else None

tables = []
for db in sys.argv[1:]:
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, kernel_name, counter_name, sum(value), min(start), max(end), max(grid_size) "
                     "from counters_collection group by dispatch_id, counter_name order by dispatch_id").fetchall()
    d = OrderedDict()
    for did, kn, cn, v, s, e, g in rows:
        sn = short(kn)
        if sn is None:
            continue
        d.setdefault(did, {"name": sn, "dur_us": (e - s) / 1e3, "grid": g})[cn] = v
    tables.append(list(d.values()))
n = min(len(t) for t in tables)
merged = []
for i in range(n):
    r = dict(tables[0][i])
    for t in tables[1:]:
        assert t[i]["name"] == r["name"], (t[i]["name"], r["name"])
        for k, v in t[i].items():
            if k not in ("name", "dur_us", "grid"):
                r[k] = v
    merged.append(r)
keys = [k for k in merged[0] if k not in ("name", "dur_us", "grid")] if merged else []
allkeys = []
for r in merged:
    for k in r:
        if k not in ("name", "dur_us", "grid") and k not in allkeys:
            allkeys.append(k)
print("| # | kernel | grid | us | " + " | ".join(allkeys) + " |")
print("|" + "---|" * (4 + len(allkeys)))
for i, r in enumerate(merged):
    print(f"| {i} | {r['name']} | {r['grid']} | {r['dur_us']:.1f} | " + " | ".join(f"{r.get(k, 0):.4g}" for k in allkeys) + " |")
