"""Derived per-dispatch table from three rocprofv3 --pmc passes (rocpd sqlite): SQ/GRBM counters, FETCH_SIZE, WRITE_SIZE.

    python tools/pmc_report.py sq.db fetch.db write.db [--json out.json --geom B H W] [--fetch-x2 REGEX] > table.md

Dispatches are joined by launch order (pnpx kernels only).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide (16 B/lane) coalesced streams (MI355X_MICROARCH.md, HBM section): kernels whose short name
matches --fetch-x2 (default: conv_hs -- its loads are 16-byte LDS-DMA) get the doubled column used for traffic totals;
for the others both the raw and the doubled figure are printed."""
import json, re, sqlite3, sys
from collections import OrderedDict

args = sys.argv[1:]
dbs = [a for a in args if a.endswith(".db")]
jpath = args[args.index("--json") + 1] if "--json" in args else None
geom = [int(v) for v in args[args.index("--geom") + 1:args.index("--geom") + 4]] if "--geom" in args else None
x2 = re.compile(args[args.index("--fetch-x2") + 1] if "--fetch-x2" in args else r"conv_hs")
cnt = re.compile(args[args.index("--count") + 1] if "--count" in args else r"conv_hs|conv_first")   # rows in the totals


def short(n):
    m = re.search(r"pnpx::(?:\(anonymous namespace\)::)?([A-Za-z0-9_]+)(<.*>)?", n)
    if not m:
        return None
    t = m.group(2) or ""
    t = re.sub(r"pnpx::", "", t)
    return m.group(1) + (t if len(t) < 60 else t[:57] + "...>")


def load(db):
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, kernel_name, counter_name, sum(value), min(start), max(end) "
                     "from counters_collection group by dispatch_id, counter_name order by dispatch_id").fetchall()
    d = OrderedDict()
    for did, kn, cn, v, s, e in rows:
        sn = short(kn)
        if sn is None:
            continue
        d.setdefault(did, {"name": sn, "us": (e - s) / 1e3})[cn] = v
    return list(d.values())


tabs = [load(d) for d in dbs]
n = min(len(t) for t in tabs)
rows = []
for i in range(n):
    r = dict(tabs[0][i])
    for t in tabs[1:]:
        assert t[i]["name"] == r["name"], (i, t[i]["name"], r["name"])
        r.update({k: v for k, v in t[i].items() if k not in ("name", "us")})
    rows.append(r)

print("| # | kernel | us | clk GHz | MFMA busy % | LDS busy % | bank-conflict % | wave issue % | wait (waitcnt/barrier) % | "
      "FETCH MB | FETCH x2 MB | WRITE MB | HBM TB/s |")
print("|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
tot = {"us": 0.0, "bytes": 0.0, "n": 0}
for i, r in enumerate(rows):
    us = r["us"]
    grbm = r.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                       # summed over the 8 XCDs
    clk = grbm / (us * 1e3) if us else 0.0                            # cycles / ns = GHz
    cyc = grbm if grbm else us * 2.0e3
    mfma = 100.0 * r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc) if cyc else 0.0
    lds = 100.0 * r.get("SQ_LDS_IDX_ACTIVE", 0.0) / (256.0 * cyc) if cyc else 0.0
    conf = 100.0 * r.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(r.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0)
    wc = max(r.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    act = 100.0 * r.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
    wait = 100.0 * r.get("SQ_WAIT_ANY", 0.0) / wc
    f = r.get("FETCH_SIZE", 0.0) * 1024.0
    w = r.get("WRITE_SIZE", 0.0) * 1024.0
    dbl = bool(x2.search(r["name"]))
    b = (2 * f if dbl else f) + w
    print(f"| {i} | {r['name']} | {us:.1f} | {clk:.2f} | {mfma:.1f} | {lds:.1f} | {conf:.1f} | {act:.0f} | {wait:.0f} | "
          f"{f / 1e6:.0f} | {2 * f / 1e6:.0f}{'*' if dbl else ''} | {w / 1e6:.0f} | {b / (us * 1e-6) / 1e12 if us else 0:.2f} |")
    if cnt.search(r["name"]):
        tot["us"] += us
        tot["bytes"] += b
        tot["n"] += 1
print()
print("`*` = doubled FETCH used (16-byte coalesced streams); HBM TB/s = (FETCH [x2 where starred] + WRITE) / duration.")
if tot["n"]:
    print(f"\n{tot['n']} convolution launches ({cnt.pattern}): {tot['us'] / 1e3:.3f} ms, {tot['bytes'] / 1e9:.2f} GB "
          f"=> {tot['bytes'] / tot['n'] / 1e6:.0f} MB per launch, {tot['bytes'] / (tot['us'] * 1e-6) / 1e12:.2f} TB/s.")
    if jpath and geom:
        json.dump({"B": geom[0], "H": geom[1], "W": geom[2], "conv_launches": tot["n"],
                   "hbm_bytes_per_forward": tot["bytes"], "hbm_bytes_per_conv_launch": tot["bytes"] / tot["n"],
                   "conv_us_profiled": tot["us"],
                   "source": "profiles/" + (re.sub(r"_pmc_traffic_fp32\.json$", "_denoiser_pmc_fp32.md", jpath.split("/")[-1])
                                            if jpath.endswith("_fp32.json") else
                                            re.sub(r"_pmc_traffic\.json$", "_denoiser_pmc_hs.md", jpath.split("/")[-1])) +
                             " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH doubled for the kernels matching "
                             f"/{x2.pattern}/ per MI355X_MICROARCH.md)"},
                  open(jpath, "w"), indent=1)
