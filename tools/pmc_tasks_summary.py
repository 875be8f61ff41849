"""Aggregate a per-dispatch PMC table (tools/pmc_report.py output) per kernel: launches, mean duration, mean FETCH /
WRITE and HBM rate.  usage: pmc_tasks_summary.py table.md [exclude-regex]"""
import collections, re, sys
rows = collections.OrderedDict()
skip = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
for line in open(sys.argv[1]):
    c = [x.strip() for x in line.strip().strip("|").split("|")]
    if len(c) < 12 or not c[0].isdigit():
        continue
    if skip and skip.search(c[1]):
        continue
    r = rows.setdefault(c[1], [0, 0.0, 0.0, 0.0])
    r[0] += 1
    r[1] += float(c[2])
    r[2] += float(c[9])
    r[3] += float(c[11])
print("| kernel | launches | mean us | FETCH_SIZE MB | WRITE_SIZE MB | HBM TB/s (raw counters) | HBM TB/s (FETCH x2) |")
print("|---|---:|---:|---:|---:|---:|---:|")
for n, (k, us, f, w) in rows.items():
    print(f"| {n} | {k} | {us / k:.1f} | {f / k:.0f} | {w / k:.0f} | {(f + w) / us:.2f} | {(2 * f + w) / us:.2f} |")
print()
print("FETCH_SIZE / WRITE_SIZE from separate rocprofv3 --pmc passes, joined by dispatch order; MB per launch.  On gfx950 "
      "FETCH_SIZE under-reports wide coalesced (16 B/lane) streams by 2x (MI355X_MICROARCH.md); narrower accesses are "
      "uncalibrated, so both readings are given: the truth lies between them (float2 / float4 streams: nearer x2).")
