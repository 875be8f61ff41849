"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace: per-kernel calls / total / avg / min / max.

    python tools/rocpd_stats.py gpurun_out/prof_r1/bench_results.db > profiles/r1_bench_kernel_stats.md
"""
import re
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
rows = c.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                 f"from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# rocprofv3 --kernel-trace --stats summary ({db})\n")
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---:|---:|---:|---:|---:|---:|")
for n, k, s, a, mn, mx in rows:
    n = re.sub(r"\s+", " ", n)
    if len(n) > 110:
        n = n[:107] + "..."
    print(f"| `{n}` | {k} | {s / 1e6:.3f} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * s / tot:.2f} |")
print(f"\ntotal GPU kernel time: {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
