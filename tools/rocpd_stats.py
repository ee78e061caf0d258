"""Per-kernel summary of a rocprofv3 rocpd database (the default output format of ROCm 7.x): calls, total and mean duration, share.
    python tools/rocpd_stats.py results.db [steps]   -> CSV on stdout, like `--stats`' kernel_stats.csv (per-step columns when steps given)"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else cols[0]
rows = cur.execute(f"select {name_col}, count(*), sum(end - start), min(end - start), max(end - start) from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","CallsPerStep","MsPerStep"')
for n, c, t, mn, mx in rows:
    print(f'"{n}",{c},{t},{t / c:.1f},{100.0 * t / tot:.2f},{mn},{mx},{c / steps:.1f},{t / steps / 1e6:.4f}')
print(f'"TOTAL",{sum(r[1] for r in rows)},{tot},,100.0,,,{sum(r[1] for r in rows) / steps:.1f},{tot / steps / 1e6:.4f}')
