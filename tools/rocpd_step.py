"""One training step out of a rocprofv3 rocpd database (bench.py --mode train): wall time, union-busy time and, per HIP queue, the kernels
by total duration.  Steps are delimited by k_adamw_ema launches.
    python tools/rocpd_step.py results.db [step index, default: the third from last] [rows per queue]"""
import collections, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end, grid_x, workgroup_x, queue_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "k_adamw_ema" in r[0]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) - 3
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
for j in range(1, len(idx)):
    st = rows[idx[j - 1] + 1: idx[j] + 1]
    ev = sorted((r[1], r[2]) for r in st)
    busy, (cs, ce) = 0, ev[0]
    for s, e in ev[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    qs = collections.Counter(r[5] for r in st)
    print(f"step {j}: {len(st)} launches, wall {(st[-1][2] - st[0][1]) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, per queue "
          + ", ".join(f"{q}: {n} launches {sum(r[2] - r[1] for r in st if r[5] == q) / 1e6:.2f} ms" for q, n in sorted(qs.items())))
st = rows[idx[which - 1] + 1: idx[which] + 1]
for q in sorted(set(r[5] for r in st)):
    agg = collections.defaultdict(lambda: [0, 0])
    for r in st:
        if r[5] == q:
            n = r[0].replace("(anonymous namespace)::", "").replace("void ", "")[:64]
            agg[n][0] += 1
            agg[n][1] += r[2] - r[1]
    print(f"step {which}, queue {q}")
    for n, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:top]:
        print(f"  {n:64s} {c:4d} {t / 1e6:7.3f} ms  {t / c / 1e3:7.1f} us")
