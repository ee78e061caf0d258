"""Timeline view of a rocprofv3 --kernel-trace of the training bench (development tool): per optimizer step, how long the GPU ran nothing,
exactly one kernel, or several at once; per queue busy time; and the kernels that account for the time in which they ran ALONE (the
critical path of a multi-stream step is made of those and of the idle gaps).

    python tools/trace_timeline.py <dir with *_kernel_trace.csv> [--skip 3]
Steps are cut at the k_adamw launches (one per optimizer step); the first --skip steps (warm-up, instrumented step) are ignored."""
import glob
import sys

import pandas as pd


def main():
    d = sys.argv[1]
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 3
    t = pd.concat(pd.read_csv(f) for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True))
    t = t.sort_values("Start_Timestamp").reset_index(drop=True)
    t["name"] = t["Kernel_Name"].str.replace(r"\(anonymous namespace\)::", "", regex=True).str.replace(r"^void ", "", regex=True).str.slice(0, 44)
    cuts = t.index[t["Kernel_Name"].str.contains("k_adamw")].tolist()
    cuts = cuts[skip:]
    if len(cuts) < 2:
        print("not enough steps")
        return
    lo, hi = t.loc[cuts[0], "End_Timestamp"], t.loc[cuts[-1], "End_Timestamp"]
    w = t[(t["Start_Timestamp"] >= lo) & (t["End_Timestamp"] <= hi)]
    nsteps = len(cuts) - 1
    ev = []
    for i, r in enumerate(w.itertuples()):
        ev.append((r.Start_Timestamp, 1, i))
        ev.append((r.End_Timestamp, -1, i))
    ev.sort()
    names = w["name"].tolist()
    running = set()
    last = lo
    idle = one = multi = 0
    alone = {}
    for ts, kind, i in ev:
        dt = ts - last
        if dt > 0:
            if len(running) == 0:
                idle += dt
            elif len(running) == 1:
                one += dt
                k = names[next(iter(running))]
                alone[k] = alone.get(k, 0) + dt
            else:
                multi += dt
        last = ts
        if kind == 1:
            running.add(i)
        else:
            running.discard(i)
    idle += hi - last
    wall = hi - lo
    print(f"steps {nsteps}: wall {wall / nsteps / 1e6:.3f} ms/step = idle {idle / nsteps / 1e6:.3f} + one kernel {one / nsteps / 1e6:.3f} + several {multi / nsteps / 1e6:.3f}")
    print(f"launches per step {len(w) / nsteps:.0f}; kernel time per step {(w['End_Timestamp'] - w['Start_Timestamp']).sum() / nsteps / 1e6:.3f} ms")
    qcol = "Queue_Id" if "Queue_Id" in w.columns else None
    if qcol:
        g = w.assign(dur=w["End_Timestamp"] - w["Start_Timestamp"]).groupby(qcol)["dur"].agg(["sum", "count"])
        for q, r in g.iterrows():
            print(f"  queue {q}: busy {r['sum'] / nsteps / 1e6:.3f} ms/step, {r['count'] / nsteps:.0f} launches/step")
    print("time per step in which the kernel ran ALONE (us), top 25:")
    for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:25]:
        tot = w[w["name"] == k]
        print(f"  {k:46s} alone {v / nsteps / 1e3:8.1f}   total {(tot['End_Timestamp'] - tot['Start_Timestamp']).sum() / nsteps / 1e3:8.1f}   calls {len(tot) / nsteps:6.1f}")
    # phases of a step, cut at the first launch of marker kernels (in launch order): SpecFormer forward .. DMT forward(s) .. loss + DMT
    # backward .. SpecFormer backward .. optimizer
    marks = [("spec fwd + batch preparation", None), ("DMT forward(s)", "k_time_feat_fwd"), ("loss + DMT backward", "k_loss"),
             ("SpecFormer backward", "k_sfa_bwd"), ("clip + optimizer", "k_sumsq_partial")]
    tot = {m[0]: 0.0 for m in marks}
    per = {m[0]: {} for m in marks}
    for a, b in zip(cuts[:-1], cuts[1:]):
        seg = t.loc[a + 1:b]
        t_begin = t.loc[a, "End_Timestamp"]
        bounds = []
        for name, key in marks:
            if key is None:
                bounds.append(t_begin)
            else:
                hit = seg[seg["Kernel_Name"].str.contains(key)]
                bounds.append(hit["Start_Timestamp"].iloc[0] if len(hit) else bounds[-1])
        bounds.append(t.loc[b, "End_Timestamp"])
        for (name, _), lo_, hi_ in zip(marks, bounds[:-1], bounds[1:]):
            tot[name] += max(0, hi_ - lo_)
            inside = seg[(seg["Start_Timestamp"] >= lo_) & (seg["Start_Timestamp"] < hi_)]
            for k, dur in zip(inside["name"], inside["End_Timestamp"] - inside["Start_Timestamp"]):
                e = per[name].setdefault(k, [0, 0.0])
                e[0] += 1
                e[1] += dur
    print("phases (ms per step, by the first launch of the next phase's marker kernel):")
    for name, v in tot.items():
        print(f"  {name:32s} {v / nsteps / 1e6:7.3f}")
    if "--phases" in sys.argv:
        for name in tot:
            print(f"kernels launched in phase '{name}' (us per step, calls per step), top 14:")
            for k, (cnt, dur) in sorted(per[name].items(), key=lambda kv: -kv[1][1])[:14]:
                print(f"    {k:46s} {dur / nsteps / 1e3:8.1f}  {cnt / nsteps:6.1f}")
    # idle gaps by the kernel that follows them
    w2 = w.sort_values("Start_Timestamp")
    ends = w2["End_Timestamp"].cummax().shift(1)
    gap = (w2["Start_Timestamp"] - ends).clip(lower=0)
    gg = pd.DataFrame({"name": w2["name"], "gap": gap}).groupby("name")["gap"].sum().sort_values(ascending=False)
    print("idle time per step in FRONT of a kernel (us), top 12:")
    for k, v in gg.head(12).items():
        print(f"  {k:46s} {v / nsteps / 1e3:8.1f}")


if __name__ == "__main__":
    main()
