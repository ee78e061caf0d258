"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (development tool)."""
import glob
import sys

import pandas as pd

frames = []
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        frames.append(pd.read_csv(f))
d = pd.concat(frames)
d["k"] = d["Kernel_Name"].str.extract(r"(k_\w+)")
d = d[d["k"].notna()]
d["dur"] = d["End_Timestamp"] - d["Start_Timestamp"]
g = d.groupby(["k", "Counter_Name"])["Counter_Value"].mean().unstack()
dur = d.groupby("k")["dur"].mean()
calls = d.groupby("k")["Dispatch_Id"].nunique()
order = dur.sort_values(ascending=False).index
for k in order:
    r = g.loc[k]
    out = f"{k:18s} n={calls[k]:3d} dur {dur[k] / 1e3:8.1f}us"
    if "SQ_WAVE_CYCLES" in r and r.SQ_WAVE_CYCLES > 0:
        w = r.SQ_WAVE_CYCLES
        out += f" WAIT_ANY {r.SQ_WAIT_ANY / w:4.2f} WAIT_INST {r.SQ_WAIT_INST_ANY / w:4.2f} ACTIVE {r.SQ_ACTIVE_INST_ANY / w:4.2f}"
        out += f" MFMA_BUSY {r.SQ_VALU_MFMA_BUSY_CYCLES / 1e6:7.1f}M mfma_util@1024simd {r.SQ_VALU_MFMA_BUSY_CYCLES / 1024 / (dur[k] * 2.4):4.2f}"
        out += f" LDS_CONF/ACT {r.SQ_LDS_BANK_CONFLICT / max(1, r.SQ_LDS_IDX_ACTIVE):4.2f}"
    if "GRBM_GUI_ACTIVE" in r:
        out += f" clk {r.GRBM_GUI_ACTIVE / 8 / dur[k]:4.2f}GHz"
    if "FETCH_SIZE" in r:
        out += f" FETCH(x2) {2 * r.FETCH_SIZE / 1024:8.1f}MB"   # kB units; gfx950 reports half of wide streams (guide)
    if "WRITE_SIZE" in r:
        out += f" WRITE {r.WRITE_SIZE / 1024:8.1f}MB"
    if "TCC_HIT_sum" in r:
        out += f" L2hit {r.TCC_HIT_sum / max(1, r.TCC_HIT_sum + r.TCC_MISS_sum):4.2f}"
    print(out)
