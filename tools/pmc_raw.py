"""Per-kernel means of every counter found in rocprofv3 --pmc output directories (development tool)."""
import glob
import sys

import pandas as pd

frames = []
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        frames.append(pd.read_csv(f))
d = pd.concat(frames)
d["k"] = d["Kernel_Name"].str.extract(r"(k_\w+)")
d = d[d["k"].notna()]
d["dur"] = d["End_Timestamp"] - d["Start_Timestamp"]
g = d.groupby(["k", "Counter_Name"])["Counter_Value"].mean().unstack()
g["dur_us"] = d.groupby("k")["dur"].mean() / 1e3
g = g.sort_values("dur_us", ascending=False)
pd.set_option("display.width", 250, "display.max_columns", 40, "display.float_format", lambda v: f"{v:.4g}")
cols = [c for c in g.columns if c != "dur_us"]
if "SQ_WAVE_CYCLES" in g.columns:
    for c in cols:
        if c.startswith("SQ_") and c not in ("SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_BUSY_CYCLES") and not c.startswith("SQ_INSTS"):
            g[c + "/wc"] = g[c] / g["SQ_WAVE_CYCLES"]
print(g.head(12).T.to_string())
