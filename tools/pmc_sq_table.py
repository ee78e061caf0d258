"""Per-kernel SQ counter table from rocprofv3 --pmc passes (development tool; the tables under profiles/rNN_pmc_sq_*.txt).

    python tools/pmc_sq_table.py <dirA> <dirB> <dirC> [--min-us 20] > profiles/r05_pmc_sq_xxx.txt
Passes (one rocprofv3 run each, `--kernel-trace --pmc <list>`):
  A: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
  B: SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
  C: SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); LDS confl = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;
wait_any = SQ_WAIT_ANY / SQ_WAVE_CYCLES; valu_act = SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES-normalised wave cycles.
"""
import glob
import sys

import pandas as pd


def load(d):
    frames = [pd.read_csv(f) for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True)]
    t = pd.concat(frames)
    t = t.assign(k=t["Kernel_Name"].str.extract(r"(k_\w+)")[0])
    t = t[t["k"].notna()]
    t["dur"] = t["End_Timestamp"] - t["Start_Timestamp"]
    g = t.groupby(["k", "Counter_Name"])["Counter_Value"].mean().unstack()
    g["dur_us"] = t.groupby("k")["dur"].mean() / 1e3
    g["launches"] = t.groupby("k")["Dispatch_Id"].nunique()
    return g


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    min_us = float(sys.argv[sys.argv.index("--min-us") + 1]) if "--min-us" in sys.argv else 20.0
    A, B, Cc = (load(d) for d in args[:3])
    ks = [k for k in Cc.sort_values("dur_us", ascending=False).index if k in A.index and k in B.index and Cc.loc[k, "dur_us"] >= min_us]
    print(f"{'kernel':26s} {'n':>5s} {'dur_us':>8s} {'MFMA insts':>11s} {'VALU insts':>11s} {'valu/mfma':>9s} {'mfma_busy':>9s} {'LDS confl':>9s} "
          f"{'wait_any':>8s} {'valu_act':>8s} {'LDS insts':>10s} {'VMEM rd':>10s} {'VMEM wr':>10s}")
    for k in ks:
        a, b, c = A.loc[k], B.loc[k], Cc.loc[k]
        mf = c.get("SQ_INSTS_MFMA", 0.0)
        busy = b["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1.0, c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0) * (c["dur_us"] / b["dur_us"])
        print(f"{k:26s} {int(c['launches']):5d} {c['dur_us']:8.1f} {mf:11.3e} {c['SQ_INSTS_VALU']:11.3e} "
              f"{(c['SQ_INSTS_VALU'] / mf if mf else float('nan')):9.2f} {busy:9.3f} "
              f"{b['SQ_LDS_BANK_CONFLICT'] / max(1.0, b['SQ_LDS_IDX_ACTIVE']):9.3f} {a['SQ_WAIT_ANY'] / a['SQ_WAVE_CYCLES']:8.2f} "
              f"{a['SQ_ACTIVE_INST_VALU'] / a['SQ_WAVE_CYCLES']:8.2f} {c['SQ_INSTS_LDS']:10.3e} {c['SQ_INSTS_VMEM_RD']:10.3e} {c['SQ_INSTS_VMEM_WR']:10.3e}")


if __name__ == "__main__":
    main()
