"""Per-kernel time per molecule at two batch sizes from `rocprofv3 --kernel-trace --stats` kernel_stats CSVs (development tool).

    python tools/kernel_table.py <statsA.csv> <molsA> <statsB.csv> <molsB>
"""
import re
import sys

import pandas as pd


def load(path, mols):
    t = pd.read_csv(path)
    t["k"] = t["Name"].str.extract(r"(k_\w+)")[0]
    t = t[t["k"].notna()]
    g = t.groupby("k").agg(calls=("Calls", "sum"), total=("TotalDurationNs", "sum"))
    g["avg_us"] = g["total"] / g["calls"] / 1e3
    g["ns_per_mol"] = g["total"] / g["calls"] / mols
    return g


def main():
    a, ma, b, mb = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    A, B = load(a, ma), load(b, mb)
    ks = [k for k in B.sort_values("total", ascending=False).index if k in A.index]
    print(f"{'kernel':22s} {'us @' + str(ma):>12s} {'us @' + str(mb):>12s} {'ns/mol @' + str(ma):>14s} {'ns/mol @' + str(mb):>14s} {'ratio':>7s} {'share @' + str(ma):>11s}")
    tot = float((A.loc[ks, "total"]).sum())
    for k in ks:
        print(f"{k:22s} {A.loc[k, 'avg_us']:12.1f} {B.loc[k, 'avg_us']:12.1f} {A.loc[k, 'ns_per_mol']:14.2f} {B.loc[k, 'ns_per_mol']:14.2f} "
              f"{A.loc[k, 'ns_per_mol'] / B.loc[k, 'ns_per_mol']:7.3f} {A.loc[k, 'total'] / tot:11.3f}")
    sa = sum(A.loc[k, "total"] / ma for k in ks)
    sb = sum(B.loc[k, "total"] / mb for k in ks)
    print(f"{'all listed kernels':22s} {'':12s} {'':12s} {'':14s} {'':14s} {sa / sb * (B.loc[ks, 'calls'].max() / A.loc[ks, 'calls'].max()):7.3f}")


if __name__ == "__main__":
    main()
