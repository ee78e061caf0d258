"""Write profiles/rNN_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

    python tools/pmc_traffic_json.py <fetch_dir> <write_dir> <molecules_per_launch> <out.json>
HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in kB and gfx950 reports half of wide
fetch streams (MI355X_MICROARCH.md, HBM / rocprofv3 section).  Development tool.
"""
import glob
import json
import sys

import pandas as pd


def per_kernel(d, counter):
    frames = [pd.read_csv(f) for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True)]
    t = pd.concat(frames)
    t = t[t["Counter_Name"] == counter]
    t = t.assign(k=t["Kernel_Name"].str.extract(r"(k_\w+)")[0])
    t = t[t["k"].notna()]
    return t.groupby("k")["Counter_Value"].mean()


def main():
    fetch_dir, write_dir, mols, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    f, w = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    rec = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 1 --warmup 0 "
                    f"--denoise-steps 8 --steps-per-pass 1 --no-cpu-baseline` at {mols} molecules per launch; bytes = (2*FETCH_SIZE + "
                    "WRITE_SIZE)*1024, the x2 is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (HBM section)"}
    for k in sorted(set(f.index) & set(w.index)):
        b = (2.0 * f[k] + w[k]) * 1024.0
        rec[k] = {"FETCH_SIZE_kB": float(f[k]), "WRITE_SIZE_kB": float(w[k]), "hbm_bytes_per_launch": b,
                  "bytes_per_launch_per_molecule": b / mols}
    json.dump(rec, open(out, "w"), indent=1)
    print("wrote", out, "kernels:", len(rec) - 1)


if __name__ == "__main__":
    main()
