"""Throughput vs resident micro-batch: runs bench.py at several --mols and collects the JSON lines.

    python tools/batch_curve.py [--out profiles/r03_throughput_vs_batch.jsonl] [--mols 64,256,...]
Each point times 4 bench steps (200 denoise iterations) after 1 warm-up step; molecules/sec is normalised to complete
1000-step samplings exactly as the headline line is.  Development tool; bench.py is the contract.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_throughput_vs_batch.jsonl"))
    ap.add_argument("--mols", default="64,128,256,512,1024,1250,2048,4096,8192,10000")
    ap.add_argument("--extra", default="", help="extra bench.py flags, e.g. --graph")
    args = ap.parse_args()
    rows = []
    for m in [int(x) for x in args.mols.split(",")]:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mols", str(m), "--steps", "4", "--warmup", "1",
               "--no-cpu-baseline", "--mode", "resident", "--no-live-traffic"] + args.extra.split()
        out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(m, "FAILED", out.stderr[-400:], flush=True)
            continue
        j = json.loads(line[-1])
        row = {"molecules_per_gpu": m, "molecules_per_sec": j["value"], "ms_per_denoise_iteration": j["ms_per_step"] / j["config"]["denoise_iterations_per_step"],
               "us_per_molecule_iteration": j["ms_per_step"] / j["config"]["denoise_iterations_per_step"] * 1e3 / m,
               "equi_frac": (j["roofline"] or {}).get("frac"), "equi_vs_fp32_mfma_peak": (j["roofline"] or {}).get("vs_fp32_mfma_peak"),
               "whole_path_vs_fp32_mfma_peak": j["whole_path"].get("vs_fp32_mfma_peak"), "flags": args.extra}
        rows.append(row)
        print(json.dumps(row), flush=True)
    with open(args.out, "a") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
