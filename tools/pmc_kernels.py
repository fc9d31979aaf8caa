"""Launches the dominant kernels a few times at the BASELINE shape (for rocprofv3 --pmc passes).
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python tools/pmc_kernels.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out -- python tools/pmc_kernels.py
then `python tools/pmc_kernels.py --summarise fetch.csv write.csv` folds the two counter_collection.csv files into
profiles/r01_pmc_traffic.json (per-launch averages; HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB, the gfx950
correction of MI355X_MICROARCH.md's HBM section)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KEYS = {"conv3_bf16x3_kernel": "conv3_bf16x3", "wgrad3_bf16x3_kernel": "wgrad3_bf16x3", "conv3_mfma_kernel<1, 1, 8, 4, 8, 16, true>": "conv3_mfma",
        "wgrad3_kernel<1, 16>": "wgrad3"}
ALGO_BYTES = 2 * 32 * 48 * 136 * 240 * 4 + 27 * 32 * 32 * 4   # read x + write y (or read x, dy) + weights


def summarise(fetch_csv, write_csv):
    acc = {}
    for path, col in ((fetch_csv, "FETCH_SIZE"), (write_csv, "WRITE_SIZE")):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != col:
                continue
            for pat, key in KEYS.items():
                if pat in r["Kernel_Name"]:
                    acc.setdefault(key, {}).setdefault(col, []).append(float(r["Counter_Value"]))
                    acc[key]["kernel"] = r["Kernel_Name"].split("(")[0]
    out = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
    for key, v in acc.items():
        f = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"])
        w = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
        out[key] = {"FETCH_SIZE_KiB": round(f, 2), "WRITE_SIZE_KiB": round(w, 2),
                    "hbm_bytes_per_launch": (2 * f + w) * 1024, "algorithmic_bytes": ALGO_BYTES, "kernel": v["kernel"],
                    "launches": len(v["FETCH_SIZE"])}
    json.dump(out, open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2], sys.argv[3])
    else:
        import torch
        import bench
        print(bench.kernel_roofline(torch.device("cuda", 0)))
