"""PMC evidence for the dominant kernels at the BASELINE shape (companion of bench.py's `roofline.traffic`).

Collect (on the GPU box; separate --pmc passes -- FETCH_SIZE and WRITE_SIZE do not fit one pass, and gpurun refuses
--pmc together with the trace domains):
    tools/pmc_collect.sh          # three rocprofv3 passes over this script -> gpurun_out/pmc_r03c_{fetch,write,sq}
then -- in the development container, where the snapshot's commit is known (the GPU box has no .git) -- fold them into
profiles/r03_pmc_traffic.json (the file bench.py reads), with the commit they were taken at:
    python tools/pmc_kernels.py --summarise gpurun_out/pmc_r03c_fetch gpurun_out/pmc_r03c_write gpurun_out/pmc_r03c_sq

HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB: gfx950's FETCH_SIZE tallies 128-byte requests at 64 bytes
(MI355X_MICROARCH.md, HBM section; exact for 16-byte-per-lane streams, other access widths are uncalibrated).
MFMA utilisation (`mfma_busy_frac`) = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles =
GRBM_GUI_ACTIVE / 8 XCDs: the share of matrix-pipe cycles that are busy AT THE CLOCK THE CHIP ACTUALLY HELD."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

V4 = 48 * 136 * 240
KEYS = {   # substring of the kernel name -> (key, algorithmic bytes per launch)
    "::conv3_f16x2_kernel<true, false, 0, true>": ("conv3_f16x2_px2", 2 * 32 * V4 * 4 + 27 * 32 * 32 * 4),
    "::conv3_f16x2_kernel<true, false, 0, false>": ("conv3_f16x2", 2 * 32 * V4 * 4 + 27 * 32 * 32 * 4),
    "wgrad3_f16x2_kernel<true, true>": ("wgrad3_f16x2_px2", 2 * 32 * V4 * 4 + 27 * 32 * 32 * 4),
    "wgrad3_f16x2_kernel<false, false>": ("wgrad3_f16x2", 2 * 32 * V4 * 4 + 27 * 32 * 32 * 4),
    "wgrad3s2_f16x2_kernel": ("wgrad3s2_f16x2", 4 * (32 * V4 + 64 * V4 // 8) + 27 * 32 * 64 * 4),
    "conv3s2_f16x2_kernel": ("conv3s2_f16x2", 4 * (32 * V4 + 64 * V4 // 8) + 27 * 32 * 64 * 4),
    "::conv3_bf16x3_kernel": ("conv3_bf16x3", 2 * 32 * V4 * 4 + 27 * 32 * 32 * 4),
    "wgrad3_bf16x3_kernel": ("wgrad3_bf16x3", 2 * 32 * V4 * 4 + 27 * 32 * 32 * 4),
    "conv3_mfma_kernel<1, 1, 8, 4, 8, 16, true": ("conv3_mfma", 2 * 32 * V4 * 4 + 27 * 32 * 32 * 4),
    "wgrad3_kernel<1, 16>": ("wgrad3", 2 * 32 * V4 * 4 + 27 * 32 * 32 * 4),
    "conv3_lp_kernelIDF16bLb0ELb0ELb1ELi1": ("conv3_lp", 2 * 32 * V4 * 2 + 27 * 32 * 32 * 2),
    "gwc_fused_kernel<8, float>": ("gwc_fused", 4 * (2 * 320 * 136 * 240 + 40 * V4)),
    "::deconv3_bf16x3_kernel": ("deconv3_bf16x3", 4 * (64 * V4 // 8 + 32 * V4) + 27 * 64 * 32 * 4),
    "conv1_x3_kernel<2, 0": ("conv1_x3", 4 * 64 * V4),
}


def _rows(directory):
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        yield from csv.DictReader(open(path))


def summarise(fetch_dir, write_dir, sq_dir):
    acc = {}
    for d in (fetch_dir, write_dir, sq_dir):
        for r in _rows(d):
            for pat, (key, algo) in KEYS.items():
                if pat in r["Kernel_Name"]:
                    e = acc.setdefault(key, {"algorithmic_bytes": algo, "kernel": r["Kernel_Name"].split("(")[0][:80]})
                    e.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    out = {"_how": __doc__.split("\n\n")[1].strip(), "_commit": head}
    for key, e in acc.items():
        mean = lambda k: (sum(e[k]) / len(e[k])) if k in e else None
        f, w = mean("FETCH_SIZE"), mean("WRITE_SIZE")
        o = {"kernel": e["kernel"], "algorithmic_bytes": e["algorithmic_bytes"], "launches": len(e.get("FETCH_SIZE", []))}
        if f is not None and w is not None:
            o.update(FETCH_SIZE_KiB=round(f, 1), WRITE_SIZE_KiB=round(w, 1), hbm_bytes_per_launch=(2 * f + w) * 1024,
                     traffic_over_algorithmic=round((2 * f + w) * 1024 / e["algorithmic_bytes"], 3))
        for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU",
                  "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"):
            if mean(k) is not None:
                o[k] = round(mean(k), 1)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in o and "GRBM_GUI_ACTIVE" in o and o["GRBM_GUI_ACTIVE"] > 0:
            # SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs (check: = 32 x SQ_INSTS_MFMA for 32x32x16
            # bf16); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so GUI_ACTIVE / 8 = the kernel's duration in shader cycles
            cyc = o["GRBM_GUI_ACTIVE"] / 8.0
            o["kernel_cycles"] = round(cyc, 1)
            o["mfma_busy_frac"] = round(o["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 4)
        out[key] = o
    path = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
        summarise(*sys.argv[2:5])
    else:
        import torch
        import bench
        print(json.dumps(bench.kernel_roofline(torch.device("cuda", 0), "bf16"))[:400])
