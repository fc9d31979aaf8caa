"""Launches the two dominant kernels a few times at the BASELINE shape (for rocprofv3 --pmc passes).
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python tools/pmc_kernels.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    print(bench.kernel_roofline(torch.device("cuda", 0)))
