#!/bin/bash
# Three separate rocprofv3 counter passes over tools/pmc_kernels.py (run on the GPU box from the repo root).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
set -e
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_r03c_fetch -- python tools/pmc_kernels.py > gpurun_out/pmc_r03c_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_r03c_write -- python tools/pmc_kernels.py > gpurun_out/pmc_r03c_write.log 2>&1
echo "write pass done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_r03c_sq -- python tools/pmc_kernels.py > gpurun_out/pmc_r03c_sq.log 2>&1
echo "sq pass done"
