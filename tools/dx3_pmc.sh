#!/bin/bash
# SQ counter passes over the transposed-conv timing script (GPU box, repo root)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_WAVE32_INSTS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/dx3_pmc_$i -- python tools/deconv_time.py > gpurun_out/dx3_pmc_$i.log 2>&1 || echo "pass $i failed"
  f=$(ls -t gpurun_out/dx3_pmc_$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "deconv3" not in k: continue
    name = ("x3 " if "bf16x3" in k else "f32 ") + "grid=%s" % r.get("Grid_Size", "?")
    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, d in agg.items():
    print(name, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done
