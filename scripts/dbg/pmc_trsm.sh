#!/bin/bash
# PMC passes over the stand-alone chain-block bench (k_trsm_rows with M rows): bash scripts/dbg/pmc_trsm.sh M
M=${1:-32768}
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/pmc_trsm_$M
mkdir -p $OUT; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -I include -I spllt_amd/csrc scripts/chain_block_bench.hip -o /tmp/cbb 2>/dev/null || exit 1
cd /tmp
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  tag=$(echo $SET | cut -d' ' -f1)
  timeout -k 10 120 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/$tag -o p -- /tmp/cbb 256 $M > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"][:28]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "trsm_rows" in k or "chain_block" in k:
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
done
find $OUT -name "*.csv" -delete
