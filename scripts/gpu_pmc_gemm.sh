#!/bin/bash
# MFMA-pipe utilisation of k_update (scripts/update_bench.hip binary) and of rocBLAS DGEMM
# (scripts/dgemm_ref.py) on the same shapes: rocprofv3 --pmc passes (kernel trace only).
#   bash scripts/gpu_pmc_gemm.sh        (on the GPU box, from the repository root; needs bin_tmp/ub_cur)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_gemm
mkdir -p $OUT
export TMPDIR=/tmp
# (the library destroys its pooled CU-masked streams at exit only when asked: rocprofv3 crashes on live ones)
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/ub -o ub -- $REPO/bin_tmp/ub_cur 8192 8192 > $OUT/ub.log 2>&1
echo "update_bench rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/rb -o rb -- python3 $REPO/scripts/dgemm_ref.py > $OUT/rb.log 2>&1
echo "dgemm_ref rc=$?"
python3 - "$OUT" <<'PY'
import glob, os, sys
import pandas as pd
out = sys.argv[1]
for tag in ("ub", "rb"):
    fs = glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        print(tag, "no counters"); continue
    t = pd.read_csv(fs[0])
    g = t.pivot_table(index=["Dispatch_Id", "Kernel_Name"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
    g = g[g["GRBM_GUI_ACTIVE"] > 0]
    # per-SIMD busy fraction: SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs' 4-cycle quads (MI355X_MICROARCH.md)
    g["mfma_busy_pct"] = 100.0 * g["SQ_VALU_MFMA_BUSY_CYCLES"] / (g["GRBM_GUI_ACTIVE"] * 256 * 4) * 4
    g["kernel"] = g["Kernel_Name"].str.slice(0, 60)
    s = g.groupby("kernel").agg(n=("mfma_busy_pct", "size"), busy_max=("mfma_busy_pct", "max"), busy_mean=("mfma_busy_pct", "mean"), active_max=("GRBM_GUI_ACTIVE", "max"))
    print("==", tag); print(s.sort_values("active_max", ascending=False).head(8).to_string())
    g[["Dispatch_Id", "kernel", "GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "mfma_busy_pct"]].to_csv(os.path.join(out, tag + "_mfma.csv"), index=False)
PY
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
