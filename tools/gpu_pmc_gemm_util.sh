#!/bin/bash
# PMC passes (kernel-trace only, one counter group per pass) over the hand-written 256x256 GEMM and hipBLASLt's kernel on the
# training shapes: MFMA-pipe busy, instruction mix, LDS activity / conflicts, wait cycles.  Summary -> gpurun_out/pmc_gemm_util_summary.txt
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM"; do
  i=$((i+1)); rm -rf $ROOT/gpurun_out/pmc_gemm_util_$i
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_gemm_util_$i -o t -- python $ROOT/tools/pmc_gemm_pair.py > $ROOT/gpurun_out/pmc_gemm_util_$i.log 2>&1
  tail -1 $ROOT/gpurun_out/pmc_gemm_util_$i.log | cut -c1-200
done
python - <<'PY'
import csv, glob, os
from collections import defaultdict
root=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
out=open(root+"/pmc_gemm_util_summary.txt","w")
print("tools/gpu_pmc_gemm_util.sh: hand-written gemm256_kernel vs hipBLASLt (Cijk_..._MT256x256x64_MI16x16x1) at M = 32768, (N, K) = (4096, 4096) / (16384, 4096) / (4096, 16384);", file=out)
print("per (kernel, grid): mean counter value per launch, one counter group per pass", file=out)
tot=defaultdict(dict)
for d in sorted(glob.glob(root+"/pmc_gemm_util_[0-9]")):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        agg=defaultdict(lambda:[0.0,0])
        for r in csv.DictReader(open(f)):
            kn=r["Kernel_Name"].replace("(anonymous namespace)::","")
            if "gemm256" not in kn and "Cijk" not in kn: continue
            k=(("mine " if "gemm256" in kn else "hipblaslt ")+r.get("Grid_Size", r.get("Grid_Size_X","?")), r["Counter_Name"])
            agg[k][0]+=float(r["Counter_Value"]); agg[k][1]+=1
        for (kn,cn),(v,n) in sorted(agg.items()):
            print(f"{kn:26s} {cn:28s} per_launch={v/n:.6g} launches={n}", file=out)
            tot[kn][cn]=v/n
print("\nderived: MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); per-MFMA instruction mix", file=out)
for kn,c in sorted(tot.items()):
    if c.get("GRBM_GUI_ACTIVE",0)>0 and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        fr=c["SQ_VALU_MFMA_BUSY_CYCLES"]/(1024.0*c["GRBM_GUI_ACTIVE"]/8.0)
        mix=""
        if c.get("SQ_INSTS_MFMA",0)>0:
            m=c["SQ_INSTS_MFMA"]
            mix=f"  per MFMA: VALU {c.get('SQ_INSTS_VALU',0)/m:.2f} SALU {c.get('SQ_INSTS_SALU',0)/m:.2f} LDS {c.get('SQ_INSTS_LDS',0)/m:.2f} VMEM {c.get('SQ_INSTS_VMEM',0)/m:.3f} SMEM {c.get('SQ_INSTS_SMEM',0)/m:.3f}"
        print(f"{kn:26s} MFMA busy {fr:.3f}  gui_active {c['GRBM_GUI_ACTIVE']:.4g}{mix}", file=out)
out.close(); print(open(root+"/pmc_gemm_util_summary.txt").read())
PY
find $ROOT/gpurun_out/pmc_gemm_util_* -name "*.csv" -size +4M -delete
