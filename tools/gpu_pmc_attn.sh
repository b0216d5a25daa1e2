#!/bin/bash
# PMC passes over the attention micro-benchmark (kernel-trace only, one counter group per pass)
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU"; do
  i=$((i+1)); rm -rf $ROOT/gpurun_out/pmc_attn_$i
  AB=16 ABWD=${ABWD:-4} timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_attn_$i -o t -- python $ROOT/tools/attn_bench.py > $ROOT/gpurun_out/pmc_attn_$i.log 2>&1
  tail -2 $ROOT/gpurun_out/pmc_attn_$i.log | cut -c1-300
done
python - <<'PY'
import csv, glob, os
from collections import defaultdict
root=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
out=open(root+"/pmc_attn_summary.txt","w")
tot=defaultdict(dict)
for d in sorted(glob.glob(root+"/pmc_attn_[0-9]")):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        agg=defaultdict(lambda:[0.0,0])
        for r in csv.DictReader(open(f)):
            k=(r["Kernel_Name"].replace("(anonymous namespace)::","")[:40], r["Counter_Name"])
            agg[k][0]+=float(r["Counter_Value"]); agg[k][1]+=1
        for (kn,cn),(v,n) in sorted(agg.items()):
            if "attn" in kn:
                print(f"{kn:42s} {cn:28s} per_launch={v/n:.6g} launches={n}", file=out)
                tot[kn][cn]=v/n
# derived: MFMA-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE cycles of the launch)
print("\nderived (per launch): MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8); GRBM_GUI_ACTIVE is summed over the 8 XCDs", file=out)
for kn,c in sorted(tot.items()):
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"]>0:
        fr=c["SQ_VALU_MFMA_BUSY_CYCLES"]/(1024.0*c["GRBM_GUI_ACTIVE"]/8.0)
        extra=""
        if c.get("SQ_INSTS_MFMA", 0) > 0:
            extra=f"  VALU/MFMA {c.get('SQ_INSTS_VALU',0)/c['SQ_INSTS_MFMA']:.2f}  SALU/MFMA {c.get('SQ_INSTS_SALU',0)/c['SQ_INSTS_MFMA']:.2f}  LDS/MFMA {c.get('SQ_INSTS_LDS',0)/c['SQ_INSTS_MFMA']:.2f}"
        print(f"{kn:42s} MFMA busy {fr:.3f}  gui_active_cycles {c['GRBM_GUI_ACTIVE']:.4g}{extra}", file=out)
out.close()
print(open(root+"/pmc_attn_summary.txt").read())
PY
find $ROOT/gpurun_out/pmc_attn_* -name "*.csv" -size +4M -delete
