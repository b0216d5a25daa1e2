#!/bin/bash
# round 6: ds_read_b64_tr_b16 / ds_read_b128 bank behaviour on candidate attention row images: timing + SQ_LDS_BANK_CONFLICT
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
$ROOT/tools/probes/tr_bank_probe > $ROOT/gpurun_out/r06_tr_bank_probe.jsonl 2>&1
rm -rf $ROOT/gpurun_out/pmc_trprobe
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_trprobe -o t -- $ROOT/tools/probes/tr_bank_probe > $ROOT/gpurun_out/pmc_trprobe.log 2>&1
python - <<'PY'
import csv, glob, os, json
root=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
names=[json.loads(l)["pattern"] for l in open(root+"/pmc_trprobe.log") if l.startswith("{")]
rows={}
for f in glob.glob(root+"/pmc_trprobe/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]]=float(r["Counter_Value"])
ids=sorted(rows)
# every pattern = 4 dispatches (2 x 1 block of 64, 2 x 1024 blocks of 256): the 4th is the full-grid one
out=open(root+"/r06_tr_bank_probe_pmc.txt","w")
print("SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the full-grid dispatch of each pattern (tools/probes/tr_bank_probe.hip)", file=out)
for i,n in enumerate(names):
    if 4*i+3 >= len(ids): break
    c=rows[ids[4*i+3]]
    print(f"{n:90s} conflict {c.get('SQ_LDS_BANK_CONFLICT',0):.4g} active {c.get('SQ_LDS_IDX_ACTIVE',0):.4g} ratio {c.get('SQ_LDS_BANK_CONFLICT',0)/max(1,c.get('SQ_LDS_IDX_ACTIVE',1)):.3f} insts {c.get('SQ_INSTS_LDS',0):.4g}", file=out)
out.close()
print(open(root+"/r06_tr_bank_probe_pmc.txt").read())
PY
cat $ROOT/gpurun_out/r06_tr_bank_probe.jsonl
find $ROOT/gpurun_out/pmc_trprobe -name "*.csv" -size +4M -delete
