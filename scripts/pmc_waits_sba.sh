#!/bin/bash
# Where do the waves of each kernel spend their cycles?  Two rocprofv3 counter passes over a 3-step bench run.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/waits_sba; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
B="python $R/scripts/sba_config5.py f64 4"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p1 -o p1 -- $B > /dev/null 2> $OUT/p1.err
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_MFMA --output-format csv -d $OUT/p2 -o p2 -- $B > /dev/null 2> $OUT/p2.err
cd $R
python - <<'PY'
import csv, glob, os, re
from collections import defaultdict
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "waits_sba")
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
        if not k.startswith("acino::"): continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
lines = []
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    vm = (c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)) or 1
    lines.append(f"{k[7:]:22s} launches {n[k]:4d} wave-cycles/launch {wc / max(n[k],1):12.0f} | wait_any {c.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst_any {c.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} "
                 f"wait_lds {c.get('SQ_WAIT_INST_LDS',0)/wc:5.2f} | active any {c.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f} valu {c.get('SQ_ACTIVE_INST_VALU',0)/wc:5.2f} lds {c.get('SQ_ACTIVE_INST_LDS',0)/wc:5.2f} "
                 f"vmem {c.get('SQ_ACTIVE_INST_VMEM',0)/wc:5.2f} | insts valu {c.get('SQ_INSTS_VALU',0):.3g} mfma {c.get('SQ_INSTS_MFMA',0):.3g} lds {c.get('SQ_INSTS_LDS',0):.3g} vmem {vm:.3g} "
                 f"| avg vmem latency {c.get('SQ_INST_LEVEL_VMEM',0)/vm:8.0f} cyc, lds {c.get('SQ_INST_LEVEL_LDS',0)/(c.get('SQ_INSTS_LDS',0) or 1):6.0f} cyc")
open(os.path.join(out, "waits.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
