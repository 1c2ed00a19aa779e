#!/bin/bash
# Where do the waves of each kernel of a workload spend their cycles?  Two rocprofv3 counter passes (+ one kernel-trace pass).
#   usage: pmc_waits_any.sh <name> <command ...>     -> gpurun_out/waits_<name>/waits.txt, kernel_stats.csv
R=${GRAFT_REPO_ROOT:-$PWD}
NAME=$1; shift
OUT=$R/gpurun_out/waits_$NAME; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p1 -o p1 -- "$@" > /dev/null 2> $OUT/p1.err
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_MFMA --output-format csv -d $OUT/p2 -o p2 -- "$@" > /dev/null 2> $OUT/p2.err
# (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2: separate passes - MI355X_MICROARCH.md)
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -o p3 -- "$@" > /dev/null 2> $OUT/p3.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p4 -o p4 -- "$@" > /dev/null 2> $OUT/p4.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- "$@" > /dev/null 2> $OUT/ks.err
find $OUT/ks -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
cd $R
WAITS_OUT=$OUT python - <<'PY'
import csv, glob, os, re
from collections import defaultdict
out = os.environ["WAITS_OUT"]
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
    nl = max(n[k], 1)
    # FETCH_SIZE / WRITE_SIZE: KB; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section)
    lines.append(f"{k[7:]:34s} launches {n[k]:5d} wave-cycles/launch {wc / nl:12.0f} | wait_any {c.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst_any {c.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} "
                 f"wait_lds {c.get('SQ_WAIT_INST_LDS',0)/wc:5.2f} | active any {c.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f} valu {c.get('SQ_ACTIVE_INST_VALU',0)/wc:5.2f} lds {c.get('SQ_ACTIVE_INST_LDS',0)/wc:5.2f} "
                 f"vmem {c.get('SQ_ACTIVE_INST_VMEM',0)/wc:5.2f} | insts/launch valu {c.get('SQ_INSTS_VALU',0)/nl:.3g} mfma {c.get('SQ_INSTS_MFMA',0)/nl:.3g} lds {c.get('SQ_INSTS_LDS',0)/nl:.3g} vmem {vm/nl:.3g} "
                 f"| avg vmem latency {c.get('SQ_INST_LEVEL_VMEM',0)/vm:8.0f} cyc | HBM per launch: fetch {2.0 * c.get('FETCH_SIZE',0)/nl/1024:9.3f} MB write {c.get('WRITE_SIZE',0)/nl/1024:9.3f} MB")
open(os.path.join(out, "waits.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/ks
