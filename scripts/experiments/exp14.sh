mkdir -p $GRAFT_REPO_ROOT/gpurun_out/exp14
O=$GRAFT_REPO_ROOT/gpurun_out/exp14
export TMPDIR=/tmp; cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 --repeats 1"
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p1 -o p1 -- $B > /dev/null 2> $O/p1.err
timeout 600 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/p2 -o p2 -- $B > /dev/null 2> $O/p2.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os, re
from collections import defaultdict
out=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out","exp14")
acc=defaultdict(lambda: defaultdict(float)); n=defaultdict(int)
for f in glob.glob(os.path.join(out,"p*","**","*counter_collection.csv"),recursive=True):
    for row in csv.DictReader(open(f)):
        k=re.sub(r"\(.*","",row["Kernel_Name"]).replace("void ","")
        if not k.startswith("acino::"): continue
        acc[k][row["Counter_Name"]]+=float(row["Counter_Value"])
lines=[]
for k,c in sorted(acc.items(), key=lambda kv:-kv[1].get("SQ_WAVE_CYCLES",0)):
    lines.append(k+" "+" ".join(f"{a}={v:.4g}" for a,v in sorted(c.items())))
open(os.path.join(out,"icache.txt"),"w").write("\n".join(lines)+"\n")
print("\n".join(lines))
PY
tail -3 $O/p1.err | cut -c1-300
rm -rf $O/p1 $O/p2
