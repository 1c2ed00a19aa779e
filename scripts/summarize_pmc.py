#!/usr/bin/env python3
"""Summarise rocprofv3 counter CSVs (gpurun_out/round/rocprof_pmc_*) per kernel -> profiles-ready JSON/CSV.

  python scripts/summarize_pmc.py gpurun_out/round profiles/roundX

HBM traffic: FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it is
doubled (MI355X_MICROARCH.md, HBM section).  MFMA: SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 = fp64 matrix flops executed;
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs), GRBM_GUI_ACTIVE taken per XCD (rocprofv3 reports
the sum over the 8 XCDs).  LDS bank-conflict rate =
SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (cycles)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acinoset_amd import _lib  # noqa: E402
# identity of the binary these counters were measured on (hash over every HIP source + header): bench.py only quotes a
# profiles/ JSON whose build_id equals the library it is running
BUILD_ID = _lib.built_id()


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "")


acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))        # counter -> kernel -> [launches, total]
for f in glob.glob(os.path.join(src, "rocprof_pmc_*", "*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        a = acc[row["Counter_Name"]][short(row["Kernel_Name"])]
        a[0] += 1
        a[1] += float(row["Counter_Value"])

rows = [("counter", "kernel", "launches", "total", "per_launch")]
for cn, ks in sorted(acc.items()):
    for k, (n, tot) in sorted(ks.items(), key=lambda kv: -kv[1][1]):
        if k.startswith("acino::"):
            rows.append((cn, k, n, tot, tot / n))
csv.writer(open(os.path.join(dst, "rocprofv3_pmc_by_kernel.csv"), "w")).writerows(rows)

traffic = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, bench.py --steps 3 --warmup 1 "
                     "--no-secondary at 10 000 frames; FETCH_SIZE doubled (gfx950: 128-B requests tallied at 64 B)",
           "build_id": BUILD_ID, "kernels": {}}
for k in acc.get("FETCH_SIZE", {}):
    if not k.startswith("acino::"):
        continue
    n, fk = acc["FETCH_SIZE"][k]
    nw, wk = acc.get("WRITE_SIZE", {}).get(k, [n, 0.0])
    traffic["kernels"][k] = dict(bytes_per_launch=(2.0 * fk / n + wk / max(nw, 1)) * 1024.0, fetch_kb_raw=fk / n,
                                 write_kb_raw=wk / max(nw, 1), launches=n)
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)

if "SQ_VALU_MFMA_BUSY_CYCLES" in acc:
    out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE "
                     "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES (one pass, own run)",
           "build_id": BUILD_ID, "kernels": {}}
    for k, (n, busy) in acc["SQ_VALU_MFMA_BUSY_CYCLES"].items():
        if not k.startswith("acino::"):
            continue
        g = lambda c: acc.get(c, {}).get(k, [1, 0.0])[1]
        gui = g("GRBM_GUI_ACTIVE") / 8.0     # the CSV value is the SUM over the 8 XCDs; the formulas use the per-die count
        out["kernels"][k] = dict(
            launches=n,
            mfma_f64_flops_per_launch=g("SQ_INSTS_VALU_MFMA_MOPS_F64") * 512.0 / n,
            mfma_util_percent=100.0 * busy / (gui * 1024.0) if gui else None,
            gpu_active_cycles_per_launch=gui / n,
            cu_busy_fraction=g("SQ_BUSY_CU_CYCLES") / (gui * 256.0) if gui else None,
            valu_active_fraction_of_wave_cycles=(g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")) if g("SQ_WAVE_CYCLES") else None,
            lds_bank_conflict_rate=(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")) if g("SQ_LDS_IDX_ACTIVE") else None)
    json.dump(out, open(os.path.join(dst, "pmc_mfma_lds.json"), "w"), indent=1)
print("wrote", os.listdir(dst))
