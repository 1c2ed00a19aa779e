import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from acinoset_amd import fte, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
seq = synth.make_sequence(n, "loop"); rig = (seq["K"], seq["D"], seq["R"], seq["t"])
d = torch.as_tensor(seq["det"], device="cuda")
def T():
    torch.cuda.synchronize(); return time.perf_counter()
xa = fte.triangulation_init_active(d, *rig, 0.5)
for rep in range(4):
    t0 = T(); c = fte.FTEContext(d, *rig, seq["Ts"]); t1 = T()
    c.set_x(xa); t2 = T()
    c.step(); t3 = T()
    c.enable_graph(True); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        c.step(); ta = T(); c.step(); tb = T(); c.step(); tc = T()
    c.close(); t4 = T()
    print(f"rep {rep}: create {1e3*(t1-t0):.2f} set_x {1e3*(t2-t1):.2f} first step {1e3*(t3-t2):.2f} graph steps {1e3*(ta-t3):.2f} {1e3*(tb-ta):.2f} {1e3*(tc-tb):.2f} close {1e3*(t4-tc):.2f} ms", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); c = fte.FTEContext(d, *rig, seq["Ts"]); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
