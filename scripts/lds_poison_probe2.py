#!/usr/bin/env python3
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acinoset_amd import fte, synth
from acinoset_amd._lib import lib, check
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2691
seq = synth.make_sequence(n, "loop")
rig = (seq["K"], seq["D"], seq["R"], seq["t"])
x0 = fte.triangulation_init(seq["det"], *rig, 0.5)[:, fte.ACTIVE]
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(sa):
    c = fte.FTEContext(seq["det"], *rig, seq["Ts"], shared_gpu=True, ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True)
    c.set_x(x0)
torch.cuda.synchronize()
for it in range(2):
    for _ in range(int(os.environ.get('POISON_N', '300'))):
        check(lib().acino_debug_poison_lds(512, int(os.environ.get('POISON_SPIN', '20000')), C.c_void_p(sb.cuda_stream)))
    with torch.cuda.stream(sa):
        c.step()
    torch.cuda.synchronize()
    print(c.state()["status_name"], c.state()["accepted"], flush=True)
