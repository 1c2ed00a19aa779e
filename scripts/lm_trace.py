#!/usr/bin/env python3
"""LM iteration trace of the FTE solve (cost, damping, accepted steps) for the config-3 workload: 1 000-frame sprint
from the reference's nose-line initialisation.  Prints one line per iteration."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acinoset_amd import fte, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
kind = sys.argv[2] if len(sys.argv) > 2 else "sprint"
init = sys.argv[3] if len(sys.argv) > 3 else "nose_line"
seq = synth.make_sequence(n, kind)
rig = (seq["K"], seq["D"], seq["R"], seq["t"])
det = torch.as_tensor(seq["det"], device="cuda")
x0 = fte.nose_line_init(det, *rig, 0.5) if init == "nose_line" else fte.triangulation_init(det, *rig, 0.5)
ctx = fte.FTEContext(det, *rig, seq["Ts"])
ctx.set_x(np.asarray(x0)[:, fte.ACTIVE])
prev = ctx.state()
print("it cost lam accepted gnorm step_inf")
for it in range(200):
    ctx.step()
    st = ctx.state()
    print(it + 1, f"{st['cost']:.6f}", f"{st['lam']:.3e}", st["accepted"] - prev["accepted"], f"{st['gnorm_inf']:.3e}",
          f"{st.get('step_inf', float('nan')):.3e}", st.get("status_name", st.get("status")))
    prev = st
    if st.get("status", 0) not in (0, None) and st.get("status_name", "") not in ("", "running"):
        break
