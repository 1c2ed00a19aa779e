"""Per-iteration trace of the LM loop on the benchmark sequence from the triangulation start (what do the 22 iterations do?).
usage: lm_trace.py [frames] [lam0]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from acinoset_amd import fte, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
lam0 = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
seq = synth.make_sequence(n, "loop"); rig = (seq["K"], seq["D"], seq["R"], seq["t"])
d = torch.as_tensor(seq["det"], device="cuda")
xa = fte.triangulation_init_active(d, *rig, 0.5)
ctx = fte.FTEContext(d, *rig, seq["Ts"], lam0=lam0)
ctx.set_x(xa)
prev = None
for it in range(60):
    ctx.step()
    st = ctx.state()
    print(f"it {it + 1:3d} cost {st['cost']:.9e} trial {st['cost_trial']:.9e} pred {st['pred']:.3e} lam {st['lam']:.2e} acc {st['last_accept']} "
          f"step {st.get('step', float('nan')):.2e} gnorm {st.get('gnorm_inf', float('nan')):.2e} status {st['status_name']}")
    if st["status"] != 0:
        break
ctx.close()
