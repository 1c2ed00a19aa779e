import sqlite3, sys, glob
f = sys.argv[1]
db = sqlite3.connect(f); cur = db.cursor()
rows = list(cur.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
sel = [r for r in rows if any(k in r[0] for k in ('bcr', 'assemble', 'k_setup', 'k_trial', 'k_totals', 'k_control'))]
idx = [i for i, r in enumerate(sel) if 'k_totals' in r[0]][-2] + 1
prev = None; tot = 0; gaps = 0
for r in sel[idx:idx + 48]:
    nm = r[0].split('(')[0].replace('acino::', '').replace('void ', '')[:22]
    gap = (r[1] - prev) / 1e3 if prev else 0
    d = (r[2] - r[1]) / 1e3
    print(f"{nm:24s} {d:8.1f} us  gap {gap:5.1f}  wgs {r[3] // max(r[4], 1):5d}")
    prev = r[2]; tot += d; gaps += gap
    if 'k_totals' in r[0]: break
print("kernel sum", tot, "gaps", gaps)
