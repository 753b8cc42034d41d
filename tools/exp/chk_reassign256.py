"""diagnostic: the cells of REASSIGN_CASES['all_order2'] (n_fft 256) that differ from the golden beyond the boundary-coefficient allowance"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import audioflux_amd as af
from oracle import restate
from tests import cases
from tests.test_reassign_host import restated
from tests.test_reassign_gpu import make
gold = np.load("tests/golden/reassign.npz")
name = "all_order2"
c = cases.REASSIGN_CASES[name]
o = make(c)
x = cases.make_input(c["x"], c["samplate"])
a = o.reassign_raw(x)
want = gold[f"{name}/re"] + 1j * gold[f"{name}/im"]
Sh, vt, vf = restated(c, x)
got = a[0] + 1j * a[1]
allow, amb = restate.reassign_allowance(Sh, vt, vf, c.get("thresh", 0.001))
scale = np.abs(want).max()
d = np.abs(got - want)
bad = d > allow + 1e-5 * scale
print("bad cells", np.argwhere(bad).tolist(), "d/scale", (d[bad] / scale).tolist(), "allow/scale", (allow[bad] / scale).tolist())
# sources whose nominal target is in / next to a bad cell
it, jf = np.floor(vt + 0.5), np.floor(vf + 0.5)
for (t, f) in np.argwhere(bad):
    src = np.argwhere((np.abs(it - t) <= 1) & (np.abs(jf - f) <= 1))
    for (st, sf) in src:
        m = abs(Sh[st, sf]) / scale
        if m > 1e-3:
            print(f"  target ({t},{f}) <- source ({st},{sf}) |S|/scale {m:.3e} vt {vt[st, sf]:.6f} vf {vf[st, sf]:.6f} amb {bool(amb[st, sf])} |S|/max|Sh| {abs(Sh[st,sf])/np.abs(Sh).max():.3e}")
