import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'python-msgwam_amd'))
import numpy as np
from helpers import load, setup_from, state_from, STATE_KEYS
from gpu_helpers import make_prop, gpu_state
from oracle import msgwam_oracle as orc
d = load("g4_saturation_online")
s = setup_from(d); st = state_from(d, "in")
p = make_prop(s, st)
cur = st
for n in range(1, 21):
    p.step(120.0, 1)
    cur = orc.rk3(s, 120.0, cur)
    g = gpu_state(p, st)
    bad = np.nonzero(np.abs(g[0]-cur[0]) > 1e-10*np.abs(cur[0]))[0]
    rel_rr = np.max(np.abs(g[3]-cur[3])/np.abs(cur[3])); rel_mm = np.max(np.abs(g[7]-cur[7])/np.abs(cur[7]))
    print(n, "bad dens rays:", bad[:10], "rel rr", rel_rr, "rel mm", rel_mm, "uu", np.max(np.abs(g[9]-cur[9])))
    if len(bad):
        i = bad[0]
        print("   gpu", g[0][i], "ora", cur[0][i], "rr", g[3][i], cur[3][i], "mm", g[7][i], cur[7][i])
        break
