#!/usr/bin/env python
"""How many contenders does the selection stage see on the config-5b shape, and what do the four stages cost there?
python scripts/dev/sel_probe.py  (on a GPU box)"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from esac_amd import api, synthetic as S
dev = torch.device("cuda", 0)
E, H, W, N = 50, 480, 640, 16384
f = S.make_frame(0, E=E, H=H, W=W, sub=1)
a = S.gating_assignment(f, N, mode="dirichlet")
eng = api.engine(0)
kw = dict(focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=1)
d_a = torch.from_numpy(a).to(dev); d_c = torch.from_numpy(f["coords"]).to(dev)
p = eng.make_params(E, H, W, N, seed=1320, call=3, **kw)
r = eng.forward_device(d_c, d_a, p)
flags = eng.read(api.BUF_EXACT_FLAGS)
print("contenders:", int(flags.sum()), "of", N)
st = eng.time_stages(d_c, d_a, p, reps=3)
print({k: round(v*1e3,1) for k,v in st.items()})
