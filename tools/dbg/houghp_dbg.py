import sys, numpy as np, torch
sys.path.insert(0, '.')
import prlib_amd as prl
from oracle import capi as oc
dev = torch.device('cuda:0')
def run(name, img, cfg):
    got = prl.houghp(torch.from_numpy(img).to(dev), *cfg); want = oc.houghp(img, *cfg)
    print(name, cfg, 'EQ' if np.array_equal(got, want) else 'DIFF', got.tolist()[:6], want.tolist()[:6], flush=True)
v = np.zeros((200, 300), np.uint8); v[20:190, 150] = 255
h = np.zeros((200, 300), np.uint8); h[50, 20:280] = 255
both = np.maximum(v, h)
rng = np.random.default_rng(1)
cl = both.copy(); cl[rng.integers(0, 200, 900), rng.integers(0, 300, 900)] = 128
for cfg in ((100, 100, 5), (60, 40, 3), (30, 40, 3), (60, 40, 20)):
    run('v', v, cfg); run('h', h, cfg); run('both', both, cfg); run('clutter', cl, cfg)
for rep in range(3):
    run('rep', cl, (60, 40, 3))
