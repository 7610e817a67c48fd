import sys, numpy as np, torch
sys.path.insert(0, '.')
import prlib_amd as prl
from oracle import capi as oc
dev = torch.device('cuda:0')
both = np.zeros((200, 300), np.uint8); both[20:190, 150] = 255; both[50, 20:280] = 255
got = prl.houghp(torch.from_numpy(both).to(dev), 100, 100, 5)
print(got.tolist(), oc.houghp(both, 100, 100, 5).tolist())
