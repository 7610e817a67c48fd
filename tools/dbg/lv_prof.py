import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, prlib_amd
from prlib_amd import synth
dev = torch.device('cuda:0')
col, _ = synth.text_pages_torch(16, 3508, 2480, dev, channels=3, seed=7100)
gray, _ = synth.text_pages_torch(16, 3508, 2480, dev, channels=1)
for _ in range(3):
    prlib_amd.binarizeByLocalVariances(col); prlib_amd.binarizeByLocalVariancesWithoutFilters(col)
    prlib_amd.backgroundNormalization(gray); prlib_amd.backgroundNormalization(col)
    prlib_amd.rotate(col, [3.0] * 16)
torch.cuda.synchronize()
