import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
import prlib_amd
from prlib_amd import synth
from oracle import capi as oc
pg = synth.page_numpy(300, 700, 3)
for m,w,k in [(0,31,0.34),(1,31,0.2),(3,21,-0.1),(4,21,0.0),(2,31,0.3)]:
    got = prlib_amd.binarize(torch.from_numpy(pg).cuda(), prlib_amd.make_params(m,w,k,0)).cpu().numpy()
    want = oc.binarize(pg, oc.make_params(m,w,k,0))
    st = prlib_amd.last_stats()
    print(m, "white got/want", (got>0).mean(), (want>0).mean(), "mismatch", (got!=want).sum(), st.refined_pixels, st.exact_pixels, np.unique(got)[:5])
