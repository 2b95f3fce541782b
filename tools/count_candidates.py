import sys; sys.path.insert(0,'/root/repo')
import numpy as np, orb_slam3_detailed_comments_kor_amd as pkg
ex = pkg.ORBextractor(1000,1.2,8,20,7)
for seed in (1234,1235,1240):
    img = pkg.synth.make_frame(480,752,seed)
    ex(img,(0,1000))
    print(seed, [len(ex.debug_candidates(l)[0]) for l in range(8)])
