import importlib, sys
sys.path.insert(0, '.')
import numpy as np
vsa = importlib.import_module("visual-slam_amd")
w,h=640,480
img = vsa.synth_frame(0,w,h)
p = vsa.make_params()
ex = vsa.Extractor(p,w,h,1,0)
ex.extract(img)
q = vsa.scale_tables(p)["quota"]
for l in range(8):
    keys = ex.debug_candidates(0,l)
    lw,lh = vsa.level_size(p,w,h,l)
    for st in (0,1):
        out = vsa.distribute_octree(keys,16,lw-16,16,lh-16,int(q[l]),device=0,key_storage=st)
    print(l, len(keys), len(out))
