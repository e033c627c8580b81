import sys,os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, ctypes as C
import bench
from fealess_amd import api, _lib as L
class A: pass
args=A(); args.levels=2; args.scenes=8; args.templates=360; args.batch=16
ctx=api.Context(0)
bank,bgrs,depths,scenes=bench.build_workload(ctx,args,0)
det=api.Detector(ctx,2,[5,8]); det.add_class(bank); det.finalize(640,480,max_batch=16,max_candidates=65536)
res=det.recognize_batch(list(bgrs),list(depths),(608.,608.,320.,240.),75.0,20,-1.0,-3e38)
print('n_matches', [r['n_matches'] for r in res])
# raw candidate counts
import torch
buf=np.zeros(4,np.int32)
for f in range(4):
    m,n=det.match(bgrs[f],depths[f],75.0)
    cnt=(C.c_int32*4)(); det.lib.fl_frame_counters(det.h, 0, cnt); print('frame',f,'matches',n,'counters',list(cnt))
