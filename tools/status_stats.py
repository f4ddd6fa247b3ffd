#!/usr/bin/env python
"""GPU-box helper: how many candidates of the bench workloads do the window filters drop (status 0), and
how are they clustered -- would compacting the survivors before the quantizer / forest pay?"""
import sys, numpy as np
sys.path.insert(0,'/root/repo')
import bench
from peakachu_amd import _lib
for w,n,band,fs in ((5,30000,200,None),(6,30000,300,None),(11,8000,200,'peakachu_amd/data/forest_w11_t500.npz')):
    Mf,e,x,y,upper=bench.build_workload(0,n,band,w,6,band)
    fo=bench.load_forest(fs,w,(2*w+1)**2)
    hm=_lib.HipMatrix(Mf.indptr,Mf.indices,Mf.data,Mf.shape[0],e,-2*w+1,upper+2*w-1); hf=_lib.HipForest(fo); cd=_lib.HipCands(x,y)
    cd.run(hm,hf,w,0.5); st,pr=cd.fetch_all()
    print('w',w,'cands',x.size,'status0',(st==0).mean(),'status2',(st==2).sum(), 'p>0.5',(pr>0.5).sum(), 'p==0', (pr==0).mean())
    # how clustered are the filtered ones? per 128-tile all-inactive
    t=st[:st.size//128*128].reshape(-1,128); print('  tiles fully inactive', (t==0).all(1).mean(), 'waves(64) fully inactive', (st[:st.size//64*64].reshape(-1,64)==0).all(1).mean())
