import sys
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
from oracle_lib import Oracle
from smallvcm_amd.renderer import VertexCM, cornell_scene
for sid,algo,res in [(1,4,256),(1,2,256),(3,4,256),(1,4,512)]:
    sc=cornell_scene(sid,res,res)
    o=Oracle(sc,algo,threads=64); o.run_iteration(0,0,10); ref=o.framebuffer()
    for strict in (False,True):
        r=VertexCM(sc,algo,0.003,0.75,1234,strict_order=strict); r.mMaxPathLength=10; r.RunIteration(0)
        fb=r.framebuffer_sum(); d=np.abs(fb-ref)
        rel=(d/np.maximum(np.abs(ref),1e-3)).max()
        print(sid,algo,res,"strict" if strict else "wavefront","maxabs %.3e maxrel %.3e rmse %.3e"%(d.max(),rel,np.sqrt((d.astype(np.float64)**2).mean())), {k:round(v,3) for k,v in r.stats().items() if k.startswith('ms')})
        r.close()
