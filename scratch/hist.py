import sys; sys.path.insert(0,'/root/repo')
from smallvcm_amd.renderer import VertexCM, cornell_scene
v=VertexCM(cornell_scene(1,2048,2048),4,0.003,0.75,1234); v.mMaxPathLength=10
n=44
for i in range(n): v.RunIteration(i)
v.backend.synchronize()
for ago in range(n-1,-1,-1):
    s=v.backend.stats_at(ago)
    print(n-1-ago, "r=%.5f"%s["radius"], "tot %.2f"%s["msTotal"], "merge %.2f"%s["msMergeKernel"], "sort %.2f"%s["msQuerySort"], "C %.2fG A %.0fM"%(s["mergeCandidates"]/1e9, s["mergeAccepted"]/1e6))
