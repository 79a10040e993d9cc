import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mesh_scenes import bumpy_room, tilted_room
from smallvcm_amd.renderer import VertexCM
which = sys.argv[1]
sc = tilted_room(resx=1024, resy=1024) if which == "tilted" else bumpy_room(grid=72, resx=1024, resy=1024)
r = VertexCM(sc, 4, 0.003, 0.75, 1234); r.mMaxPathLength = 10
for it in range(8): r.RunIteration(it)
r.backend.synchronize()
print(r.stats())
r.close()
