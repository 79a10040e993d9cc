"""Follow-up to cpu_reference_timing.py: the unmodified reference at 2048^2 with one renderer per PHYSICAL core
(128 on the 2 x EPYC 9575F box), one iteration each (154 GB of light vertices + grids)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
from smallvcm_amd._abi import SCENE_CONFIGS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
_, wall = oracle_lib.ref_render_stock(SCENE_CONFIGS[1], 2048, 2048, 4, iterations=n, threads=n)
print(json.dumps({"res": 2048, "threads": n, "iterations": n, "wall_s": round(wall, 3),
                  "Mpaths_s": round(2.0 * 2048 * 2048 * n / wall / 1e6, 4)}))
