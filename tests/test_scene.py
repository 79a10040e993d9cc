"""vcm_scene_cornell (product, host) must reproduce, bit for bit, the scene
descs flattened from the reference's own Scene objects (tests/golden/scene_*)."""
import glob
import os
import re

import pytest

from smallvcm_amd._abi import SCENE_CONFIGS, SceneDesc
from smallvcm_amd.renderer import cornell_scene, load_library
import oracle_lib

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases():
    out = []
    for p in sorted(glob.glob(os.path.join(GOLD, "scene_*.bin"))):
        m = re.match(r"scene_(\d+)_(\d+)\.bin", os.path.basename(p))
        out.append((int(m.group(1)), int(m.group(2)), p))
    return out


@pytest.mark.parametrize("sid,res,path", _cases())
def test_scene_builder_matches_reference_golden(sid, res, path):
    gold = open(path, "rb").read()
    mine = cornell_scene(sid, res, res).tobytes()
    assert mine == gold


def test_scene_masks():
    L = load_library(require_gpu=False)
    for sid in range(4):
        assert L.vcm_scene_config_mask(sid) == SCENE_CONFIGS[sid]


@pytest.mark.skipif(not oracle_lib.have_ref(), reason="oracle/_ref not built (no /root/reference)")
def test_scene_builder_matches_live_reference_nonsquare():
    for sid in range(4):
        for (rx, ry) in [(40, 24), (17, 33)]:
            assert cornell_scene(sid, rx, ry).tobytes() == oracle_lib.ref_scene(SCENE_CONFIGS[sid], rx, ry).tobytes()
    # masks outside g_SceneConfigs: no light box + ceiling light, large glass sphere, both large spheres
    for mask in (1 | 64 | 128, 1 | 32, 16 | 32 | 1, 4 | 16, 8 | 2 | 1 | 256):
        assert cornell_scene(mask, 32, 32, is_mask=True).tobytes() == oracle_lib.ref_scene(mask, 32, 32).tobytes()
