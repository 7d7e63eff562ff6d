"""Generates the committed golden fixtures from the CPU oracle (run in the authoring container:
`python tests/golden/make_golden.py`).  The reference holds no fixtures of its own for this path
(SURVEY.md section 4); these pin the oracle's current behaviour so that later edits to it, or to
the host scene layer, cannot drift silently.  The GPU tests compare the HIP path against them too."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C

from oracle import oracle_py as orc
from raytracer_amd import scenes

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    L = orc.lib()
    out = {}
    # RNG stream (GPURaytracer.cu:819-825)
    seed = C.c_uint32(12345)
    out["rng_seed12345"] = np.array([L.or_random_value(C.byref(seed)) for _ in range(64)], np.float32)
    # config 1 at 64x64 (camera moved in so that the objects cover a fair share of the frame)
    for name, (spp, depth, aa) in {"c1_s1": (1, 4, True), "c1_s5": (5, 3, True), "c1_noaa": (1, 2, False)}.items():
        scene, cam, st = scenes.config1(64, 64, spp, depth)
        cam.setPosition((0.6, 0.3, 4.0))
        st.antiAliasingEnabled = aa
        st.frameCount = 3
        fb, ids, n = orc.render(scene.flatten(), cam, st)
        out[name + "_fb"] = fb
        out[name + "_ids"] = ids
        out[name + "_rays"] = np.array([n], np.int64)
    # textured config 2 at 96x54
    scene, cam, st = scenes.config2(96, 54, 4, 4, 64)
    fb, ids, n = orc.render(scene.flatten(), cam, st)
    out["c2_fb"] = fb
    out["c2_ids"] = ids
    out["c2_rays"] = np.array([n], np.int64)
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
