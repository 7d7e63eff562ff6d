"""Flattens the reference's own model files (data, not code) into tests/golden/ref_assets.npz so that
BASELINE.json configs[2] and [3] render the reference's real geometry on the GPU box, where
/root/reference does not exist.  Run in the container:  python tests/golden/make_assets.py

  res/pbrsword/source/murasama.fbx        -> sword_*   (1 mesh, 8 972 triangles, 18 664 corners)
  res/lowpolypbrtrash/source/model.dae    -> trash{k}_* (6 meshes, 3 024 triangles) + trash_material (slot per mesh)
  res/pbrsword/texture/*_Roughness/_Metallic.png      -> sword_roughness / sword_metallic   (2048^2, 8-bit gray)
  res/metalmaterial/*_Roughness/_Metalness.png        -> metal_roughness / metal_metalness  (1024^2, 8-bit gray)

The flattening is raytracer_amd.assets' (fan triangulation, per-polygon-vertex expansion, v -> 1 - v):
the reference does it with assimp, which is not available here (parity with assimp's vertex order is
unpinned).  8-bit gray PNGs decode to the same bytes with any conforming decoder
(PNG is lossless), so these texel arrays are what the reference's stb_image yields.  The 16-bit base
colour (3.4 MB), the JPEG maps of lowpolypbrtrash (decoder dependent) and the normal maps (missing from
the reference checkout) are not included; scenes.py keeps procedural maps in their place."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from raytracer_amd import assets  # noqa: E402

REF = "/root/reference/res"


def gray(path):
    from PIL import Image
    im = Image.open(path)
    assert im.mode == "L", (path, im.mode)
    return np.asarray(im).copy()


def main():
    out = {}
    a = assets.load_asset(f"{REF}/pbrsword/source/murasama.fbx")
    assert len(a.meshes) == 1 and a.meshes[0].indices.shape == (8972, 3)
    m = a.meshes[0]
    out.update(sword_pos=m.positions, sword_idx=m.indices, sword_uv=m.uvs, sword_nrm=m.normals)
    a = assets.load_asset(f"{REF}/lowpolypbrtrash/source/model.dae")
    assert len(a.meshes) == 6 and sum(x.indices.shape[0] for x in a.meshes) == 3024
    for k, m in enumerate(a.meshes):
        out.update({f"trash{k}_pos": m.positions, f"trash{k}_idx": m.indices, f"trash{k}_uv": m.uvs, f"trash{k}_nrm": m.normals})
    out["trash_material"] = np.array([m.material for m in a.meshes], np.int32)
    out["sword_roughness"] = gray(f"{REF}/pbrsword/texture/Material.002_Roughness.png")
    out["sword_metallic"] = gray(f"{REF}/pbrsword/texture/Material.002_Metallic.png")
    out["metal_roughness"] = gray(f"{REF}/metalmaterial/Metal046B_1K-PNG_Roughness.png")
    out["metal_metalness"] = gray(f"{REF}/metalmaterial/Metal046B_1K-PNG_Metalness.png")
    path = os.path.join(HERE, "ref_assets.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
