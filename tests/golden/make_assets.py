"""Flattens the reference's own model files (data, not code) into tests/golden/ref_assets.npz so that
BASELINE.json configs[2] and [3] render the reference's real geometry on the GPU box, where
/root/reference does not exist.  Run in the container:  python tests/golden/make_assets.py

  res/pbrsword/source/murasama.fbx        -> sword_*   (1 mesh, 8 972 triangles, 18 664 corners)
  res/lowpolypbrtrash/source/model.dae    -> trash{k}_* (6 meshes, 3 024 triangles) + trash_material (slot per mesh)
  res/pbrsword/texture/*_Roughness/_Metallic.png      -> sword_roughness / sword_metallic   (2048^2, 8-bit gray)
  res/metalmaterial/*_Roughness/_Metalness.png        -> metal_roughness / metal_metalness  (1024^2, 8-bit gray)
  res/basiclowpoly/Airplane Texture.png               -> airplane_albedo  (1024^2, the bytes stbi_load returns: RGBA8)
  res/pbrsword/texture/Material.002_Base_Color.png    -> sword_base_color (2048^2 RGB; the file is 16-bit, stored here already
                                                         reduced to 8 bits as stb_image does, >> 8, stb_image.h:1198)

The flattening is raytracer_amd.assets' (fan triangulation, per-polygon-vertex expansion, v -> 1 - v):
the reference does it with assimp, which is not available here (parity with assimp's vertex order is
unpinned).  8-bit gray PNGs decode to the same bytes with any conforming decoder
(PNG is lossless), so these texel arrays are what the reference's stb_image yields.  The two colour maps are decoded by
the library's rt_image_load (what Texture(path) calls, Material.cu:179) and must equal, byte for byte, what the reference's
own stb_image (oracle/_ref/libstbi_ref.so, `make -C oracle ref`) returns for the same file.  The JPEG maps of lowpolypbrtrash
(4096^2 x 12 files) and the normal maps (missing from the reference checkout) are not included; scenes.py keeps procedural
maps in their place."""
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


def stb_bytes(path):
    """HxWxC uint8 through rt_image_load, checked against the reference's stb_image."""
    import ctypes as C
    from raytracer_amd import capi
    got = capi.image_load(path)
    assert got is not None, path
    L = C.CDLL(os.path.join(os.path.dirname(HERE), "..", "oracle", "_ref", "libstbi_ref.so"))
    L.stbi_load.restype = C.POINTER(C.c_ubyte)
    L.stbi_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    L.stbi_image_free.argtypes = [C.c_void_p]
    w, h, n = C.c_int(), C.c_int(), C.c_int()
    p = L.stbi_load(path.encode(), C.byref(w), C.byref(h), C.byref(n), 0)   # Material.cu:179
    assert p, path
    ref = np.ctypeslib.as_array(p, shape=(h.value, w.value, n.value)).copy()
    L.stbi_image_free(p)
    assert got.shape == ref.shape and np.array_equal(got, ref), (path, got.shape, ref.shape)
    return got


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
    out["airplane_albedo"] = stb_bytes(f"{REF}/basiclowpoly/Airplane Texture.png")
    out["sword_base_color"] = stb_bytes(f"{REF}/pbrsword/texture/Material.002_Base_Color.png")
    path = os.path.join(HERE, "ref_assets.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
