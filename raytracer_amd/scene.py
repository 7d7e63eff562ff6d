"""Python mirror of the reference's scene-description API (the host side of the drop-in
boundary): Texture / Material / Mesh / DisplayObject / Cube / Sphere / Scene / Camera.

Same names, argument meaning and bool-returning error behaviour as the reference classes
(src/Material.hpp:17-117, src/Mesh.hpp:27-62, src/DisplayObject.hpp:31-105, src/Cube.hpp,
src/Sphere.hpp, src/Scene.hpp:41-176, src/Camera.hpp:16-132).  This layer is pure host
logic (numpy); it owns no device memory.  `Scene.flatten()` turns the object graph into
the plain arrays the C-ABI (include/rt_api.h) takes; tests hand the same arrays to the
CPU oracle.  The C++ mirror with the same class names lives in raytracer_amd/host/.

Documented deviations:
 * object iteration order is insertion order (the reference iterates an unordered_map);
 * there are no process-wide static registries: materials/textures are plain objects.
"""
from __future__ import annotations

import math
import os
import weakref
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import hostmath

F = np.float32
FLT_MAX = np.finfo(np.float32).max

DEFAULT_POSITION = (0.0, 0.0, 0.0)
DEFAULT_ROTATION = (0.0, 0.0, 0.0, 0.0)  # glm::quat() value-initialises to all zero
DEFAULT_SCALE = (1.0, 1.0, 1.0)


def expand_to_rgba8(img: np.ndarray) -> np.ndarray:
    """Texture ctor's 4-channel expansion (src/Material.cu:184-197), quirk included: the
    loop reads idata[i*n+1] and [i*n+2] whatever n is, so for 1- and 2-channel images G and
    B come from the following bytes.  Bytes past the end of the decoded image (a 2-byte
    over-read in the reference) are defined as 0 here."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    h, w, n = img.shape
    flat = np.concatenate([img.reshape(-1), np.zeros(4, np.uint8)])
    i = np.arange(w * h, dtype=np.int64) * n
    out = np.empty((h * w, 4), np.uint8)
    out[:, 0] = flat[i]
    out[:, 1] = flat[i + 1]
    out[:, 2] = flat[i + 2]
    out[:, 3] = flat[i + 3] if n == 4 else 255
    return out.reshape(h, w, 4)


class _TexelStore:
    """The decoded image several Texture objects of one path share (the reference's textureMap entry,
    src/Material.cu:165-178): `rgba8` is uploaded once per FlatScene, whatever the number of Textures."""

    def __init__(self):
        self.rgba8: Optional[np.ndarray] = None
        self.loadSuccess = False


_TEXTURE_MAP = weakref.WeakValueDictionary()  # path -> _TexelStore, alive while any Texture of that path is (ref count, :234-252)
TEXTURE_DECODES = 0                           # files actually decoded (tests: two loads of one path decode once)


def _decode_elsewhere(path):
    """Files stb_image reads and rt_image_load does not (progressive and four-component JPEGs; anything else Pillow knows):
    decoded with Pillow where it is installed, with stb_image's channel count.  A parity deviation, stated in DESIGN.md: a
    JPEG's bytes then differ from the reference's by up to a few levels (another inverse DCT); the C++ mirror refuses such
    files (getLoadSuccess() false: the reference's convention for an unreadable one)."""
    import os
    if not os.path.isfile(path):
        return None
    try:
        from PIL import Image
        im = Image.open(path)
        im.load()
    except Exception:
        return None
    if im.mode not in ("L", "LA", "RGB", "RGBA"):
        im = im.convert("RGBA" if (im.mode == "P" and "transparency" in im.info) else "RGB")
    return np.asarray(im, np.uint8)


class Texture:
    """src/Material.hpp:17-50.  `Texture(path)` decodes with the library's rt_image_load (include/rt_api.h: PNG, and
    baseline JPEG with stb_image's arithmetic, so the texel bytes are what the reference's stbi_load gives -- pinned
    against that stb_image itself in tests/test_texture_decode.py) or wraps an already decoded HxWxC uint8 array.
    Textures are shared by path as in the reference (src/Material.cu:165-178): a second Texture of a path
    that is still alive decodes nothing and flattens to the same device texture."""

    def __init__(self, path_or_array):
        global TEXTURE_DECODES
        self.path = ""
        self._store = _TexelStore()
        if isinstance(path_or_array, np.ndarray):
            self._store.rgba8 = expand_to_rgba8(path_or_array)
            self._store.loadSuccess = True
        else:
            self.path = str(path_or_array)
            shared = _TEXTURE_MAP.get(self.path)
            if shared is not None:
                self._store = shared
                return
            _TEXTURE_MAP[self.path] = self._store
            TEXTURE_DECODES += 1
            try:
                from . import capi   # late: capi imports this module
                arr = capi.image_load(self.path)
                if arr is None:
                    arr = _decode_elsewhere(self.path)
                if arr is not None:
                    self._store.rgba8 = expand_to_rgba8(arr)
                    self._store.loadSuccess = True
            except Exception:
                self._store.loadSuccess = False

    rgba8 = property(lambda self: self._store.rgba8)
    loadSuccess = property(lambda self: self._store.loadSuccess)

    def getWidth(self):
        return 0 if self.rgba8 is None else self.rgba8.shape[1]

    def getHeight(self):
        return 0 if self.rgba8 is None else self.rgba8.shape[0]

    def getLoadSuccess(self):
        return self.loadSuccess

    def getPath(self):
        return self.path


class Material:
    """src/Material.hpp:92-117, src/Material.cu:10-49,113-160."""

    def __init__(self, name, albedo=(0.9, 0.9, 0.9, 1.0), roughness=0.5, metal=0.0,
                 emissionColor=(0.0, 0.0, 0.0), emission=0.0):
        self.materialName = str(name)
        self.simpleAlbedo = np.asarray(albedo, np.float32).reshape(4)
        self.simpleRoughness = F(roughness)
        self.simpleMetal = F(metal)
        self.simpleEmissionColor = np.asarray(emissionColor, np.float32).reshape(3)
        self.emissionStrength = F(emission)
        self.normalTexture: Optional[Texture] = None
        self.albedoTexture: Optional[Texture] = None
        self.roughnessTexture: Optional[Texture] = None
        self.metalTexture: Optional[Texture] = None
        self.ambientOcclusionTexture: Optional[Texture] = None  # never reaches the device
        self.emissionTexture: Optional[Texture] = None  # never reaches the device
        self.success = True

    def getName(self):
        return self.materialName

    def getSuccess(self):
        return self.success

    def _set(self, attr, src) -> bool:
        # setTextureRef (src/Material.cu:113-130): empty path clears, failed load keeps the old one
        if src is None or (isinstance(src, str) and src == ""):
            setattr(self, attr, None)
            return True
        tex = src if isinstance(src, Texture) else Texture(src)
        if not tex.getLoadSuccess():
            return False
        setattr(self, attr, tex)
        return True

    def setNormalTexture(self, p):
        return self._set("normalTexture", p)

    def setAlbedoTexture(self, p):
        return self._set("albedoTexture", p)

    def setRoughnessTexture(self, p):
        return self._set("roughnessTexture", p)

    def setMetalTexture(self, p):
        return self._set("metalTexture", p)

    def setAmbientOcclusionTexture(self, p):
        return self._set("ambientOcclusionTexture", p)

    def setEmissionTexture(self, p):
        return self._set("emissionTexture", p)


class Mesh:
    """src/Mesh.hpp:27-62: flat AoS arrays, local AABB (calcBounds :106-122)."""

    def __init__(self, vertices, indices, uvCoords=None, normals=None):
        self.vertices = np.ascontiguousarray(vertices, np.float32).reshape(-1)
        self.indices = np.ascontiguousarray(indices, np.int32).reshape(-1)
        self.uvCoords = None if uvCoords is None or len(uvCoords) == 0 else \
            np.ascontiguousarray(uvCoords, np.float32).reshape(-1)
        self.normals = None if normals is None or len(normals) == 0 else \
            np.ascontiguousarray(normals, np.float32).reshape(-1)
        self.triangleCount = self.indices.size // 3
        v = self.vertices.reshape(-1, 3)
        if v.shape[0]:
            self.minBound = v.min(axis=0).astype(np.float32)
            self.maxBound = v.max(axis=0).astype(np.float32)
        else:
            self.minBound = np.full(3, FLT_MAX, np.float32)
            self.maxBound = np.full(3, -FLT_MAX, np.float32)

    def getTriangleCount(self):
        return self.triangleCount

    def getMinBound(self):
        return self.minBound

    def getMaxBound(self):
        return self.maxBound


class DisplayObject:
    """src/DisplayObject.hpp:31-105, src/DisplayObject.cpp:19-28,83-105."""

    def __init__(self, position=DEFAULT_POSITION, rotation=DEFAULT_ROTATION, scale=DEFAULT_SCALE):
        self.Position = np.asarray(position, np.float32).reshape(3)
        self.Rotation = np.asarray(rotation, np.float32).reshape(4)  # x, y, z, w
        self.Scale = np.asarray(scale, np.float32).reshape(3)
        self.meshes: List[Tuple[Mesh, int]] = []
        self.materials: List[Material] = []
        self.isComposite = False
        self.loadSuccess = False
        self.compositeMinBounds = np.zeros(3, np.float32)
        self.compositeMaxBounds = np.zeros(3, np.float32)

    def getModelMatrix(self):
        return hostmath.model_matrix(self.Position, self.Rotation, self.Scale)

    def getInverseModelMatrix(self):
        return hostmath.inverse(self.getModelMatrix())

    def setPosition(self, p):
        self.Position = np.asarray(p, np.float32).reshape(3)

    def setRotation(self, q):
        self.Rotation = np.asarray(q, np.float32).reshape(4)

    def setScale(self, s):
        self.Scale = np.asarray(s, np.float32).reshape(3)

    def setMaterial(self, material: Material):
        """setMaterialName (DisplayObject.hpp:54) without the global name registry."""
        self.materials.append(material)

    def setMaterialVector(self, materials: Sequence[Material]):
        self.materials.extend(materials)

    def getMeshes(self):
        return self.meshes

    def getMaterials(self):
        return self.materials

    def getLoadSuccess(self):
        return self.loadSuccess

    def getMinBound(self):
        return self.compositeMinBounds

    def getMaxBound(self):
        return self.compositeMaxBounds

    def updateCompositeBounds(self):
        if not self.isComposite:
            self.compositeMaxBounds = self.meshes[0][0].getMaxBound().copy()
            self.compositeMinBounds = self.meshes[0][0].getMinBound().copy()
            return
        mx = np.full(3, -FLT_MAX, np.float32)
        mn = np.full(3, FLT_MAX, np.float32)
        for mesh, _ in self.meshes:
            mx = np.maximum(mx, mesh.getMaxBound())
            mn = np.minimum(mn, mesh.getMinBound())
        self.compositeMaxBounds, self.compositeMinBounds = mx.astype(np.float32), mn.astype(np.float32)


_CUBE_MESH: Optional[Mesh] = None
_SPHERE_MESH: Optional[Mesh] = None


def cube_mesh() -> Mesh:
    """The 36-vertex / 12-triangle unit cube of src/Cube.hpp:10-111 (UVs, no normals),
    generated from its face/corner pattern instead of a literal table."""
    global _CUBE_MESH
    if _CUBE_MESH is None:
        # per face: 4 corners (sign triples); vertices are corners 0,1,2,2,3,0
        faces = [
            ((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1)),
            ((1, -1, 1), (-1, -1, 1), (-1, 1, 1), (1, 1, 1)),
            ((-1, -1, 1), (-1, -1, -1), (-1, 1, -1), (-1, 1, 1)),
            ((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1)),
            ((1, -1, -1), (-1, -1, -1), (-1, -1, 1), (1, -1, 1)),
            ((1, 1, 1), (-1, 1, 1), (-1, 1, -1), (1, 1, -1)),
        ]
        order = (0, 1, 2, 2, 3, 0)
        corner_uv = ((0.0, 0.0), (0.0, 1.0), (1.0, 1.0), (1.0, 0.0))
        verts, uvs = [], []
        for f in faces:
            for k in order:
                verts.extend(0.5 * c for c in f[k])
                uvs.extend(corner_uv[k])
        _CUBE_MESH = Mesh(np.array(verts, np.float32), np.arange(36, dtype=np.int32),
                          np.array(uvs, np.float32), None)
    return _CUBE_MESH


def sphere_mesh(subdiv_x=16, subdiv_y=12) -> Mesh:
    """UV sphere of src/Sphere.hpp:19-125: radius 0.5, 219 vertices / 352 triangles for the
    default 16x12, normals = 2*position, Sphere.hpp's own PI = 3.14159f.  sinf/cosf are
    evaluated in binary64 and rounded to binary32 (pinned; the reference used MSVC's)."""
    global _SPHERE_MESH
    if (subdiv_x, subdiv_y) == (16, 12) and _SPHERE_MESH is not None:
        return _SPHERE_MESH
    SX, SY = subdiv_x, subdiv_y
    PI = F(3.14159)
    v, n, uv = [], [], []
    for i in range(SX):  # south pole, one vertex per segment
        v += [F(0), F(-0.5), F(0)]
        n += [F(0), F(-1), F(0)]
        uv += [F(2 * i + 1) / F(2 * SX), F(0)]
    for i in range(1, SY):
        for j in range(SX + 1):
            theta = F(i) / F(SY) * PI
            phi = F(j) / F(SX) * F(2) * PI
            st, ct = F(math.sin(float(theta))), F(math.cos(float(theta)))
            sp, cp = F(math.sin(float(phi))), F(math.cos(float(phi)))
            x = F(0.5) * st * cp
            y = F(-0.5) * ct
            z = F(0.5) * st * sp
            v += [x, y, z]
            n += [x * F(2), y * F(2), z * F(2)]
            uv += [F(j) / F(SX), F(i) / F(SY)]
    for i in range(SX):  # north pole
        v += [F(0), F(0.5), F(0)]
        n += [F(0), F(1), F(0)]
        uv += [F(2 * i + 1) / F(2 * SX), F(1)]
    idx = []
    for i in range(SX):
        idx += [i, i + SX + 1, i + SX]
    for i in range(SY - 2):
        for j in range(SX):
            a = SX + (SX + 1) * i + j
            b = SX + (SX + 1) * (i + 1) + j
            idx += [a, b + 1, b, a, a + 1, b + 1]
    bottom = SX + (SX + 1) * (SY - 2)
    for i in range(SX):
        idx += [bottom + i, bottom + i + 1, bottom + i + 1 + SX]
    m = Mesh(np.array(v, np.float32), np.array(idx, np.int32), np.array(uv, np.float32), np.array(n, np.float32))
    if (subdiv_x, subdiv_y) == (16, 12):
        _SPHERE_MESH = m
    return m


class Cube(DisplayObject):
    """src/Cube.hpp:5-118.  The reference constructor does NOT forward position / rotation /
    scale to DisplayObject (Cube.hpp:8), so a Cube always has the identity transform until
    setPosition/... is called.  Reproduced."""

    def __init__(self, position=DEFAULT_POSITION, rotation=DEFAULT_ROTATION, scale=DEFAULT_SCALE):
        super().__init__()
        self.meshes.append((cube_mesh(), 0))
        self.updateCompositeBounds()
        self.loadSuccess = True


class Sphere(DisplayObject):
    """src/Sphere.hpp:9-134."""

    def __init__(self, position=DEFAULT_POSITION, rotation=DEFAULT_ROTATION, scale=DEFAULT_SCALE):
        super().__init__(position, rotation, scale)
        self.meshes.append((sphere_mesh(), 0))
        self.updateCompositeBounds()
        self.loadSuccess = True


class MeshObject(DisplayObject):
    """A DisplayObject built from explicit (Mesh, materialIndex) pairs -- what CustomModel
    (src/CustomModel.cpp:135-145) produces after the importer has run."""

    def __init__(self, meshes: Sequence[Tuple[Mesh, int]], materials: Sequence[Material] = (),
                 position=DEFAULT_POSITION, rotation=DEFAULT_ROTATION, scale=DEFAULT_SCALE):
        super().__init__(position, rotation, scale)
        self.meshes = list(meshes)
        self.materials = list(materials)
        self.isComposite = len(self.meshes) > 1
        if self.meshes:
            self.updateCompositeBounds()
        self.loadSuccess = len(self.meshes) > 0


class CustomModel(MeshObject):
    """src/CustomModel.hpp / CustomModel.cpp:47-194: a model file turned into (Mesh, materialIndex)
    pairs and Materials.  The importer is raytracer_amd.assets (COLLADA, OBJ/MTL, binary FBX) in
    place of assimp; its flattening rules, and why parity with assimp's is unpinned, are stated
    there.  As in the reference, a file that cannot be read gives getLoadSuccess() == False rather
    than an exception, textures are looked up next to the asset (or next to its "source"
    directory, :110-112), and a texture that fails to load leaves the material's constant."""

    _MODELS = weakref.WeakValueDictionary()   # path -> _ModelData, alive while any CustomModel of that path is (customModels, :10-45)
    PARSES = 0                                # files actually parsed (tests)

    class _ModelData:
        def __init__(self, meshes, materials):
            self.meshes, self.materials = meshes, materials

    def __init__(self, path, position=DEFAULT_POSITION, rotation=DEFAULT_ROTATION, scale=DEFAULT_SCALE):
        from . import assets

        self.path = str(path)
        shared = CustomModel._MODELS.get(self.path)
        if shared is not None:   # copyHostLLData (:10-19): the same Mesh and Material objects, hence the same device handles
            self._data = shared
            super().__init__(list(shared.meshes), list(shared.materials), position, rotation, scale)
            return
        CustomModel.PARSES += 1
        meshes, materials = [], []
        try:
            asset = assets.load_asset(self.path)
            directory = asset.texture_directory()
            for fm in asset.materials:
                m = Material(fm.name)
                setters = {"albedo": m.setAlbedoTexture, "normal": m.setNormalTexture, "roughness": m.setRoughnessTexture,
                           "metal": m.setMetalTexture, "ao": m.setAmbientOcclusionTexture, "emission": m.setEmissionTexture}
                for role, rel in fm.textures.items():
                    if rel:
                        setters[role]((directory + "/" + rel).replace("\\", "/"))
                materials.append(m)
            for fmesh in asset.meshes:
                if fmesh.indices.shape[0] == 0:
                    continue
                meshes.append((Mesh(fmesh.positions, fmesh.indices, fmesh.uvs, fmesh.normals), int(fmesh.material)))
        except Exception:
            meshes, materials = [], []
        self._data = CustomModel._ModelData(meshes, materials)
        CustomModel._MODELS[self.path] = self._data
        super().__init__(list(meshes), list(materials), position, rotation, scale)


@dataclass
class FlatScene:
    """Plain arrays crossing the C-ABI (and handed to the oracle)."""
    meshes: List[Mesh] = field(default_factory=list)
    textures: List[np.ndarray] = field(default_factory=list)  # HxWx4 uint8
    materials: List[dict] = field(default_factory=list)
    objects: List[dict] = field(default_factory=list)

    def triangle_count(self) -> int:
        return sum(sum(self.meshes[m].triangleCount for m in o["meshIds"]) for o in self.objects)


class Scene:
    """src/Scene.hpp:41-176, src/Scene.cpp:3-29."""

    def __init__(self):
        self.objectMap: Dict[str, DisplayObject] = {}

    def addToScene(self, name: str, obj: DisplayObject) -> bool:
        if name in self.objectMap:
            return False
        self.objectMap[name] = obj
        return True

    def removeFromSceen(self, name: str) -> bool:  # sic, Scene.hpp:55
        return self.objectMap.pop(name, None) is not None

    def getObjects(self):
        return self.objectMap

    def flatten(self) -> FlatScene:
        fs = FlatScene()
        mesh_id: Dict[int, int] = {}
        tex_id: Dict[int, int] = {}
        mat_id: Dict[int, int] = {}

        def tex(t: Optional[Texture]) -> int:
            if t is None:
                return -1
            key = id(t._store)   # Textures of one path share their texels: one device texture
            if key not in tex_id:
                tex_id[key] = len(fs.textures)
                fs.textures.append(t.rgba8)
            return tex_id[key]

        for obj in self.objectMap.values():
            mids, mat_indices = [], []
            for mesh, mi in obj.meshes:
                if id(mesh) not in mesh_id:
                    mesh_id[id(mesh)] = len(fs.meshes)
                    fs.meshes.append(mesh)
                mids.append(mesh_id[id(mesh)])
                mat_indices.append(int(mi))
            mats = []
            for m in obj.materials:
                if id(m) not in mat_id:
                    mat_id[id(m)] = len(fs.materials)
                    # Material::sendToGPU (src/Material.cu:30-49): only these four textures
                    fs.materials.append(dict(
                        albedo=m.simpleAlbedo.copy(), roughness=F(m.simpleRoughness), metal=F(m.simpleMetal),
                        emissionColor=m.simpleEmissionColor.copy(), emissionStrength=F(m.emissionStrength),
                        texNormal=tex(m.normalTexture), texAlbedo=tex(m.albedoTexture),
                        texRoughness=tex(m.roughnessTexture), texMetal=tex(m.metalTexture)))
                mats.append(mat_id[id(m)])
            model = obj.getModelMatrix()
            fs.objects.append(dict(
                model=model.reshape(16).copy(), invModel=hostmath.inverse(model).reshape(16).copy(),
                bmin=obj.compositeMinBounds.astype(np.float32), bmax=obj.compositeMaxBounds.astype(np.float32),
                meshIds=mids, materialIndices=mat_indices, materialIds=mats))
        return fs


class Camera:
    """src/Camera.hpp:16-132."""

    def __init__(self, position=(0.0, 0.0, 10.0), rotation=DEFAULT_ROTATION, fov=45.0, exposure=1.0):
        self.Position = np.asarray(position, np.float32).reshape(3)
        self.Rotation = np.asarray(rotation, np.float32).reshape(4)
        self.verticalFOV = F(fov)
        self.exposureValue = F(exposure)
        self.width = 0
        self.height = 0
        self.preTransformRays: Optional[np.ndarray] = None

    def setWidth(self, w):
        self.width = int(w)

    def setHeight(self, h):
        self.height = int(h)

    def getWidth(self):
        return self.width

    def getHeight(self):
        return self.height

    def getExposure(self):
        return self.exposureValue

    def getPosition(self):
        return self.Position

    def getRotation(self):
        return self.Rotation

    def setPosition(self, p):
        self.Position = np.asarray(p, np.float32).reshape(3)

    def setRotation(self, q):
        self.Rotation = np.asarray(q, np.float32).reshape(4)

    def update(self):
        if self.preTransformRays is None or self.preTransformRays.shape[:2] != (self.height, self.width):
            self.calcRays()

    def calcRays(self):
        self.preTransformRays = hostmath.camera_rays(self.width, self.height, self.verticalFOV)


@dataclass
class RenderSettings:
    """Raytracer's protected state + defaults (src/processor/Raytracer.hpp:12-62) and the GUI
    defaults pushed every frame (src/RenderWindow.cpp:323-335)."""
    bounceCount: int = 3
    maxDistance: float = 100.0
    aoIntensity: float = 0.01
    frameCount: int = 0
    progressiveFrameCount: int = 0
    progressiveRendering: bool = False
    antiAliasingEnabled: bool = True
    sampleCount: int = 1
    lightDirection: np.ndarray = field(default_factory=lambda: hostmath.sky_direction(0.0, 0.0))
    lightColor: np.ndarray = field(default_factory=lambda: np.array([1, 1, 1, 100], np.float32))
    skyColor: np.ndarray = field(default_factory=lambda: np.array([0.5, 0.5, 1.0, 0.3], np.float32))


def default_material() -> Material:
    """The `Default` material RenderWindow creates (src/RenderWindow.cpp:167-168)."""
    return Material("Default", (0.5, 0.5, 0.5, 1.0), 0.5, 0.0)
