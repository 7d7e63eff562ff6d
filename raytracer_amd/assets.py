"""assets.py -- native asset flatteners: COLLADA, Wavefront OBJ/MTL and binary FBX into the
(positions, indices, uvs, normals, material) arrays that the upload path takes.

Replaces the assimp dependency of the reference's importer (src/CustomModel.cpp:47-194:
`ReadFile(path, Triangulate | GenUVCoords | FlipUVs)`, then per aiMesh the vertex / uv / normal
arrays and the triangle indices, `:147-194`; per aiMaterial the texture file names by type,
`:76-133`).  assimp itself is not vendored in the reference and its version is not pinned, so
its triangulation and vertex order cannot be reproduced bit for bit: PARITY UNPINNED (SURVEY.md
8c).  This module defines the flattening instead, deterministically:

  * one output mesh per primitive group (COLLADA `<triangles>/<polylist>/<polygons>`, OBJ
    `usemtl` run, FBX geometry x material), in file order -- like assimp, which splits by material;
  * one vertex per polygon corner, in file order (no welding), attributes gathered per corner;
  * polygons with more than three corners are triangulated as a fan around their first corner;
  * texture v is flipped (v -> 1 - v), as aiProcess_FlipUVs does; meshes without uvs get zeros
    (`CustomModel.cpp:164-168`); meshes without normals get none (`:170-175`);
  * node transforms are NOT applied: the reference reads `scene->mMeshes` directly (`:137-141`).

Materials: name plus the texture path per role (albedo, normal, roughness, metal, ao, emission),
resolved like `processMaterial` does (`:110-113`: the asset's directory, or its parent when the
asset sits in a directory called "source").
"""
import os
import re
import struct
import zlib
import xml.etree.ElementTree as ET

import numpy as np

ROLES = ("albedo", "normal", "roughness", "metal", "ao", "emission")


class FlatMesh:
    """One primitive group: per-corner vertices and fan-triangulated indices."""

    def __init__(self, name, positions, indices, uvs, normals, material):
        self.name = name
        self.positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1, 3)
        self.uvs = np.ascontiguousarray(uvs, dtype=np.float32).reshape(-1, 2)
        self.normals = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        self.material = material  # index into FlatAsset.materials


class FlatMaterial:
    def __init__(self, name):
        self.name = name
        self.textures = {}  # role -> path relative to the asset's texture directory (as written in the file)
        self.albedo = None  # constant colour when the file gives one


class FlatAsset:
    def __init__(self, path):
        self.path = path
        self.meshes = []
        self.materials = []

    @property
    def triangle_count(self):
        return int(sum(m.indices.shape[0] for m in self.meshes))

    def texture_directory(self):
        """CustomModel.cpp:110-112: strip the file name, and a trailing "/source"."""
        d = os.path.dirname(self.path.replace("\\", "/"))
        return os.path.dirname(d) if os.path.basename(d) == "source" else d


def _fan(corner_counts):
    """Triangle indices of consecutive polygons with the given corner counts (per-corner vertices)."""
    tris = []
    base = 0
    for n in corner_counts:
        for k in range(1, n - 1):
            tris.append((base, base + k, base + k + 1))
        base += n
    return np.asarray(tris, dtype=np.int32).reshape(-1, 3)


def _material_index(asset, name):
    for i, m in enumerate(asset.materials):
        if m.name == name:
            return i
    asset.materials.append(FlatMaterial(name))
    return len(asset.materials) - 1


# ------------------------------------------------------------------------------------ COLLADA
def _strip_ns(root):
    for el in root.iter():
        if "}" in el.tag:
            el.tag = el.tag.split("}", 1)[1]


def load_dae(path):
    """COLLADA 1.4: every <triangles>, <polylist> and <polygons> of every <geometry>, in file order."""
    root = ET.parse(path).getroot()
    _strip_ns(root)
    asset = FlatAsset(path)

    # material id -> effect -> image file, for the roles COLLADA's common profile knows
    images = {im.get("id"): (im.findtext("init_from") or "").strip() for im in root.iter("image")}
    effects = {}
    for ef in root.iter("effect"):
        params = {}
        for np_ in ef.iter("newparam"):
            surf = np_.find("surface")
            samp = np_.find("sampler2D")
            if surf is not None:
                params[np_.get("sid")] = ("surface", (surf.findtext("init_from") or "").strip())
            elif samp is not None:
                params[np_.get("sid")] = ("sampler", (samp.findtext("source") or "").strip())

        def resolve(tex):
            ref = tex.get("texture")
            for _ in range(3):
                if ref in params:
                    ref = params[ref][1]
            return images.get(ref, ref)

        info = {}
        for role, tag in (("albedo", "diffuse"), ("emission", "emission"), ("normal", "bump")):
            el = next(iter(ef.iter(tag)), None)
            if el is not None:
                tex = el.find("texture")
                if tex is not None:
                    info[role] = resolve(tex)
                col = el.findtext("color")
                if col and role == "albedo":
                    info["albedo_color"] = [float(v) for v in col.split()]
        effects[ef.get("id")] = info
    for mat in root.iter("material"):
        fm = FlatMaterial(mat.get("name") or mat.get("id"))
        ie = mat.find("instance_effect")
        info = effects.get((ie.get("url") or "#")[1:], {}) if ie is not None else {}
        for role in ROLES:
            if role in info:
                fm.textures[role] = info[role]
        fm.albedo = info.get("albedo_color")
        fm.collada_id = mat.get("id")
        asset.materials.append(fm)

    def mat_index(symbol):
        for i, m in enumerate(asset.materials):
            if getattr(m, "collada_id", None) == symbol or m.name == symbol:
                return i
        return _material_index(asset, symbol or "MissingName")

    for geom in root.iter("geometry"):
        mesh = geom.find("mesh")
        if mesh is None:
            continue
        sources = {}
        for src in mesh.findall("source"):
            fa = src.find("float_array")
            acc = src.find("technique_common/accessor")
            stride = int(acc.get("stride", "1")) if acc is not None else 1
            vals = np.asarray((fa.text or "").split(), dtype=np.float64) if fa is not None else np.zeros(0)
            sources["#" + src.get("id")] = vals.reshape(-1, stride)
        vertices = {}
        for v in mesh.findall("vertices"):
            vertices["#" + v.get("id")] = {i.get("semantic"): i.get("source") for i in v.findall("input")}
        for prim in mesh:
            if prim.tag not in ("triangles", "polylist", "polygons"):
                continue
            inputs = [(i.get("semantic"), i.get("source"), int(i.get("offset", "0")), int(i.get("set", "0")))
                      for i in prim.findall("input")]
            if not inputs:
                continue
            stride = max(o for _, _, o, _ in inputs) + 1
            if prim.tag == "polygons":
                polys = [np.asarray(p.text.split(), dtype=np.int64) for p in prim.findall("p")]
                counts = [len(p) // stride for p in polys]
                p = np.concatenate(polys) if polys else np.zeros(0, dtype=np.int64)
            else:
                p = np.asarray((prim.findtext("p") or "").split(), dtype=np.int64)
                if prim.tag == "triangles":
                    counts = [3] * (len(p) // (3 * stride))
                else:
                    counts = [int(c) for c in (prim.findtext("vcount") or "").split()]
            p = p.reshape(-1, stride)
            pos = uv = nrm = None
            for sem, src, off, set_ in inputs:
                idx = p[:, off]
                if sem == "VERTEX":
                    for vsem, vsrc in vertices[src].items():
                        if vsem == "POSITION":
                            pos = sources[vsrc][idx][:, :3]
                        elif vsem == "NORMAL" and nrm is None:
                            nrm = sources[vsrc][idx][:, :3]
                        elif vsem == "TEXCOORD" and uv is None:
                            uv = sources[vsrc][idx][:, :2]
                elif sem == "NORMAL":
                    nrm = sources[src][idx][:, :3]
                elif sem == "TEXCOORD" and (uv is None or set_ == 0):
                    uv = sources[src][idx][:, :2]
            if pos is None:
                continue
            keep = [c >= 3 for c in counts]  # points and lines are dropped (the reference skips non-triangles, :183-187)
            if not all(keep):
                sel = np.concatenate([np.full(c, k) for c, k in zip(counts, keep)]) if counts else np.zeros(0, bool)
                pos = pos[sel]
                uv = uv[sel] if uv is not None else None
                nrm = nrm[sel] if nrm is not None else None
                counts = [c for c in counts if c >= 3]
            uvs = np.zeros((len(pos), 2)) if uv is None else np.stack([uv[:, 0], 1.0 - uv[:, 1]], axis=1)
            asset.meshes.append(FlatMesh(geom.get("name") or geom.get("id"), pos, _fan(counts), uvs, nrm,
                                         mat_index(prim.get("material"))))
    if not asset.materials:
        asset.materials.append(FlatMaterial("MissingName0"))
    return asset


# ------------------------------------------------------------------------------------ OBJ / MTL
_MTL_ROLES = {"map_kd": "albedo", "map_bump": "normal", "bump": "normal", "norm": "normal", "map_ns": "roughness",
              "map_pr": "roughness", "map_refl": "metal", "refl": "metal", "map_pm": "metal", "map_ke": "emission",
              "map_ka": "ao"}


def load_mtl(path, asset):
    cur = None
    with open(path, "r", errors="replace") as f:
        for line in f:
            t = line.strip()
            if not t or t.startswith("#"):
                continue
            key, _, rest = t.partition(" ")
            key = key.lower()
            rest = rest.strip()
            if key == "newmtl":
                cur = asset.materials[_material_index(asset, rest)]
            elif cur is None:
                continue
            elif key == "kd":
                cur.albedo = [float(v) for v in rest.split()[:3]] + [1.0]
            elif key in _MTL_ROLES:
                # options such as "-bm 1.0" precede the file name; the name may contain blanks
                toks = rest.split(" ")
                while toks and toks[0].startswith("-"):
                    toks = toks[2:]
                cur.textures[_MTL_ROLES[key]] = " ".join(toks)


def load_obj(path):
    """Wavefront OBJ: one mesh per run of faces that share a material (`usemtl`), per-corner vertices."""
    asset = FlatAsset(path)
    v, vt, vn = [], [], []
    groups = []  # (material name, [polygons of (vi, ti, ni)])
    cur = None
    base = os.path.dirname(path)
    with open(path, "r", errors="replace") as f:
        for line in f:
            t = line.strip()
            if not t or t.startswith("#"):
                continue
            key, _, rest = t.partition(" ")
            if key == "v":
                v.append([float(x) for x in rest.split()[:3]])
            elif key == "vt":
                c = [float(x) for x in rest.split()[:2]]
                vt.append(c + [0.0] * (2 - len(c)))
            elif key == "vn":
                vn.append([float(x) for x in rest.split()[:3]])
            elif key == "mtllib":
                mp = os.path.join(base, rest.strip())
                if os.path.exists(mp):
                    load_mtl(mp, asset)
            elif key == "usemtl":
                cur = (rest.strip(), [])
                groups.append(cur)
            elif key == "f":
                if cur is None:
                    cur = ("MissingName0", [])
                    groups.append(cur)
                poly = []
                for tok in rest.split():
                    parts = tok.split("/")
                    vi = int(parts[0])
                    ti = int(parts[1]) if len(parts) > 1 and parts[1] else 0
                    ni = int(parts[2]) if len(parts) > 2 and parts[2] else 0
                    # 1-based; negative = relative to the end
                    poly.append((vi - 1 if vi > 0 else len(v) + vi, ti - 1 if ti > 0 else (len(vt) + ti if ti < 0 else -1),
                                 ni - 1 if ni > 0 else (len(vn) + ni if ni < 0 else -1)))
                if len(poly) >= 3:
                    cur[1].append(poly)
    V = np.asarray(v, dtype=np.float64).reshape(-1, 3)
    VT = np.asarray(vt, dtype=np.float64).reshape(-1, 2)
    VN = np.asarray(vn, dtype=np.float64).reshape(-1, 3)
    for name, polys in groups:
        if not polys:
            continue
        corners = [c for p in polys for c in p]
        vi = np.asarray([c[0] for c in corners])
        ti = np.asarray([c[1] for c in corners])
        ni = np.asarray([c[2] for c in corners])
        pos = V[vi]
        if (ti >= 0).all() and len(VT):
            uv = np.stack([VT[ti][:, 0], 1.0 - VT[ti][:, 1]], axis=1)
        else:
            uv = np.zeros((len(corners), 2))
        nrm = VN[ni] if (ni >= 0).all() and len(VN) else None
        asset.meshes.append(FlatMesh(name, pos, _fan([len(p) for p in polys]), uv, nrm, _material_index(asset, name)))
    if not asset.materials:
        asset.materials.append(FlatMaterial("MissingName0"))
    return asset


# ------------------------------------------------------------------------------------ binary FBX
class _FbxNode:
    __slots__ = ("name", "props", "children")

    def __init__(self, name, props, children):
        self.name, self.props, self.children = name, props, children

    def find(self, name):
        for c in self.children:
            if c.name == name:
                return c
        return None

    def all(self, name):
        return [c for c in self.children if c.name == name]


_FBX_ARRAY = {b"f": "<f4", b"d": "<f8", b"l": "<i8", b"i": "<i4", b"b": "u1"}
_FBX_SCALAR = {b"Y": "<h", b"C": "<?", b"I": "<i", b"F": "<f", b"D": "<d", b"L": "<q"}


_FBX_MAX_DEPTH = 64


def _fbx_read_node(d, pos, wide, depth=0):
    """One node record.  A malformed or truncated file raises (struct.error / ValueError), which CustomModel turns into
    getLoadSuccess() == False: reads are bounds-checked by struct, a compressed array may inflate to n * itemsize bytes
    and no further, a record must end inside the file and after its start, nesting is limited."""
    if depth > _FBX_MAX_DEPTH:
        raise ValueError("FBX: nesting too deep")
    if wide:
        end, nprop, _plen = struct.unpack_from("<QQQ", d, pos)
        pos += 24
    else:
        end, nprop, _plen = struct.unpack_from("<III", d, pos)
        pos += 12
    nlen = d[pos]
    pos += 1
    if end == 0:
        return None, pos
    if end > len(d) or end < pos + nlen:
        raise ValueError("FBX: record end outside the file")
    name = d[pos:pos + nlen].decode("ascii", "replace")
    pos += nlen
    props = []
    for _ in range(nprop):
        t = d[pos:pos + 1]
        pos += 1
        if t in _FBX_SCALAR:
            fmt = _FBX_SCALAR[t]
            props.append(struct.unpack_from(fmt, d, pos)[0])
            pos += struct.calcsize(fmt)
        elif t in _FBX_ARRAY:
            n, enc, clen = struct.unpack_from("<III", d, pos)
            pos += 12
            if pos + clen > len(d):
                raise ValueError("FBX: array past the end of the file")
            raw = d[pos:pos + clen]
            pos += clen
            want = n * np.dtype(_FBX_ARRAY[t]).itemsize
            if enc == 1:
                if want > clen * 1100 + 64:   # deflate expands at most ~1032-fold
                    raise ValueError("FBX: array larger than its compressed bytes can hold")
                raw = zlib.decompressobj().decompress(raw, want + 1)
            if len(raw) < want:
                raise ValueError("FBX: array shorter than declared")
            props.append(np.frombuffer(raw, dtype=_FBX_ARRAY[t], count=n))
        elif t in (b"S", b"R"):
            n = struct.unpack_from("<I", d, pos)[0]
            pos += 4
            props.append(d[pos:pos + n] if t == b"R" else d[pos:pos + n].decode("utf-8", "replace"))
            pos += n
        else:
            raise ValueError(f"FBX: unknown property type {t!r}")
    children = []
    while pos < end:
        c, pos = _fbx_read_node(d, pos, wide, depth + 1)
        if c is None:
            break
        children.append(c)
    return _FbxNode(name, props, children), end


def parse_fbx(path):
    d = open(path, "rb").read()
    if d[:20] != b"Kaydara FBX Binary  ":
        raise ValueError("not a binary FBX file (ASCII FBX is not supported)")
    version = struct.unpack_from("<I", d, 23)[0]
    wide = version >= 7500
    pos = 27
    top = []
    while pos < len(d) - (25 if wide else 13):
        n, pos = _fbx_read_node(d, pos, wide)
        if n is None:
            break
        top.append(n)
    return _FbxNode("", [version], top)


_FBX_TEX_ROLES = {"diffusecolor": "albedo", "basecolor": "albedo", "normalmap": "normal", "bump": "normal",
                  "emissivecolor": "emission", "shininessexponent": "roughness", "shininess": "roughness",
                  "reflectionfactor": "metal", "specularcolor": "metal", "ambientcolor": "ao"}


def _layer(geom, tag, data_tag, index_tag, width, corner_vertex, n_corner, poly_of_corner):
    el = geom.find(tag)
    if el is None or el.find(data_tag) is None:
        return None
    data = np.asarray(el.find(data_tag).props[0], dtype=np.float64).reshape(-1, width)
    mapping = (el.find("MappingInformationType").props[0] if el.find("MappingInformationType") else "ByPolygonVertex")
    ref = (el.find("ReferenceInformationType").props[0] if el.find("ReferenceInformationType") else "Direct")
    if mapping == "ByPolygonVertex":
        key = np.arange(n_corner)
    elif mapping in ("ByVertice", "ByVertex", "ByControlPoint"):
        key = corner_vertex
    elif mapping == "ByPolygon":
        key = poly_of_corner
    elif mapping == "AllSame":
        key = np.zeros(n_corner, dtype=np.int64)
    else:
        return None
    if ref in ("IndexToDirect", "Index") and el.find(index_tag) is not None:
        idx = np.asarray(el.find(index_tag).props[0], dtype=np.int64)
        key = idx[key]
    key = np.clip(key, 0, len(data) - 1)
    return data[key]


def load_fbx(path):
    """Binary FBX 7.x: every Geometry of type Mesh, split by material, in file order."""
    root = parse_fbx(path)
    asset = FlatAsset(path)
    objects = root.find("Objects")
    conns = root.find("Connections")
    if objects is None:
        raise ValueError("FBX: no Objects section")

    def clean(name):
        return name.split("\x00")[0]

    mats, texs, models, geoms = {}, {}, {}, []
    for n in objects.children:
        if n.name == "Material":
            mats[n.props[0]] = FlatMaterial(clean(n.props[1]))
        elif n.name == "Texture":
            fn = n.find("RelativeFilename") or n.find("FileName")
            texs[n.props[0]] = fn.props[0].replace("\\", "/") if fn is not None else ""
        elif n.name == "Model":
            models[n.props[0]] = n
        elif n.name == "Geometry" and n.find("Vertices") is not None:
            geoms.append(n)
    geom_model, model_mats = {}, {}
    if conns is not None:
        for c in conns.children:
            kind, src, dst = c.props[0], c.props[1], c.props[2]
            if kind == "OO" and src in mats and dst in models:
                model_mats.setdefault(dst, []).append(src)
            elif kind == "OO" and dst in models:
                geom_model[src] = dst
            elif kind == "OP" and src in texs and dst in mats:
                role = _FBX_TEX_ROLES.get(str(c.props[3]).lower())
                if role:
                    mats[dst].textures.setdefault(role, texs[src])
    mat_ids = list(mats.keys())
    asset.materials = [mats[k] for k in mat_ids]

    for g in geoms:
        verts = np.asarray(g.find("Vertices").props[0], dtype=np.float64).reshape(-1, 3)
        pvi = np.asarray(g.find("PolygonVertexIndex").props[0], dtype=np.int64)
        ends = pvi < 0
        corner_vertex = np.where(ends, ~pvi, pvi)
        n_corner = len(pvi)
        end_pos = np.nonzero(ends)[0]
        counts = np.diff(np.concatenate([[-1], end_pos]))
        n_corner = int(counts.sum())  # corners after the last closed polygon are dropped
        corner_vertex = corner_vertex[:n_corner]
        poly_of_corner = np.repeat(np.arange(len(counts)), counts)
        nrm = _layer(g, "LayerElementNormal", "Normals", "NormalsIndex", 3, corner_vertex, n_corner, poly_of_corner)
        uv = _layer(g, "LayerElementUV", "UV", "UVIndex", 2, corner_vertex, n_corner, poly_of_corner)
        pos = verts[corner_vertex]
        # material per polygon
        poly_mat = np.zeros(len(counts), dtype=np.int64)
        lm = g.find("LayerElementMaterial")
        if lm is not None and lm.find("Materials") is not None:
            m = np.asarray(lm.find("Materials").props[0], dtype=np.int64)
            mapping = lm.find("MappingInformationType").props[0] if lm.find("MappingInformationType") else "AllSame"
            if mapping == "ByPolygon" and len(m) == len(counts):
                poly_mat = m
            elif len(m):
                poly_mat[:] = m[0]
        model = geom_model.get(g.props[0])
        local = model_mats.get(model, [])
        name = clean(g.props[1]) or (clean(models[model].props[1]) if model in models else "mesh")
        for lm_idx in sorted(set(poly_mat.tolist())):
            sel_poly = poly_mat == lm_idx
            sel = sel_poly[poly_of_corner]
            cts = counts[sel_poly]
            gm = mat_ids.index(local[lm_idx]) if lm_idx < len(local) else None
            if gm is None:
                gm = _material_index(asset, "MissingName%d" % lm_idx)
            p = pos[sel]
            u = uv[sel] if uv is not None else None
            n_ = nrm[sel] if nrm is not None else None
            uvs = np.zeros((len(p), 2)) if u is None else np.stack([u[:, 0], 1.0 - u[:, 1]], axis=1)
            asset.meshes.append(FlatMesh(name, p, _fan([int(c) for c in cts]), uvs, n_, gm))
    if not asset.materials:
        asset.materials.append(FlatMaterial("MissingName0"))
    return asset


def load_asset(path):
    ext = os.path.splitext(path)[1].lower()
    if ext == ".dae":
        return load_dae(path)
    if ext == ".obj":
        return load_obj(path)
    if ext == ".fbx":
        return load_fbx(path)
    raise ValueError(f"unsupported asset format {ext!r} (COLLADA .dae, Wavefront .obj and binary .fbx are)")
