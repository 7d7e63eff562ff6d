"""Asset flatteners (raytracer_amd/assets.py): COLLADA, OBJ/MTL and binary FBX files
written by the test itself, checked against the arrays they were written from; plus the reference's
own model files when the checkout is present (counts from SURVEY.md 8: 3 024 and 8 972 triangles,
18 664 corners).  assimp's own output is not available: parity with it is unpinned (assets.py)."""
import os
import struct
import zlib

import numpy as np
import pytest

from raytracer_amd import assets
from raytracer_amd.scene import CustomModel, Scene

# a quad, a triangle and a pentagon sharing vertices, with per-corner uvs and normals
V = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [2, 0, 0.5], [2, 1, 0.5], [1.5, 2, 1]], dtype=np.float64)
POLYS = [[0, 1, 2, 3], [1, 4, 2], [2, 4, 5, 6, 3]]
CORNERS = [c for p in POLYS for c in p]
UV = np.array([[0.1 * k, 0.05 * k + 0.2] for k in range(len(CORNERS))])
NRM = np.array([[0, 0, 1]] * 4 + [[0, 1, 0]] * 3 + [[1, 0, 0]] * 5, dtype=np.float64)
EXPECT_TRIS = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6], [7, 8, 9], [7, 9, 10], [7, 10, 11]])


def check_mesh(m):
    assert np.array_equal(m.indices, EXPECT_TRIS)
    assert np.allclose(m.positions, V[CORNERS].astype(np.float32))
    assert np.allclose(m.uvs, np.stack([UV[:, 0], 1.0 - UV[:, 1]], axis=1).astype(np.float32))  # FlipUVs
    assert np.allclose(m.normals, NRM.astype(np.float32))


def test_obj_mtl(tmp_path):
    (tmp_path / "m.mtl").write_text("newmtl skin\nKd 0.2 0.4 0.6\nmap_Kd tex/a b.png\nmap_Bump -bm 1.0 tex/n.png\nnewmtl other\n")
    lines = ["mtllib m.mtl"] + ["v %r %r %r" % tuple(float(x) for x in v) for v in V] + \
            ["vt %r %r" % tuple(float(x) for x in t) for t in UV] + \
            ["vn %r %r %r" % tuple(float(x) for x in n) for n in NRM] + ["usemtl skin"]
    k = 0
    for p in POLYS:
        lines.append("f " + " ".join("%d/%d/%d" % (c + 1, k + j + 1, k + j + 1) for j, c in enumerate(p)))
        k += len(p)
    lines += ["usemtl other", "f -7 -6 -5"]  # relative indices, no uvs / normals
    (tmp_path / "m.obj").write_text("\n".join(lines) + "\n")
    a = assets.load_asset(str(tmp_path / "m.obj"))
    assert [m.name for m in a.materials] == ["skin", "other"]
    assert a.materials[0].textures == {"albedo": "tex/a b.png", "normal": "tex/n.png"}
    assert a.materials[0].albedo == [0.2, 0.4, 0.6, 1.0]
    assert len(a.meshes) == 2 and a.triangle_count == 7
    check_mesh(a.meshes[0])
    assert a.meshes[1].material == 1 and a.meshes[1].normals is None and not a.meshes[1].uvs.any()
    assert np.allclose(a.meshes[1].positions, V[[0, 1, 2]].astype(np.float32))


def test_collada_polylist_and_triangles(tmp_path):
    def arr(name, a):
        return ('<source id="%s"><float_array id="%s-a" count="%d">%s</float_array><technique_common>'
                '<accessor source="#%s-a" count="%d" stride="%d"/></technique_common></source>') % (
            name, name, a.size, " ".join(repr(float(x)) for x in a.reshape(-1)), name, a.shape[0], a.shape[1])
    p = " ".join("%d %d %d" % (c, k, k) for k, c in enumerate(CORNERS))
    tri = " ".join("%d %d %d" % (c, 0, 0) for c in (0, 1, 2))
    doc = ('<?xml version="1.0"?><COLLADA xmlns="http://www.collada.org/2005/11/COLLADASchema" version="1.4.1">'
           '<library_images><image id="img0"><init_from>textures/albedo.png</init_from></image></library_images>'
           '<library_effects><effect id="fx"><profile_COMMON>'
           '<newparam sid="surf"><surface type="2D"><init_from>img0</init_from></surface></newparam>'
           '<newparam sid="samp"><sampler2D><source>surf</source></sampler2D></newparam>'
           '<technique sid="common"><lambert><diffuse><texture texture="samp" texcoord="UV"/></diffuse></lambert></technique>'
           '</profile_COMMON></effect></library_effects>'
           '<library_materials><material id="mat-material" name="mat"><instance_effect url="#fx"/></material></library_materials>'
           '<library_geometries><geometry id="g" name="shape"><mesh>' + arr("pos", V) + arr("nrm", NRM) + arr("uv", UV) +
           '<vertices id="vtx"><input semantic="POSITION" source="#pos"/></vertices>'
           '<polylist material="mat-material" count="3"><input semantic="VERTEX" source="#vtx" offset="0"/>'
           '<input semantic="NORMAL" source="#nrm" offset="1"/><input semantic="TEXCOORD" source="#uv" offset="2" set="0"/>'
           '<vcount>4 3 5</vcount><p>' + p + '</p></polylist>'
           '<triangles material="mat-material" count="1"><input semantic="VERTEX" source="#vtx" offset="0"/>'
           '<input semantic="NORMAL" source="#nrm" offset="1"/><input semantic="TEXCOORD" source="#uv" offset="2" set="0"/>'
           '<p>' + tri + '</p></triangles>'
           '</mesh></geometry></library_geometries></COLLADA>')
    (tmp_path / "m.dae").write_text(doc)
    a = assets.load_asset(str(tmp_path / "m.dae"))
    assert len(a.meshes) == 2 and a.triangle_count == 7
    check_mesh(a.meshes[0])
    assert np.array_equal(a.meshes[1].indices, [[0, 1, 2]])
    assert a.materials[0].name == "mat" and a.materials[0].textures == {"albedo": "textures/albedo.png"}
    assert a.meshes[0].material == 0


# ---- a minimal binary FBX 7.4 writer (node records as in the public format description)
def fbx_prop(v):
    if isinstance(v, bool):
        return b"C" + struct.pack("<?", v)
    if isinstance(v, int):
        return b"L" + struct.pack("<q", v)
    if isinstance(v, float):
        return b"D" + struct.pack("<d", v)
    if isinstance(v, str):
        b = v.encode()
        return b"S" + struct.pack("<I", len(b)) + b
    if isinstance(v, np.ndarray):
        code = {"float64": b"d", "int32": b"i"}[str(v.dtype)]
        raw = v.tobytes()
        z = zlib.compress(raw)
        return code + struct.pack("<III", v.size, 1, len(z)) + z  # zlib-encoded, like Blender's exporter
    raise TypeError(type(v))


def fbx_node(name, props=(), children=(), base=0):
    head_len = 13 + len(name)
    pbytes = b"".join(fbx_prop(p) for p in props)
    body = b""
    pos = base + head_len + len(pbytes)
    for c in children:
        cb = c(pos)
        body += cb
        pos += len(cb)
    if children:
        body += b"\x00" * 13
        pos += 13
    return struct.pack("<III", pos, len(props), len(pbytes)) + bytes([len(name)]) + name.encode() + pbytes + body


def N(name, props=(), children=()):
    return lambda base: fbx_node(name, props, children, base)


def write_fbx(path):
    pvi = []
    for p in POLYS:
        pvi += p[:-1] + [~p[-1]]
    geom = N("Geometry", [1001, "shape\x00\x01Geometry", "Mesh"], [
        N("Vertices", [V.reshape(-1).astype(np.float64)]),
        N("PolygonVertexIndex", [np.asarray(pvi, dtype=np.int32)]),
        N("LayerElementNormal", [0], [N("MappingInformationType", ["ByPolygonVertex"]), N("ReferenceInformationType", ["Direct"]),
                                      N("Normals", [NRM.reshape(-1).astype(np.float64)])]),
        N("LayerElementUV", [0], [N("MappingInformationType", ["ByPolygonVertex"]), N("ReferenceInformationType", ["IndexToDirect"]),
                                  N("UV", [UV[::-1].reshape(-1).astype(np.float64)]),
                                  N("UVIndex", [np.arange(len(CORNERS) - 1, -1, -1, dtype=np.int32)])]),
        N("LayerElementMaterial", [0], [N("MappingInformationType", ["ByPolygon"]), N("ReferenceInformationType", ["IndexToDirect"]),
                                        N("Materials", [np.asarray([0, 1, 0], dtype=np.int32)])]),
    ])
    objects = N("Objects", [], [
        geom,
        N("Model", [2001, "shape\x00\x01Model", "Mesh"]),
        N("Material", [3001, "first\x00\x01Material", ""]),
        N("Material", [3002, "second\x00\x01Material", ""]),
        N("Texture", [4001, "t\x00\x01Texture", ""], [N("RelativeFilename", ["texture\\base.png"])]),
    ])
    conns = N("Connections", [], [N("C", ["OO", 1001, 2001]), N("C", ["OO", 3001, 2001]), N("C", ["OO", 3002, 2001]),
                                   N("C", ["OP", 4001, 3001, "DiffuseColor"])])
    data = b"Kaydara FBX Binary  \x00\x1a\x00" + struct.pack("<I", 7400)
    for n in (objects, conns):
        data += n(len(data))
    data += b"\x00" * 13
    open(path, "wb").write(data)


def test_fbx_binary(tmp_path):
    write_fbx(str(tmp_path / "m.fbx"))
    a = assets.load_asset(str(tmp_path / "m.fbx"))
    assert [m.name for m in a.materials] == ["first", "second"]
    assert a.materials[0].textures == {"albedo": "texture/base.png"}
    assert len(a.meshes) == 2 and a.triangle_count == 6
    quad_pent, tri = a.meshes
    assert quad_pent.material == 0 and tri.material == 1
    sel = [0, 1, 2, 3, 7, 8, 9, 10, 11]
    assert np.allclose(quad_pent.positions, V[[CORNERS[k] for k in sel]].astype(np.float32))
    assert np.array_equal(quad_pent.indices, [[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7], [4, 7, 8]])
    assert np.allclose(quad_pent.uvs, np.stack([UV[sel, 0], 1.0 - UV[sel, 1]], axis=1).astype(np.float32))
    assert np.allclose(quad_pent.normals, NRM[sel].astype(np.float32))
    assert np.allclose(tri.positions, V[[1, 4, 2]].astype(np.float32)) and np.array_equal(tri.indices, [[0, 1, 2]])


def test_custom_model_object(tmp_path):
    write_fbx(str(tmp_path / "m.fbx"))
    m = CustomModel(str(tmp_path / "m.fbx"))
    assert m.getLoadSuccess() and len(m.meshes) == 2 and len(m.materials) == 2
    assert m.materials[0].albedoTexture is None  # the texture file does not exist: the constant stays (bool convention)
    sc = Scene()
    sc.addToScene("model", m)
    assert sc.flatten().triangle_count() == 6
    assert not CustomModel(str(tmp_path / "missing.dae")).getLoadSuccess()
    (tmp_path / "bad.fbx").write_bytes(b"not an fbx")
    assert not CustomModel(str(tmp_path / "bad.fbx")).getLoadSuccess()


REF = "/root/reference/res"


@pytest.mark.skipif(not os.path.exists(REF), reason="reference checkout absent (it is on the GPU box)")
def test_reference_model_files():
    dae = assets.load_asset(REF + "/lowpolypbrtrash/source/model.dae")
    assert dae.triangle_count == 3024 and len(dae.meshes) == 6
    assert [m.indices.shape[0] for m in dae.meshes] == [640, 896, 372, 372, 372, 372]
    fbx = assets.load_asset(REF + "/pbrsword/source/murasama.fbx")
    assert fbx.triangle_count == 8972 and sum(m.positions.shape[0] for m in fbx.meshes) == 18664
    assert fbx.materials[0].textures["albedo"] == "texture/Material.002_Base_Color.png"
    assert fbx.texture_directory() == REF + "/pbrsword"
    for m in dae.meshes + fbx.meshes:
        assert np.isfinite(m.positions).all() and m.indices.min() >= 0 and m.indices.max() < m.positions.shape[0]


def test_malformed_fbx_files_fail_cleanly(tmp_path):
    """Truncated and corrupted binary FBX files give getLoadSuccess() == False in both mirrors: no read past the buffer,
    no allocation the file cannot back, no unbounded recursion (the reference would hand them to assimp)."""
    import struct
    import subprocess
    from raytracer_amd.scene import CustomModel
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer_amd", "host", "rt_headless")
    good = str(tmp_path / "good.fbx")
    write_fbx(good)
    data = open(good, "rb").read()
    assert CustomModel(good).getLoadSuccess()
    cases = {}
    for cut in (28, 40, 100, len(data) // 2, len(data) - 30):
        cases[f"cut{cut}"] = data[:cut]
    bomb = bytearray(data)
    k = data.find(b"d", 60)                       # some array property: declare 2^31 elements
    i = data.find(b"Vertices")
    j = i + len(b"Vertices")
    assert data[j:j + 1] == b"d"
    bomb[j + 1:j + 5] = struct.pack("<I", 0x7FFFFFFF)
    cases["bomb"] = bytes(bomb)
    end_bad = bytearray(data)
    end_bad[27:31] = struct.pack("<I", 0xFFFFFF00)  # first record ends far outside the file
    cases["end"] = bytes(end_bad)
    nest = bytearray(data[:27])
    for _ in range(200):                          # 200 nested empty records, each "ending" at the end of the file
        nest += struct.pack("<III", 27 + 200 * 14 + 13, 0, 0) + b"\\x01N"
    nest += b"\\x00" * 13
    cases["nest"] = bytes(nest)
    for name, blob in cases.items():
        p = str(tmp_path / (name + ".fbx"))
        open(p, "wb").write(blob)
        assert not CustomModel(p).getLoadSuccess(), name
        r = subprocess.run([exe, "asset", p], capture_output=True, text=True, timeout=60)
        assert r.returncode == 3 and "load failed" in r.stdout, (name, r.returncode, r.stdout, r.stderr)
