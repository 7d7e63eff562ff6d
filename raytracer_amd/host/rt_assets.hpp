// rt_assets.hpp -- native asset flatteners of the C++ host mirror: COLLADA, Wavefront OBJ/MTL and
// binary FBX into the (positions, indices, uvs, normals, material) arrays the upload path takes.
//
// Replaces the assimp dependency of the reference's importer (src/CustomModel.cpp:47-194).  assimp
// is not vendored in the reference and its version is not pinned, so its triangulation and vertex
// order cannot be reproduced bit for bit: PARITY UNPINNED.  The flattening defined
// here is the same as raytracer_amd/assets.py (the two are compared array for array by
// tests/test_host_cpp.py): one mesh per primitive group in file order, one vertex per polygon
// corner, fan triangulation, v -> 1 - v, zeros for missing uvs, no normals when the file has none,
// node transforms not applied (the reference reads scene->mMeshes directly, :137-141).
#pragma once
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

namespace rth {
namespace assets {

struct FlatMesh {
    std::string name;
    std::vector<float> positions, uvs, normals; // normals empty when the file has none
    std::vector<int> indices;
    int material = 0;
};
struct FlatMaterial {
    std::string name;
    std::map<std::string, std::string> textures; // role (albedo, normal, roughness, metal, ao, emission) -> path as written
    bool hasAlbedo = false;
    float albedo[4] = { 0, 0, 0, 1 };
    std::string colladaId;
};
struct FlatAsset {
    std::string path;
    std::vector<FlatMesh> meshes;
    std::vector<FlatMaterial> materials;
    size_t triangleCount() const { size_t n = 0; for (auto& m : meshes) n += m.indices.size() / 3; return n; }
    // CustomModel.cpp:110-112: strip the file name, and a trailing "/source"
    std::string textureDirectory() const
    {
        std::string p = path;
        for (auto& ch : p) if (ch == '\\') ch = '/';
        size_t s = p.find_last_of('/');
        std::string d = s == std::string::npos ? std::string(".") : p.substr(0, s);
        if (d.size() >= 6 && d.compare(d.size() - 6, 6, "source") == 0 && (d.size() == 6 || d[d.size() - 7] == '/'))
            d = d.size() > 7 ? d.substr(0, d.size() - 7) : std::string(".");
        return d;
    }
};

inline int materialIndex(FlatAsset& a, const std::string& name)
{
    for (size_t i = 0; i < a.materials.size(); i++) if (a.materials[i].name == name) return (int)i;
    FlatMaterial m; m.name = name; a.materials.push_back(m);
    return (int)a.materials.size() - 1;
}
inline void fan(const std::vector<int>& counts, std::vector<int>& tris)
{
    int base = 0;
    for (int n : counts) { for (int k = 1; k + 1 < n; k++) { tris.push_back(base); tris.push_back(base + k); tris.push_back(base + k + 1); } base += n; }
}
inline bool readFile(const std::string& path, std::string& out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::ostringstream ss; ss << f.rdbuf(); out = ss.str();
    return true;
}
inline void splitFloats(const std::string& s, std::vector<double>& out)
{
    const char* p = s.c_str(); char* e = nullptr;
    for (;;) { const double v = strtod(p, &e); if (e == p) break; out.push_back(v); p = e; }
}
inline void splitInts(const std::string& s, std::vector<long long>& out)
{
    const char* p = s.c_str(); char* e = nullptr;
    for (;;) { const long long v = strtoll(p, &e, 10); if (e == p) break; out.push_back(v); p = e; }
}

// ------------------------------------------------------------------ minimal XML tree (COLLADA)
struct Xml {
    std::string tag, text;
    std::map<std::string, std::string> attr;
    std::vector<std::unique_ptr<Xml>> kids;
    const Xml* find(const char* t) const { for (auto& k : kids) if (k->tag == t) return k.get(); return nullptr; }
    std::string get(const char* a, const char* def = "") const { auto i = attr.find(a); return i == attr.end() ? std::string(def) : i->second; }
    std::string childText(const char* t) const { const Xml* k = find(t); return k ? k->text : std::string(); }
    void all(const char* t, std::vector<const Xml*>& out) const { for (auto& k : kids) { if (k->tag == t) out.push_back(k.get()); k->all(t, out); } }
};
inline std::string trim(const std::string& s)
{
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) a++;
    while (b > a && isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}
inline bool parseXml(const std::string& d, Xml& root)
{
    std::vector<Xml*> stack; stack.push_back(&root);
    size_t i = 0;
    while (i < d.size()) {
        if (d[i] != '<') { size_t j = d.find('<', i); if (j == std::string::npos) j = d.size(); stack.back()->text += d.substr(i, j - i); i = j; continue; }
        if (d.compare(i, 4, "<!--") == 0) { size_t j = d.find("-->", i); if (j == std::string::npos) return false; i = j + 3; continue; }
        if (d.compare(i, 2, "<?") == 0 || d.compare(i, 2, "<!") == 0) { size_t j = d.find('>', i); if (j == std::string::npos) return false; i = j + 1; continue; }
        size_t j = d.find('>', i);
        if (j == std::string::npos) return false;
        if (d[i + 1] == '/') { if (stack.size() > 1) stack.pop_back(); i = j + 1; continue; }
        const bool selfClose = d[j - 1] == '/';
        std::string body = d.substr(i + 1, j - i - 1 - (selfClose ? 1 : 0));
        auto el = std::make_unique<Xml>();
        size_t p = 0;
        while (p < body.size() && !isspace((unsigned char)body[p])) p++;
        el->tag = body.substr(0, p);
        size_t colon = el->tag.find(':');
        if (colon != std::string::npos) el->tag = el->tag.substr(colon + 1);
        while (p < body.size()) {
            while (p < body.size() && isspace((unsigned char)body[p])) p++;
            size_t eq = body.find('=', p);
            if (eq == std::string::npos) break;
            std::string key = trim(body.substr(p, eq - p));
            size_t q = body.find_first_of("\"'", eq);
            if (q == std::string::npos) break;
            size_t q2 = body.find(body[q], q + 1);
            if (q2 == std::string::npos) break;
            el->attr[key] = body.substr(q + 1, q2 - q - 1);
            p = q2 + 1;
        }
        Xml* raw = el.get();
        stack.back()->kids.push_back(std::move(el));
        if (!selfClose) stack.push_back(raw);
        i = j + 1;
    }
    return true;
}

inline bool loadDae(const std::string& path, FlatAsset& a)
{
    std::string d;
    if (!readFile(path, d)) return false;
    Xml root;
    if (!parseXml(d, root)) return false;
    a.path = path;
    std::map<std::string, std::string> images;
    { std::vector<const Xml*> v; root.all("image", v); for (auto* im : v) images[im->get("id")] = trim(im->childText("init_from")); }
    struct Info { std::map<std::string, std::string> tex; bool hasCol = false; float col[4] = { 0, 0, 0, 1 }; };
    std::map<std::string, Info> effects;
    { std::vector<const Xml*> v; root.all("effect", v);
      for (auto* ef : v) {
          std::map<std::string, std::string> params;
          std::vector<const Xml*> nps; ef->all("newparam", nps);
          for (auto* np : nps) {
              if (const Xml* s = np->find("surface")) params[np->get("sid")] = trim(s->childText("init_from"));
              else if (const Xml* s2 = np->find("sampler2D")) params[np->get("sid")] = trim(s2->childText("source"));
          }
          Info info;
          const char* roles[3][2] = { { "albedo", "diffuse" }, { "emission", "emission" }, { "normal", "bump" } };
          for (auto& r : roles) {
              std::vector<const Xml*> els; ef->all(r[1], els);
              if (els.empty()) continue;
              if (const Xml* tex = els[0]->find("texture")) {
                  std::string ref = tex->get("texture");
                  for (int k = 0; k < 3; k++) { auto it = params.find(ref); if (it != params.end()) ref = it->second; }
                  auto im = images.find(ref);
                  info.tex[r[0]] = im == images.end() ? ref : im->second;
              }
              if (std::string(r[0]) == "albedo") { std::vector<double> c; splitFloats(els[0]->childText("color"), c); if (c.size() >= 3) { info.hasCol = true; for (size_t k = 0; k < 4 && k < c.size(); k++) info.col[k] = (float)c[k]; } }
          }
          effects[ef->get("id")] = info;
      } }
    { std::vector<const Xml*> v; root.all("material", v);
      for (auto* m : v) {
          FlatMaterial fm; fm.name = m->get("name").empty() ? m->get("id") : m->get("name"); fm.colladaId = m->get("id");
          if (const Xml* ie = m->find("instance_effect")) {
              std::string url = ie->get("url"); if (!url.empty() && url[0] == '#') url = url.substr(1);
              auto it = effects.find(url);
              if (it != effects.end()) { fm.textures = it->second.tex; fm.hasAlbedo = it->second.hasCol; memcpy(fm.albedo, it->second.col, sizeof(fm.albedo)); }
          }
          a.materials.push_back(fm);
      } }
    auto matIndex = [&](const std::string& sym) {
        for (size_t i = 0; i < a.materials.size(); i++) if (a.materials[i].colladaId == sym || a.materials[i].name == sym) return (int)i;
        return materialIndex(a, sym.empty() ? std::string("MissingName") : sym);
    };
    std::vector<const Xml*> geoms; root.all("geometry", geoms);
    for (auto* g : geoms) {
        const Xml* mesh = g->find("mesh");
        if (!mesh) continue;
        struct Src { std::vector<double> v; int stride = 1; };
        std::map<std::string, Src> sources;
        std::map<std::string, std::map<std::string, std::string>> vertices;
        for (auto& k : mesh->kids) {
            if (k->tag == "source") {
                Src s;
                if (const Xml* fa = k->find("float_array")) splitFloats(fa->text, s.v);
                if (const Xml* tc = k->find("technique_common")) if (const Xml* ac = tc->find("accessor")) s.stride = std::max(1, atoi(ac->get("stride", "1").c_str()));
                sources["#" + k->get("id")] = s;
            } else if (k->tag == "vertices") {
                for (auto& in : k->kids) if (in->tag == "input") vertices["#" + k->get("id")][in->get("semantic")] = in->get("source");
            }
        }
        for (auto& prim : mesh->kids) {
            if (prim->tag != "triangles" && prim->tag != "polylist" && prim->tag != "polygons") continue;
            struct In { std::string sem, src; int off, set; };
            std::vector<In> inputs;
            int stride = 0;
            for (auto& in : prim->kids) if (in->tag == "input") { In x{ in->get("semantic"), in->get("source"), atoi(in->get("offset", "0").c_str()), atoi(in->get("set", "0").c_str()) }; stride = std::max(stride, x.off + 1); inputs.push_back(x); }
            if (inputs.empty()) continue;
            std::vector<long long> p; std::vector<int> counts;
            if (prim->tag == "polygons") {
                for (auto& k : prim->kids) if (k->tag == "p") { size_t before = p.size(); splitInts(k->text, p); counts.push_back((int)((p.size() - before) / stride)); }
            } else {
                splitInts(prim->childText("p"), p);
                if (prim->tag == "triangles") counts.assign(p.size() / (3 * (size_t)stride), 3);
                else { std::vector<long long> vc; splitInts(prim->childText("vcount"), vc); for (auto c : vc) counts.push_back((int)c); }
            }
            const size_t nCorner = p.size() / stride;
            const Src *pos = nullptr, *uv = nullptr, *nrm = nullptr; int posOff = 0, uvOff = 0, nrmOff = 0;
            for (auto& in : inputs) {
                if (in.sem == "VERTEX") {
                    for (auto& vs : vertices[in.src]) {
                        if (vs.first == "POSITION") { pos = &sources[vs.second]; posOff = in.off; }
                        else if (vs.first == "NORMAL" && !nrm) { nrm = &sources[vs.second]; nrmOff = in.off; }
                        else if (vs.first == "TEXCOORD" && !uv) { uv = &sources[vs.second]; uvOff = in.off; }
                    }
                } else if (in.sem == "NORMAL") { nrm = &sources[in.src]; nrmOff = in.off; }
                else if (in.sem == "TEXCOORD" && (!uv || in.set == 0)) { uv = &sources[in.src]; uvOff = in.off; }
            }
            if (!pos) continue;
            FlatMesh fm; fm.name = g->get("name").empty() ? g->get("id") : g->get("name"); fm.material = matIndex(prim->get("material"));
            std::vector<int> kept;
            size_t corner = 0;
            for (int c : counts) {
                if (corner + (size_t)c > nCorner) break;
                if (c >= 3) {
                    kept.push_back(c);
                    for (int k = 0; k < c; k++) {
                        const long long* t = &p[(corner + k) * stride];
                        auto fetch = [](const Src* s, long long i, int comp) { const size_t o = (size_t)i * s->stride + comp; return o < s->v.size() ? (float)s->v[o] : 0.0f; };
                        for (int q = 0; q < 3; q++) fm.positions.push_back(fetch(pos, t[posOff], q));
                        if (uv) { fm.uvs.push_back(fetch(uv, t[uvOff], 0)); fm.uvs.push_back((float)(1.0 - (double)(uv->v.size() > (size_t)t[uvOff] * uv->stride + 1 ? uv->v[(size_t)t[uvOff] * uv->stride + 1] : 0.0))); }
                        else { fm.uvs.push_back(0.0f); fm.uvs.push_back(0.0f); }
                        if (nrm) for (int q = 0; q < 3; q++) fm.normals.push_back(fetch(nrm, t[nrmOff], q));
                    }
                }
                corner += c;
            }
            fan(kept, fm.indices);
            a.meshes.push_back(std::move(fm));
        }
    }
    if (a.materials.empty()) materialIndex(a, "MissingName0");
    return true;
}

// ------------------------------------------------------------------ OBJ / MTL
inline void loadMtl(const std::string& path, FlatAsset& a)
{
    std::ifstream f(path);
    if (!f) return;
    static const std::map<std::string, std::string> roles = { { "map_kd", "albedo" }, { "map_bump", "normal" }, { "bump", "normal" }, { "norm", "normal" },
        { "map_ns", "roughness" }, { "map_pr", "roughness" }, { "map_refl", "metal" }, { "refl", "metal" }, { "map_pm", "metal" }, { "map_ke", "emission" }, { "map_ka", "ao" } };
    std::string line; int cur = -1;
    while (std::getline(f, line)) {
        std::string t = trim(line);
        if (t.empty() || t[0] == '#') continue;
        size_t sp = t.find(' ');
        std::string key = t.substr(0, sp), rest = sp == std::string::npos ? std::string() : trim(t.substr(sp + 1));
        for (auto& ch : key) ch = (char)tolower((unsigned char)ch);
        if (key == "newmtl") cur = materialIndex(a, rest);
        else if (cur < 0) continue;
        else if (key == "kd") { std::vector<double> c; splitFloats(rest, c); if (c.size() >= 3) { a.materials[cur].hasAlbedo = true; for (int k = 0; k < 3; k++) a.materials[cur].albedo[k] = (float)c[k]; a.materials[cur].albedo[3] = 1.0f; } }
        else if (roles.count(key)) {
            // options such as "-bm 1.0" precede the file name; the name may contain blanks
            std::vector<std::string> toks; { std::string w; std::istringstream ss(rest); size_t pos = 0; while (pos <= rest.size()) { size_t e = rest.find(' ', pos); if (e == std::string::npos) e = rest.size(); toks.push_back(rest.substr(pos, e - pos)); pos = e + 1; } }
            size_t k = 0;
            while (k < toks.size() && !toks[k].empty() && toks[k][0] == '-') k += 2;
            std::string name;
            for (; k < toks.size(); k++) { if (!name.empty()) name += " "; name += toks[k]; }
            a.materials[cur].textures[roles.at(key)] = name;
        }
    }
}

inline bool loadObj(const std::string& path, FlatAsset& a)
{
    std::ifstream f(path);
    if (!f) return false;
    a.path = path;
    std::string base = path; { size_t s = base.find_last_of("/\\"); base = s == std::string::npos ? std::string(".") : base.substr(0, s); }
    std::vector<double> v, vt, vn;
    struct Corner { long long v, t, n; };
    struct Group { std::string mat; std::vector<std::vector<Corner>> polys; };
    std::vector<Group> groups;
    std::string line;
    while (std::getline(f, line)) {
        std::string t = trim(line);
        if (t.empty() || t[0] == '#') continue;
        size_t sp = t.find(' ');
        std::string key = t.substr(0, sp), rest = sp == std::string::npos ? std::string() : trim(t.substr(sp + 1));
        if (key == "v") { std::vector<double> c; splitFloats(rest, c); c.resize(3, 0.0); v.insert(v.end(), c.begin(), c.begin() + 3); }
        else if (key == "vt") { std::vector<double> c; splitFloats(rest, c); c.resize(2, 0.0); vt.insert(vt.end(), c.begin(), c.begin() + 2); }
        else if (key == "vn") { std::vector<double> c; splitFloats(rest, c); c.resize(3, 0.0); vn.insert(vn.end(), c.begin(), c.begin() + 3); }
        else if (key == "mtllib") loadMtl(base + "/" + rest, a);
        else if (key == "usemtl") groups.push_back({ rest, {} });
        else if (key == "f") {
            if (groups.empty()) groups.push_back({ "MissingName0", {} });
            std::vector<Corner> poly;
            std::istringstream ss(rest); std::string tok;
            while (ss >> tok) {
                long long idx[3] = { 0, 0, 0 }; int k = 0; size_t p = 0;
                while (k < 3 && p <= tok.size()) { size_t e = tok.find('/', p); if (e == std::string::npos) e = tok.size(); if (e > p) idx[k] = atoll(tok.substr(p, e - p).c_str()); k++; p = e + 1; }
                const long long nv = (long long)v.size() / 3, nt = (long long)vt.size() / 2, nn = (long long)vn.size() / 3;
                poly.push_back({ idx[0] > 0 ? idx[0] - 1 : nv + idx[0], idx[1] > 0 ? idx[1] - 1 : (idx[1] < 0 ? nt + idx[1] : -1), idx[2] > 0 ? idx[2] - 1 : (idx[2] < 0 ? nn + idx[2] : -1) });
            }
            if (poly.size() >= 3) groups.back().polys.push_back(poly);
        }
    }
    for (auto& g : groups) {
        if (g.polys.empty()) continue;
        FlatMesh fm; fm.name = g.mat; fm.material = materialIndex(a, g.mat);
        bool allT = !vt.empty(), allN = !vn.empty();
        for (auto& p : g.polys) for (auto& c : p) { if (c.t < 0) allT = false; if (c.n < 0) allN = false; }
        // an index past the arrays: the file is refused (assimp: "vertex index out of range"; the Python mirror raises)
        for (auto& p : g.polys) for (auto& c : p) {
            if (c.v < 0 || (size_t)c.v * 3 + 2 >= v.size()) return false;
            if (allT && (size_t)c.t * 2 + 1 >= vt.size()) return false;
            if (allN && (size_t)c.n * 3 + 2 >= vn.size()) return false;
        }
        std::vector<int> counts;
        for (auto& p : g.polys) {
            counts.push_back((int)p.size());
            for (auto& c : p) {
                for (int q = 0; q < 3; q++) fm.positions.push_back((float)v[c.v * 3 + q]);
                if (allT) { fm.uvs.push_back((float)vt[c.t * 2]); fm.uvs.push_back((float)(1.0 - vt[c.t * 2 + 1])); } else { fm.uvs.push_back(0.0f); fm.uvs.push_back(0.0f); }
                if (allN) for (int q = 0; q < 3; q++) fm.normals.push_back((float)vn[c.n * 3 + q]);
            }
        }
        fan(counts, fm.indices);
        a.meshes.push_back(std::move(fm));
    }
    if (a.materials.empty()) materialIndex(a, "MissingName0");
    return true;
}

// ------------------------------------------------------------------ binary FBX 7.x
struct FbxProp {
    char type = 0;
    long long i = 0; double d = 0;
    std::string s;
    std::vector<double> fa; std::vector<long long> ia;
};
struct FbxNode {
    std::string name;
    std::vector<FbxProp> props;
    std::vector<std::unique_ptr<FbxNode>> kids;
    const FbxNode* find(const char* n) const { for (auto& k : kids) if (k->name == n) return k.get(); return nullptr; }
};
template <class T> inline T rd(const std::string& d, size_t pos) { T v; memcpy(&v, d.data() + pos, sizeof(T)); return v; }

// Every read is checked against the buffer, array sizes against what the bytes present can hold (a deflate stream expands
// at most ~1032-fold), the record's end against its start and the file, and the nesting against a fixed depth: a malformed
// or truncated file gives false, never a read past the buffer, an allocation the file cannot back, or unbounded recursion.
constexpr int kFbxMaxDepth = 64;
inline bool fbxRead(const std::string& d, size_t& pos, bool wide, std::unique_ptr<FbxNode>& out, int depth = 0)
{
    const size_t head = wide ? 24 : 12;
    if (depth > kFbxMaxDepth || pos + head + 1 > d.size()) return false;
    uint64_t end, nprop;
    if (wide) { end = rd<uint64_t>(d, pos); nprop = rd<uint64_t>(d, pos + 8); } else { end = rd<uint32_t>(d, pos); nprop = rd<uint32_t>(d, pos + 4); }
    pos += head;
    const unsigned nlen = (unsigned char)d[pos++];
    if (end == 0) { out.reset(); return true; }
    if (end > d.size() || end < pos || pos + nlen > end) return false;
    out = std::make_unique<FbxNode>();
    out->name = d.substr(pos, nlen); pos += nlen;
    for (uint64_t k = 0; k < nprop; k++) {
        if (pos >= d.size()) return false;
        FbxProp p; p.type = d[pos++];
        const size_t left = d.size() - pos;
        const size_t need = p.type == 'Y' ? 2 : p.type == 'C' ? 1 : (p.type == 'I' || p.type == 'F' || p.type == 'S' || p.type == 'R') ? 4
                          : (p.type == 'L' || p.type == 'D') ? 8 : 12;
        if (left < need) return false;
        switch (p.type) {
        case 'Y': p.i = rd<int16_t>(d, pos); pos += 2; break;
        case 'C': p.i = d[pos] != 0; pos += 1; break;
        case 'I': p.i = rd<int32_t>(d, pos); pos += 4; break;
        case 'L': p.i = rd<int64_t>(d, pos); pos += 8; break;
        case 'F': p.d = rd<float>(d, pos); pos += 4; break;
        case 'D': p.d = rd<double>(d, pos); pos += 8; break;
        case 'S': case 'R': { const uint32_t n = rd<uint32_t>(d, pos); pos += 4; if (pos + n > d.size()) return false; p.s = d.substr(pos, n); pos += n; break; }
        case 'f': case 'd': case 'l': case 'i': case 'b': {
            const uint32_t n = rd<uint32_t>(d, pos), enc = rd<uint32_t>(d, pos + 4), clen = rd<uint32_t>(d, pos + 8);
            pos += 12;
            if (pos + clen > d.size()) return false;
            const size_t esz = p.type == 'f' || p.type == 'i' ? 4 : (p.type == 'b' ? 1 : 8);
            if ((uint64_t)n * esz > (enc == 1 ? (uint64_t)clen * 1100u + 64u : (uint64_t)clen)) return false;
            std::vector<unsigned char> raw((size_t)n * esz);
            if (enc == 1) { uLongf rl = (uLongf)raw.size(); if (uncompress(raw.data(), &rl, (const Bytef*)d.data() + pos, clen) != Z_OK || rl != raw.size()) return false; }
            else { if (clen < raw.size()) return false; memcpy(raw.data(), d.data() + pos, raw.size()); }
            pos += clen;
            for (uint32_t q = 0; q < n; q++) {
                if (p.type == 'f') { float x; memcpy(&x, &raw[q * 4], 4); p.fa.push_back(x); }
                else if (p.type == 'd') { double x; memcpy(&x, &raw[q * 8], 8); p.fa.push_back(x); }
                else if (p.type == 'i') { int32_t x; memcpy(&x, &raw[q * 4], 4); p.ia.push_back(x); }
                else if (p.type == 'l') { int64_t x; memcpy(&x, &raw[q * 8], 8); p.ia.push_back(x); }
                else p.ia.push_back(raw[q]);
            }
            break;
        }
        default: return false;
        }
        out->props.push_back(std::move(p));
    }
    while (pos < end) {
        std::unique_ptr<FbxNode> c;
        if (!fbxRead(d, pos, wide, c, depth + 1)) return false;
        if (!c) break;
        out->kids.push_back(std::move(c));
    }
    pos = end;
    return true;
}

inline std::string fbxClean(const std::string& s) { return s.substr(0, s.find('\0')); }

inline bool loadFbx(const std::string& path, FlatAsset& a)
{
    std::string d;
    if (!readFile(path, d) || d.size() < 27 || d.compare(0, 20, "Kaydara FBX Binary  ") != 0) return false;
    a.path = path;
    const bool wide = rd<uint32_t>(d, 23) >= 7500;
    size_t pos = 27;
    std::vector<std::unique_ptr<FbxNode>> top;
    while (pos + (wide ? 25 : 13) < d.size()) {
        std::unique_ptr<FbxNode> n;
        if (!fbxRead(d, pos, wide, n)) return false;
        if (!n) break;
        top.push_back(std::move(n));
    }
    const FbxNode *objects = nullptr, *conns = nullptr;
    for (auto& n : top) { if (n->name == "Objects") objects = n.get(); else if (n->name == "Connections") conns = n.get(); }
    if (!objects) return false;
    std::vector<long long> matIds; std::map<long long, std::string> texs; std::map<long long, const FbxNode*> models; std::vector<const FbxNode*> geoms;
    for (auto& n : objects->kids) {
        if (n->props.size() < 2) continue;
        if (n->name == "Material") { matIds.push_back(n->props[0].i); FlatMaterial m; m.name = fbxClean(n->props[1].s); a.materials.push_back(m); }
        else if (n->name == "Texture") { const FbxNode* fn = n->find("RelativeFilename"); if (!fn) fn = n->find("FileName"); std::string s = fn && !fn->props.empty() ? fn->props[0].s : std::string(); for (auto& ch : s) if (ch == '\\') ch = '/'; texs[n->props[0].i] = s; }
        else if (n->name == "Model") models[n->props[0].i] = n.get();
        else if (n->name == "Geometry" && n->find("Vertices")) geoms.push_back(n.get());
    }
    auto matSlot = [&](long long id) { for (size_t i = 0; i < matIds.size(); i++) if (matIds[i] == id) return (int)i; return -1; };
    static const std::map<std::string, std::string> roles = { { "diffusecolor", "albedo" }, { "basecolor", "albedo" }, { "normalmap", "normal" }, { "bump", "normal" },
        { "emissivecolor", "emission" }, { "shininessexponent", "roughness" }, { "shininess", "roughness" }, { "reflectionfactor", "metal" }, { "specularcolor", "metal" }, { "ambientcolor", "ao" } };
    std::map<long long, long long> geomModel; std::map<long long, std::vector<long long>> modelMats;
    if (conns) for (auto& c : conns->kids) {
        if (c->props.size() < 3) continue;
        const std::string kind = c->props[0].s; const long long src = c->props[1].i, dst = c->props[2].i;
        if (kind == "OO" && matSlot(src) >= 0 && models.count(dst)) modelMats[dst].push_back(src);
        else if (kind == "OO" && models.count(dst)) geomModel[src] = dst;
        else if (kind == "OP" && texs.count(src) && matSlot(dst) >= 0 && c->props.size() >= 4) {
            std::string pn = c->props[3].s; for (auto& ch : pn) ch = (char)tolower((unsigned char)ch);
            auto r = roles.find(pn);
            if (r != roles.end() && !a.materials[matSlot(dst)].textures.count(r->second)) a.materials[matSlot(dst)].textures[r->second] = texs[src];
        }
    }
    for (auto* g : geoms) {
        const FbxNode* vn = g->find("Vertices"); const FbxNode* pn = g->find("PolygonVertexIndex");
        if (!vn || !pn || vn->props.empty() || pn->props.empty()) continue;
        const std::vector<double>& verts = vn->props[0].fa; const std::vector<long long>& pvi = pn->props[0].ia;
        std::vector<int> counts; std::vector<long long> cornerVertex; std::vector<int> polyOfCorner;
        { int run = 0; std::vector<long long> tmp;
          for (long long x : pvi) { run++; tmp.push_back(x < 0 ? ~x : x); if (x < 0) { for (long long y : tmp) { cornerVertex.push_back(y); polyOfCorner.push_back((int)counts.size()); } counts.push_back(run); run = 0; tmp.clear(); } } }
        const size_t nCorner = cornerVertex.size();
        auto layer = [&](const char* tag, const char* dataTag, const char* indexTag, int width, std::vector<double>& out) -> bool {
            const FbxNode* el = g->find(tag);
            if (!el) return false;
            const FbxNode* dn = el->find(dataTag);
            if (!dn || dn->props.empty()) return false;
            const std::vector<double>& data = dn->props[0].fa;
            const size_t nData = data.size() / width;
            if (nData == 0) return false;
            const FbxNode* mn = el->find("MappingInformationType"); const FbxNode* rn = el->find("ReferenceInformationType");
            const std::string mapping = mn && !mn->props.empty() ? mn->props[0].s : "ByPolygonVertex", ref = rn && !rn->props.empty() ? rn->props[0].s : "Direct";
            const FbxNode* in = el->find(indexTag);
            const bool indexed = (ref == "IndexToDirect" || ref == "Index") && in && !in->props.empty();
            out.resize(nCorner * width);
            for (size_t c = 0; c < nCorner; c++) {
                long long key;
                if (mapping == "ByPolygonVertex") key = (long long)c;
                else if (mapping == "ByVertice" || mapping == "ByVertex" || mapping == "ByControlPoint") key = cornerVertex[c];
                else if (mapping == "ByPolygon") key = polyOfCorner[c];
                else if (mapping == "AllSame") key = 0;
                else return false;
                if (indexed) { const auto& idx = in->props[0].ia; key = key >= 0 && (size_t)key < idx.size() ? idx[key] : 0; }
                if (key < 0) key = 0;
                if ((size_t)key >= nData) key = (long long)nData - 1;
                for (int q = 0; q < width; q++) out[c * width + q] = data[key * width + q];
            }
            return true;
        };
        std::vector<double> nrm, uv;
        const bool hasN = layer("LayerElementNormal", "Normals", "NormalsIndex", 3, nrm);
        const bool hasUv = layer("LayerElementUV", "UV", "UVIndex", 2, uv);
        std::vector<long long> polyMat(counts.size(), 0);
        if (const FbxNode* lm = g->find("LayerElementMaterial")) if (const FbxNode* mm = lm->find("Materials")) if (!mm->props.empty()) {
            const auto& m = mm->props[0].ia;
            const FbxNode* mn = lm->find("MappingInformationType");
            const std::string mapping = mn && !mn->props.empty() ? mn->props[0].s : "AllSame";
            if (mapping == "ByPolygon" && m.size() == counts.size()) polyMat = m;
            else if (!m.empty()) std::fill(polyMat.begin(), polyMat.end(), m[0]);
        }
        const long long gid = g->props[0].i;
        const long long model = geomModel.count(gid) ? geomModel[gid] : -1;
        const std::vector<long long>& local = modelMats[model];
        std::string name = fbxClean(g->props[1].s);
        if (name.empty() && models.count(model)) name = fbxClean(models[model]->props[1].s);
        if (name.empty()) name = "mesh";
        std::vector<long long> distinct(polyMat.begin(), polyMat.end());
        std::sort(distinct.begin(), distinct.end()); distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
        for (long long lmIdx : distinct) {
            FlatMesh fm; fm.name = name;
            int gm = lmIdx >= 0 && (size_t)lmIdx < local.size() ? matSlot(local[lmIdx]) : -1;
            if (gm < 0) gm = materialIndex(a, "MissingName" + std::to_string(lmIdx));
            fm.material = gm;
            std::vector<int> kept;
            size_t c = 0;
            for (size_t pI = 0; pI < counts.size(); pI++) {
                if (polyMat[pI] == lmIdx) {
                    kept.push_back(counts[pI]);
                    for (int k = 0; k < counts[pI]; k++) {
                        const size_t cc = c + k; const long long vi = cornerVertex[cc];
                        for (int q = 0; q < 3; q++) fm.positions.push_back(vi >= 0 && (size_t)vi * 3 + q < verts.size() ? (float)verts[vi * 3 + q] : 0.0f);
                        if (hasUv) { fm.uvs.push_back((float)uv[cc * 2]); fm.uvs.push_back((float)(1.0 - uv[cc * 2 + 1])); } else { fm.uvs.push_back(0.0f); fm.uvs.push_back(0.0f); }
                        if (hasN) for (int q = 0; q < 3; q++) fm.normals.push_back((float)nrm[cc * 3 + q]);
                    }
                }
                c += counts[pI];
            }
            fan(kept, fm.indices);
            a.meshes.push_back(std::move(fm));
        }
    }
    if (a.materials.empty()) materialIndex(a, "MissingName0");
    return true;
}

inline bool loadAsset(const std::string& path, FlatAsset& a)
{
    std::string ext; { size_t dot = path.find_last_of('.'); if (dot != std::string::npos) ext = path.substr(dot); for (auto& ch : ext) ch = (char)tolower((unsigned char)ch); }
    try {
        if (ext == ".dae") return loadDae(path, a);
        if (ext == ".obj") return loadObj(path, a);
        if (ext == ".fbx") return loadFbx(path, a);
    } catch (...) { return false; }
    return false;
}

} // namespace assets
} // namespace rth
