// obj_loader.cpp — see obj_loader.h.
#include "obj_loader.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace okh {

namespace {

struct MtlMaps {
    std::string kd, pr, pm, ks, norm, disp;
};

std::string dirOf(const std::string& path) {
    size_t s = path.find_last_of("/\\");
    return s == std::string::npos ? std::string() : path.substr(0, s + 1);
}
std::string restOfLine(std::istringstream& in) {  // the last token of a map statement is the file name (options precede it)
    std::string tok, last;
    while (in >> tok) last = tok;
    return last;
}

void loadMtl(const std::string& path, std::map<std::string, MtlMaps>& out) {
    std::ifstream f(path);
    if (!f) return;
    std::string line, cur;
    while (std::getline(f, line)) {
        std::istringstream in(line);
        std::string key;
        if (!(in >> key) || key[0] == '#') continue;
        if (key == "newmtl") {
            in >> cur;
            out[cur];
        } else if (!cur.empty()) {
            MtlMaps& m = out[cur];
            if (key == "map_Kd") m.kd = restOfLine(in);              // aiTextureType_DIFFUSE
            else if (key == "map_Pr") m.pr = restOfLine(in);         // aiTextureType_DIFFUSE_ROUGHNESS
            else if (key == "map_Pm") m.pm = restOfLine(in);         // aiTextureType_METALNESS
            else if (key == "map_Ks") m.ks = restOfLine(in);         // aiTextureType_SPECULAR
            else if (key == "norm" || key == "map_Kn") m.norm = restOfLine(in);  // aiTextureType_NORMALS
            else if (key == "disp" || key == "map_disp") m.disp = restOfLine(in);  // aiTextureType_DISPLACEMENT
            // map_bump / bump land in aiTextureType_HEIGHT, which Importer.cpp never queries
        }
    }
}

struct Idx {
    int v = 0, t = 0, n = 0;
};
Idx parseIdx(const std::string& tok, int nv, int nt, int nn) {
    Idx r;
    const char* s = tok.c_str();
    char* e;
    r.v = (int)std::strtol(s, &e, 10);
    if (*e == '/') {
        ++e;
        if (*e != '/') r.t = (int)std::strtol(e, &e, 10);
        if (*e == '/') r.n = (int)std::strtol(e + 1, &e, 10);
    }
    if (r.v < 0) r.v += nv + 1;  // negative = relative to the end
    if (r.t < 0) r.t += nt + 1;
    if (r.n < 0) r.n += nn + 1;
    return r;
}

}  // namespace

bool loadObjects(const std::string& filePath, std::vector<ObjObject>& outObjects, float scale, std::string& error) {
    std::ifstream f(filePath);
    if (!f) {
        error = "cannot open " + filePath;
        return false;
    }
    std::vector<Vec3> V, N;
    std::vector<float> T;  // u, v pairs
    std::map<std::string, MtlMaps> mtl;
    std::map<std::string, size_t> meshOf;  // "group\nmaterial" -> outObjects index
    std::string group = "default", material;
    ObjObject* cur = nullptr;
    auto select = [&]() {
        std::string key = group + "\n" + material;
        auto it = meshOf.find(key);
        if (it == meshOf.end()) {
            meshOf[key] = outObjects.size();
            outObjects.emplace_back();
            outObjects.back().name = group;
            const MtlMaps& m = mtl[material];
            outObjects.back().albedoTexturePath = m.kd;
            outObjects.back().roughnessTexturePath = m.pr;
            outObjects.back().metallicTexturePath = m.pm;
            outObjects.back().specularTexturePath = m.ks;
            outObjects.back().normalTexturePath = !m.norm.empty() ? m.norm : m.disp;  // NORMALS, else DISPLACEMENT (Importer.cpp:110-113)
            cur = &outObjects.back();
        } else {
            cur = &outObjects[it->second];
        }
    };
    std::string line;
    size_t lineNo = 0;
    while (std::getline(f, line)) {
        ++lineNo;
        std::istringstream in(line);
        std::string key;
        if (!(in >> key) || key[0] == '#') continue;
        if (key == "v") {
            Vec3 p;
            in >> p.x >> p.y >> p.z;
            V.push_back(p);
        } else if (key == "vn") {
            Vec3 p;
            in >> p.x >> p.y >> p.z;
            N.push_back(p);
        } else if (key == "vt") {
            float u = 0, v = 0;
            in >> u >> v;
            T.push_back(u);
            T.push_back(v);
        } else if (key == "mtllib") {
            std::string name;
            in >> name;
            if (!name.empty()) loadMtl(dirOf(filePath) + name, mtl);  // (a bare "mtllib" would name the directory itself)
        } else if (key == "usemtl") {
            in >> material;
            cur = nullptr;
        } else if (key == "o" || key == "g") {
            group.clear();
            in >> group;
            if (group.empty()) group = "default";
            cur = nullptr;
        } else if (key == "f") {
            std::vector<Idx> poly;
            std::string tok;
            while (in >> tok) poly.push_back(parseIdx(tok, (int)V.size(), (int)T.size() / 2, (int)N.size()));
            if (poly.size() < 3) continue;
            for (const Idx& i : poly)
                if (i.v < 1 || i.v > (int)V.size() || i.t < 0 || i.t > (int)T.size() / 2 || i.n < 0 || i.n > (int)N.size()) {  // 0 = absent; relative indices are resolved by parseIdx
                    error = "index out of range at line " + std::to_string(lineNo);
                    return false;
                }
            if (!cur) select();
            for (size_t k = 1; k + 1 < poly.size(); ++k) {  // Triangulate: fan
                const Idx tri[3] = {poly[0], poly[k], poly[k + 1]};
                Vec3 p[3], n[3];
                float uv[3][2];
                for (int c = 0; c < 3; ++c) {
                    p[c] = V[tri[c].v - 1] * scale;
                    uv[c][0] = tri[c].t ? T[2 * (tri[c].t - 1)] : 0.f;
                    uv[c][1] = tri[c].t ? T[2 * (tri[c].t - 1) + 1] : 0.f;
                }
                Vec3 fn = cross(p[1] - p[0], p[2] - p[0]);
                float fl = std::sqrt(dot(fn, fn));
                fn = fl > 0.f ? fn * (1.f / fl) : Vec3(0.f, 1.f, 0.f);
                for (int c = 0; c < 3; ++c) n[c] = tri[c].n ? N[tri[c].n - 1] : fn;
                // CalcTangentSpace: dP = T du + B dv on the (right-handed, unflipped) input
                Vec3 e1 = p[1] - p[0], e2 = p[2] - p[0];
                float du1 = uv[1][0] - uv[0][0], dv1 = uv[1][1] - uv[0][1], du2 = uv[2][0] - uv[0][0], dv2 = uv[2][1] - uv[0][1];
                float det = du1 * dv2 - du2 * dv1;
                Vec3 tan(0.f, 0.f, 0.f), bit(0.f, 0.f, 0.f);
                const bool hasUV = tri[0].t && tri[1].t && tri[2].t && det != 0.f;
                if (hasUV) {
                    float r = 1.f / det;
                    tan = (e1 * dv2 - e2 * dv1) * r;
                    bit = (e2 * du1 - e1 * du2) * r;
                }
                Triangle tp;
                TriangleInfo ti;
                for (int c = 0; c < 3; ++c) {
                    Vec3 nn = n[c], t2 = tan, b2 = bit;
                    if (hasUV) {  // Gram-Schmidt against the vertex normal
                        t2 = t2 - nn * dot(nn, t2);
                        b2 = b2 - nn * dot(nn, b2);
                        float tl = std::sqrt(dot(t2, t2)), bl = std::sqrt(dot(b2, b2));
                        if (tl > 0.f) t2 = t2 * (1.f / tl);
                        if (bl > 0.f) b2 = b2 * (1.f / bl);
                    }
                    // ConvertToLeftHanded: MakeLeftHanded (z negated), FlipUVs (v -> 1 - v), FlipWindingOrder (a,b,c -> c,b,a)
                    const int dst = 2 - c;
                    tp.position[dst] = Vec3(p[c].x, p[c].y, -p[c].z);
                    VertexInfo& vi = ti.vertexInfo[dst];
                    vi.normal[0] = nn.x; vi.normal[1] = nn.y; vi.normal[2] = -nn.z;
                    vi.uv[0] = uv[c][0]; vi.uv[1] = tri[c].t ? 1.f - uv[c][1] : 0.f;  // no uv channel -> (0,0) (Importer.cpp:43)
                    vi.tangent[0] = t2.x; vi.tangent[1] = t2.y; vi.tangent[2] = -t2.z;
                    vi.bitangent[0] = b2.x; vi.bitangent[1] = b2.y; vi.bitangent[2] = -b2.z;
                }
                cur->positions.push_back(tp);
                cur->info.push_back(ti);
            }
        }
    }
    // assimp drops meshes without faces
    std::vector<ObjObject> kept;
    for (ObjObject& o : outObjects)
        if (!o.positions.empty()) kept.push_back(std::move(o));
    outObjects.swap(kept);
    if (outObjects.empty()) {
        error = "no faces in " + filePath;
        return false;
    }
    return true;
}

}  // namespace okh
