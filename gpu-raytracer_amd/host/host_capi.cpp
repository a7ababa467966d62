// host_capi.cpp — flat C view of the host layer (okrt_host.h).
#include <cstdio>
#include <cstring>

#include <string>

#include "obj_loader.h"
#include "okrt_host.h"
#include "png_writer.h"
#include "raytracer_host.h"

struct okh_resources { okh::ResourceManager rm; };
struct okh_scene { okh::Scene scene; };
struct okh_raytracer {
    okh::RayTracer rt;
    okh_raytracer(const okrt_api* api, int device) : rt(api, device) {}
};

static_assert(sizeof(okh_material) == sizeof(okh::Material), "material layout");
static okh::Material toMat(const okh_material* m) {
    okh::Material r;
    if (m) std::memcpy(&r, m, sizeof(r));
    return r;
}
static okh::Vec3 v3(const float* p) { return okh::Vec3(p[0], p[1], p[2]); }

extern "C" {

void okh_material_default(okh_material* m) {
    okh::Material d;
    std::memcpy(m, &d, sizeof(d));
}

okh_resources* okh_resources_create(void) { return new okh_resources(); }
void okh_resources_destroy(okh_resources* r) { delete r; }
uint32_t okh_resources_add_mesh(okh_resources* r, const float* positions, const float* info, uint32_t numTris) {
    return r->rm.addMesh(positions, info, numTris);
}
uint32_t okh_resources_add_texture(okh_resources* r, const uint8_t* rgba8, uint32_t w, uint32_t h) { return r->rm.addTexture(rgba8, w, h); }

int okh_resources_import_obj(okh_resources* r, const char* filePath, float scale, char* json, uint32_t cap) {
    std::vector<okh::ObjObject> objs;
    std::string err;
    auto put = [&](const std::string& s) {
        if (json && cap) {
            std::strncpy(json, s.c_str(), cap - 1);
            json[cap - 1] = 0;
        }
    };
    if (!okh::loadObjects(filePath, objs, scale, err)) {
        put(err);
        return -1;
    }
    auto esc = [](const std::string& s) {  // always valid ASCII JSON: names come from files and need not be UTF-8
        std::string o;
        for (char ch : s) {
            const unsigned char c = (unsigned char)ch;
            if (c == '"' || c == '\\') {
                o += '\\';
                o += ch;
            } else if (c < 0x20 || c >= 0x7F) {
                char buf[8];
                std::snprintf(buf, sizeof buf, "\\u%04x", (unsigned)c);  // (the byte value as a code point: a Latin-1 reading)
                o += buf;
            } else {
                o += ch;
            }
        }
        return o;
    };
    std::string out = "[";
    for (size_t i = 0; i < objs.size(); ++i) {
        const okh::ObjObject& o = objs[i];
        uint32_t id = r->rm.addMesh((const float*)o.positions.data(), (const float*)o.info.data(), (uint32_t)o.positions.size());
        out += std::string(i ? "," : "") + "{\"name\":\"" + esc(o.name) + "\",\"meshId\":" + std::to_string(id) + ",\"numTris\":" +
               std::to_string(o.positions.size()) + ",\"albedo\":\"" + esc(o.albedoTexturePath) + "\",\"roughness\":\"" + esc(o.roughnessTexturePath) +
               "\",\"metallic\":\"" + esc(o.metallicTexturePath) + "\",\"specular\":\"" + esc(o.specularTexturePath) + "\",\"normal\":\"" +
               esc(o.normalTexturePath) + "\"}";
    }
    out += "]";
    if (out.size() + 1 > cap) {
        put("json buffer too small");
        return -1;
    }
    put(out);
    return (int)objs.size();
}

okh_scene* okh_scene_create(void) { return new okh_scene(); }
void okh_scene_destroy(okh_scene* s) { delete s; }
uint32_t okh_scene_create_entity(okh_scene* s) { return s->scene.createEntity(); }
void okh_scene_set_transform(okh_scene* s, uint32_t e, const float pos[3], const float rot[3], const float scale[3]) {
    okh::Transform& t = s->scene.get(e).transform;
    if (pos) t.position = v3(pos);
    if (rot) t.rotation = v3(rot);
    if (scale) t.scale = v3(scale);
}
void okh_scene_add_mesh(okh_scene* s, uint32_t e, uint32_t meshID, const okh_material* m) {
    okh::Entity& en = s->scene.get(e);
    en.hasMesh = true;
    en.mesh.meshID = meshID;
    en.mesh.material = toMat(m);
}
void okh_scene_add_sphere(okh_scene* s, uint32_t e, float radius, const okh_material* m) {
    okh::Entity& en = s->scene.get(e);
    en.hasSphere = true;
    en.sphere.radius = radius;
    en.sphere.material = toMat(m);
}
void okh_scene_add_dir_light(okh_scene* s, uint32_t e, const float c[3], float intensity, float angle) {
    okh::Entity& en = s->scene.get(e);
    en.hasDirLight = true;
    std::memcpy(en.dirLight.colour, c, 12);
    en.dirLight.intensity = intensity;
    en.dirLight.effectiveAngle = angle;
}
void okh_scene_add_point_light(okh_scene* s, uint32_t e, const float c[3], float intensity, float radius) {
    okh::Entity& en = s->scene.get(e);
    en.hasPointLight = true;
    std::memcpy(en.pointLight.colour, c, 12);
    en.pointLight.intensity = intensity;
    en.pointLight.radius = radius;
}
void okh_scene_add_spot_light(okh_scene* s, uint32_t e, const float c[3], float intensity, float radius, float maxAngle) {
    okh::Entity& en = s->scene.get(e);
    en.hasSpotLight = true;
    std::memcpy(en.spotLight.colour, c, 12);
    en.spotLight.intensity = intensity;
    en.spotLight.radius = radius;
    en.spotLight.maxAngle = maxAngle;
}
void okh_scene_add_camera(okh_scene* s, uint32_t e, float fov, float nearZ, float farZ) {
    okh::Entity& en = s->scene.get(e);
    en.hasCamera = true;
    en.camera.fov = fov;
    en.camera.nearZ = nearZ;
    en.camera.farZ = farZ;
}

okh_raytracer* okh_raytracer_create(const okrt_api* api, int device) { return new okh_raytracer(api, device); }
void okh_raytracer_destroy(okh_raytracer* r) { delete r; }
const char* okh_raytracer_last_error(const okh_raytracer* r) { return r->rt.lastError().c_str(); }
int okh_raytracer_initiate(okh_raytracer* r, uint32_t w, uint32_t h, const okh_resources* res, const uint8_t* env, uint32_t ew, uint32_t eh) {
    return r->rt.initiate(w, h, res->rm, env, ew, eh) ? 1 : 0;
}
int okh_raytracer_loadMeshAndBvhData(okh_raytracer* r, uint32_t maxDepth, uint32_t maxLeaf) { return r->rt.loadMeshAndBvhData(maxDepth, maxLeaf) ? 1 : 0; }
int okh_raytracer_createOctTree(okh_raytracer* r, const okh_scene* s, uint32_t maxDepth, uint32_t maxLeafObjects) {
    return r->rt.createOctTree(s->scene, maxDepth, maxLeafObjects) ? 1 : 0;
}
void okh_raytracer_setScene(okh_raytracer* r, const okh_scene* s) { r->rt.setScene(s->scene); }
int okh_raytracer_render(okh_raytracer* r) { return r->rt.render() ? 1 : 0; }
int okh_raytracer_renderFrames(okh_raytracer* r, uint32_t frames) { return r->rt.renderFrames(frames) ? 1 : 0; }
void okh_raytracer_toggleAccumulation(okh_raytracer* r, int enable) { r->rt.toggleAccumulation(enable != 0); }
void okh_raytracer_resetAccumulation(okh_raytracer* r) { r->rt.resetAccumulation(); }
uint32_t okh_raytracer_getNumAccumulationFrames(const okh_raytracer* r) { return r->rt.getNumAccumulationFrames(); }
void okh_raytracer_setDOF(okh_raytracer* r, float strength, float distance) {
    r->rt.getDOFStrength() = strength;
    r->rt.getDOFDistance() = distance;
}
int okh_raytracer_onResize(okh_raytracer* r, uint32_t w, uint32_t h) { return r->rt.onResize(w, h) ? 1 : 0; }
void okh_raytracer_setDebugMode(okh_raytracer* r, uint32_t mode, uint32_t maxCount) {
    r->rt.setDebugMode((okh::RayTracer::DebugDisplayMode)mode);
    r->rt.getDebugMaxCount() = maxCount;
}
okrt_ctx* okh_raytracer_device_ctx(okh_raytracer* r) { return r->rt.deviceContext(); }
okrt_render_data* okh_raytracer_render_data(okh_raytracer* r) { return &r->rt.renderDataMutable(); }
void okh_raytracer_calculateProjectionData(okh_raytracer* r) { r->rt.calculateProjectionData(); }
int okh_raytracer_updateBuffers(okh_raytracer* r) { return r->rt.updateBuffers() ? 1 : 0; }

int okh_raytracer_table(const okh_raytracer* r, uint32_t which, const void** data, uint32_t* count, uint32_t* elemBytes) {
    const okh::RayTracer& rt = r->rt;
    switch (which) {
        case OKRT_TABLE_TRI_POS: *data = rt.getTrianglesPos().data(); *count = (uint32_t)rt.getTrianglesPos().size(); *elemBytes = 36; return 1;
        case OKRT_TABLE_TRI_INFO: *data = rt.getTrianglesInfo().data(); *count = (uint32_t)rt.getTrianglesInfo().size(); *elemBytes = 132; return 1;
        case OKRT_TABLE_BVH: *data = rt.getBvhTreeNodes().data(); *count = (uint32_t)rt.getBvhTreeNodes().size(); *elemBytes = 36; return 1;
        case OKRT_TABLE_OCT: *data = rt.getOctTreeNodes().data(); *count = (uint32_t)rt.getOctTreeNodes().size(); *elemBytes = 48; return 1;
        case OKRT_TABLE_MESH: *data = rt.getGpuMeshes().data(); *count = (uint32_t)rt.getGpuMeshes().size(); *elemBytes = 232; return 1;
        case 100: *data = rt.getMeshDescriptors().data(); *count = (uint32_t)rt.getMeshDescriptors().size(); *elemBytes = 16; return 1;
        case 3: *data = rt.scaledTextures().data(); *count = (uint32_t)rt.scaledTextures().size(); *elemBytes = rt.textureArraySize(); return 1;
        default: return 0;
    }
}

uint32_t okh_bvh_build(const float* tris9, uint32_t numTris, uint32_t maxLeaf, uint32_t maxDepth, void* nodes36, uint32_t cap, uint32_t* triOrder) {
    okh::AABB box;
    const okh::Triangle* tris = (const okh::Triangle*)tris9;
    for (uint32_t t = 0; t < numTris; ++t)
        for (int k = 0; k < 3; ++k) {
            box.min = okh::vmin(tris[t].position[k], box.min);
            box.max = okh::vmax(tris[t].position[k], box.max);
        }
    okh::BvhBuilder b(maxLeaf, maxDepth);
    b.buildTree(tris, numTris, box);
    const std::vector<okh::BvhNode>& nodes = b.getTree();
    uint32_t n = (uint32_t)nodes.size();
    if (!nodes36 || cap < n) return n;
    okh::GPUNode* out = (okh::GPUNode*)nodes36;
    uint32_t local = 0;
    for (uint32_t k = 0; k < n; ++k) {
        okh::GPUNode g;
        std::memcpy(g.bbMin, &nodes[k].boundingBox.min, 12);
        std::memcpy(g.bbMax, &nodes[k].boundingBox.max, 12);
        g.firstChildIdx = nodes[k].firstChildIdx;
        if (nodes[k].isLeaf()) {
            g.triStart = local;
            g.triEnd = local + (uint32_t)nodes[k].triIndicies.size();
            for (uint32_t t : nodes[k].triIndicies) triOrder[local++] = t;
        }
        out[k] = g;
    }
    return n;
}

void okh_instance_matrices(const float pos[3], const float rot[3], const float scale[3], float outM[16], float outInv[16]) {
    okh::Transform t;
    t.position = v3(pos);
    t.rotation = v3(rot);
    t.scale = v3(scale);
    okh::Mat4 m = okh::transpose(t.calculateMatrix());
    okh::Mat4 inv = okh::inverse(m);
    std::memcpy(outM, m.c, 64);
    std::memcpy(outInv, inv.c, 64);
}

int okh_write_png(const char* path, uint32_t width, uint32_t height, const uint8_t* rgba8) {
    return path && okh::writePng(path, width, height, rgba8) ? 1 : 0;
}

uint32_t okh_cut_environment_cross(const uint8_t* img, uint32_t w, uint32_t h, uint8_t* faces, uint32_t cap) {
    std::vector<uint8_t> f;
    uint32_t face = 0;
    if (!okh::RayTracer::cutEnvironmentCross(img, w, h, f, face)) return 0;
    if (faces && cap >= f.size()) std::memcpy(faces, f.data(), f.size());
    return face;
}
}
