/*
 * okrt_host.h — flat C view of the C++ host layer (raytracer_host.h), so the Python test/bench
 * harness (ctypes) and C callers can drive it. Object handles are opaque; functions returning int
 * return 1 on success and 0 on failure (the reference's `bool` convention, DX11.h:24-27);
 * okh_raytracer_last_error gives the message.
 */
#ifndef OKRT_HOST_H
#define OKRT_HOST_H
#include <stdint.h>

#include "../../include/okrt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct okh_resources okh_resources;
typedef struct okh_scene okh_scene;
typedef struct okh_raytracer okh_raytracer;

/* 68-byte Material, reference layout (GPU-Structs.hlsli:37-71 / Components.h:42-56) */
typedef struct okh_material {
    float albedo[3]; uint32_t albedoTextureId;
    float roughness; uint32_t roughnessTextureId;
    float metallic; uint32_t metallicTextureId;
    float specular; uint32_t specularTextureId;
    uint32_t normalMapIdx;
    float emissionColour[3]; float emissionPower;
    float transparency; float indexOfRefraction;
} okh_material;
void okh_material_default(okh_material* m); /* Components.h defaults */

okh_resources* okh_resources_create(void);
void okh_resources_destroy(okh_resources*);
/* positions: numTris*9 floats; info: numTris*33 floats (TriangleInfo) or NULL */
uint32_t okh_resources_add_mesh(okh_resources*, const float* positions, const float* info, uint32_t numTris);
uint32_t okh_resources_add_texture(okh_resources*, const uint8_t* rgba8, uint32_t w, uint32_t h);

okh_scene* okh_scene_create(void);
void okh_scene_destroy(okh_scene*);
uint32_t okh_scene_create_entity(okh_scene*);
void okh_scene_set_transform(okh_scene*, uint32_t e, const float pos[3], const float rotDeg[3], const float scale[3]);
void okh_scene_add_mesh(okh_scene*, uint32_t e, uint32_t meshID, const okh_material*);
void okh_scene_add_sphere(okh_scene*, uint32_t e, float radius, const okh_material*);
void okh_scene_add_dir_light(okh_scene*, uint32_t e, const float colour[3], float intensity, float effectiveAngleDeg);
void okh_scene_add_point_light(okh_scene*, uint32_t e, const float colour[3], float intensity, float radius);
void okh_scene_add_spot_light(okh_scene*, uint32_t e, const float colour[3], float intensity, float radius, float maxAngleDeg);
void okh_scene_add_camera(okh_scene*, uint32_t e, float fovDeg, float nearZ, float farZ);

okh_raytracer* okh_raytracer_create(const okrt_api* api, int device);
void okh_raytracer_destroy(okh_raytracer*);
const char* okh_raytracer_last_error(const okh_raytracer*);
int okh_raytracer_initiate(okh_raytracer*, uint32_t w, uint32_t h, const okh_resources*, const uint8_t* envCrossRgba8, uint32_t envW, uint32_t envH);
int okh_raytracer_loadMeshAndBvhData(okh_raytracer*, uint32_t maxDepth, uint32_t maxLeafTriangles);
int okh_raytracer_createOctTree(okh_raytracer*, const okh_scene*, uint32_t maxDepth, uint32_t maxLeafObjects);
void okh_raytracer_setScene(okh_raytracer*, const okh_scene*);
int okh_raytracer_render(okh_raytracer*);
int okh_raytracer_renderFrames(okh_raytracer*, uint32_t frames);
void okh_raytracer_toggleAccumulation(okh_raytracer*, int enable);
void okh_raytracer_resetAccumulation(okh_raytracer*);
uint32_t okh_raytracer_getNumAccumulationFrames(const okh_raytracer*);
void okh_raytracer_setDOF(okh_raytracer*, float strength, float distance);
int okh_raytracer_onResize(okh_raytracer*, uint32_t w, uint32_t h);
void okh_raytracer_setDebugMode(okh_raytracer*, uint32_t mode, uint32_t maxCount);
/* harness access (not in the reference) */
okrt_ctx* okh_raytracer_device_ctx(okh_raytracer*);
okrt_render_data* okh_raytracer_render_data(okh_raytracer*);
void okh_raytracer_calculateProjectionData(okh_raytracer*);
int okh_raytracer_updateBuffers(okh_raytracer*);
/* which: the okrt_table slot numbers; 100 = MeshDesc (16 B), 3 = scaled texture array (elem = side) */
int okh_raytracer_table(const okh_raytracer*, uint32_t which, const void** data, uint32_t* count, uint32_t* elemBytes);

/* ResourceManager::importAssets (ResourceManager.cpp:44-81) for Wavefront OBJ/MTL: one mesh per object, appended to
 * `res`. Writes a JSON array describing the objects into `json` (name, meshId, numTris and the five texture file
 * names as the .mtl gives them; the caller decodes images and calls okh_resources_add_texture, which keeps the C++
 * layer free of an image decoder). Returns the number of objects, or -1 (message in `json`). */
int okh_resources_import_obj(okh_resources*, const char* filePath, float scale, char* json, uint32_t jsonCapacity);

/* standalone pieces for tests */
/* returns node count; nodes36/triOrder may be NULL to size */
uint32_t okh_bvh_build(const float* tris9, uint32_t numTris, uint32_t maxLeafTriangles, uint32_t maxDepth,
                       void* nodes36, uint32_t nodeCapacity, uint32_t* triOrder);
void okh_instance_matrices(const float pos[3], const float rotDeg[3], const float scale[3], float outM[16], float outInv[16]);
/* "Save screenshot" (Application.cpp:661-674): 8-bit RGBA PNG, row 0 = top; 1 on success */
int okh_write_png(const char* path, uint32_t width, uint32_t height, const uint8_t* rgba8);
/* returns face side (0 = rejected); faces must hold 6*face*face*4 bytes */
uint32_t okh_cut_environment_cross(const uint8_t* img, uint32_t w, uint32_t h, uint8_t* faces, uint32_t facesCapacity);

#ifdef __cplusplus
}
#endif
#endif
