// okrt_device.h — device-side building blocks of the gfx950 path tracer: table layouts, the
// float lowering conventions, RNG, intersection tests, texture/environment sampling and the
// BSDF. Everything here is __device__ __forceinline__ and shared by all traversal kernels.
//
// What it computes is the reference's compute shader (resources/shaders/RaytracerCS.hlsl with
// GPU-Utilities.hlsli / GPU-Structs.hlsli; cited per function). How it computes it is fixed by
// these conventions so that results are reproducible to the bit outside libm transcendentals:
//   dot = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x)); cross.x = fma(a.y,b.z, -(a.z*b.y));
//   lerp = fma(t, b-a, a); reflect = fma(n, -(2 dot(i,n)), i); normalize = v * (1/sqrt(dot));
//   matrix*vector = fma chain along a memory row; x*y+z = fma where the source has that shape;
//   pow(x,y) = exp2(y*log2(x)); min/max = fminf/fmaxf; IEEE divide and sqrt.
// Compile with -ffp-contract=off so that only the fmas written here are fused.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/okrt.h"

namespace okrt {

#define OKRT_DEV __device__ __forceinline__

constexpr uint32_t kInvalid = 0xFFFFFFFFu;  // UINT_MAX  (GPU-Structs.hlsli:1)
constexpr float kFltMax = 3.402823466e+38F;  // FLT_MAX   (GPU-Structs.hlsli:2)
constexpr float kPi = 3.14159265f;           // PI        (GPU-Structs.hlsli:3)

// ---- table layouts, byte for byte (GPU-Structs.hlsli) --------------------------------------------
struct Material {  // :37-71
    float albedo[3]; uint32_t albedoTex;
    float roughness; uint32_t roughnessTex;
    float metallic; uint32_t metallicTex;
    float specular; uint32_t specularTex;
    uint32_t normalMapIdx;
    float emission[3]; float emissionPower;
    float transparency; float ior;
};
struct SphereRec { float position[3]; Material material; float radius; };            // :86-93
struct TriangleRec { float p[9]; };                                                  // :95-98
struct VertexInfo { float normal[3]; float uv[2]; float tangent[3]; float bitangent[3]; };  // :100-106
struct TriangleInfoRec { VertexInfo v[3]; };                                         // :108-111
struct MeshRec {                                                                     // :113-125
    float M[16]; float Minv[16];
    uint32_t triStart, triEnd;
    float bbMin[3], bbMax[3];
    Material material;
    uint32_t bvhNodeStartIdx;
};
struct DirLightRec { float colour[3]; float intensity; float effectiveAngle; float direction[3]; };      // :127-135
struct PointLightRec { float colour[3]; float intensity; float radius; float position[3]; };             // :137-145
struct SpotLightRec { float colour[3]; float intensity; float radius; float maxAngle; float position[3]; float direction[3]; };  // :147-157
struct BvhNodeRec { float bbMin[3], bbMax[3]; uint32_t triStart, triEnd, firstChild; };  // :159-165
struct OctNodeRec { float bbMin[3], bbMax[3]; uint32_t meshesStart, meshesEnd, spheresStart, spheresEnd, firstChild, numChildren; };  // :167-179
static_assert(sizeof(Material) == 68 && sizeof(SphereRec) == 84 && sizeof(TriangleRec) == 36, "abi");
static_assert(sizeof(TriangleInfoRec) == 132 && sizeof(MeshRec) == 232 && sizeof(BvhNodeRec) == 36, "abi");
static_assert(sizeof(OctNodeRec) == 48 && sizeof(DirLightRec) == 32 && sizeof(PointLightRec) == 32 && sizeof(SpotLightRec) == 48, "abi");

// Everything a kernel reads, passed by value as a kernel argument.
struct SceneView {
    const TriangleRec* triPos;          // t0
    const TriangleInfoRec* triInfo;     // t1
    const float4* shadeInfo;            // derived from t1 at upload: 64 B per triangle = {n0,uv0.x} {n1,uv0.y} {n2,uv1.x} {uv1.y,uv2.x,uv2.y,-}:
                                        // the normals + uvs every mesh hit needs in ONE aligned 64-byte read instead of a 132-byte
                                        // unaligned record (tangent frames stay in t1: only normal-mapped hits read them)
    const BvhNodeRec* bvh;              // t2
    const uint8_t* textures;            // t3  count x texH x texW RGBA8
    const uint8_t* env;                 // t4  6 x face x face RGBA8
    const OctNodeRec* oct;              // t5
    const SphereRec* spheres;           // t6
    const MeshRec* meshes;              // t7
    const DirLightRec* dirLights;       // t8
    const PointLightRec* pointLights;   // t9
    const SpotLightRec* spotLights;     // t10
    uint32_t numOct, numTextures, texW, texH, envFace;
    uint32_t maxBounces;
    // 1 = a HIT on the last path segment adds nothing (okrt_context.hip: lastSegmentDark): no light of any kind in this frame's
    // RenderData, every material's emission colour * power is a zero vector and every albedo lies in [-1, 1]. Then the shader's
    // `light += emission * contribution * (1 - transparencyFactor)` (RaytracerCS.hlsl:509-510) is light + (+-0): the path ends
    // with the radiance it carried, which is +0 (emission and lights are the only sources before a miss, :448-493).
    uint32_t lastSegmentDark;
    uint32_t lastSegmentAnyHit;  // OKRT_OPT_LAST_SEGMENT_ANY_HIT (only with lastSegmentDark): the last segment's ray ends at its first accepted hit
};

// Per-thread counters; summed into okrt_counters when OKRT_OPT_COUNTERS is on.
struct PathCounters {
    uint32_t segments, nOct, nMesh, nNode, nTri, meshHits, sphereHits, texTaps, envTaps, fastNodes, fastTris;
};

// ---- float3 with the lowering conventions -------------------------------------------------------
struct V3 { float x, y, z; };
OKRT_DEV V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
OKRT_DEV V3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }
OKRT_DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
OKRT_DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
OKRT_DEV V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
OKRT_DEV V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
OKRT_DEV V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
OKRT_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
OKRT_DEV V3 fma3(V3 a, float s, V3 c) { return mk(fma_(a.x, s, c.x), fma_(a.y, s, c.y), fma_(a.z, s, c.z)); }
OKRT_DEV V3 fma3(V3 a, V3 b, V3 c) { return mk(fma_(a.x, b.x, c.x), fma_(a.y, b.y, c.y), fma_(a.z, b.z, c.z)); }
OKRT_DEV float dot(V3 a, V3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
OKRT_DEV V3 cross(V3 a, V3 b) {
    return mk(fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x)));
}
// IEEE 1/x and sqrt(x) — the same bits as the compiler's `1.f / x` and `sqrtf(x)` for EVERY float (tests/test_gpu_kats.py
// runs all 2^32 patterns through both) — without the general-purpose sequences (10 and 16 instructions: operand scaling for
// denormal inputs / results, v_div_fixup, class checks). In the middle of the exponent range, where every operand of a
// traversal lives, the hardware estimate is within one ulp and needs only the last step of those sequences: one Newton
// step in fma arithmetic for the reciprocal, the residual test against the two neighbours for the root. Anything else
// takes the compiler's sequence.
#ifndef OKRT_EXACT_SHORT
#define OKRT_EXACT_SHORT 1
#endif
OKRT_DEV float ieeeRcp(float x) {
#if OKRT_EXACT_SHORT
    if (((__float_as_uint(x) >> 23) & 255u) - 27u > 200u) return 1.0f / x;  // outside 2^-100 <= |x| < 2^101 (zero, denormal, huge, inf, nan)
    const float r = __builtin_amdgcn_rcpf(x);
    return fma_(r, fma_(-x, r, 1.0f), r);
#else
    return 1.0f / x;
#endif
}
OKRT_DEV float ieeeSqrt(float x) {
#if OKRT_EXACT_SHORT
    if (!(x >= 0x1p-100f && x < 0x1p101f)) return sqrtf(x);
    const float s = __builtin_amdgcn_sqrtf(x);
    const float below = __uint_as_float(__float_as_uint(s) - 1u), above = __uint_as_float(__float_as_uint(s) + 1u);
    const float resBelow = fma_(-below, s, x), resAbove = fma_(-above, s, x);
    const float t = resBelow <= 0.f ? below : s;
    return resAbove > 0.f ? above : t;
#else
    return sqrtf(x);
#endif
}
OKRT_DEV float rsq(float x) { return ieeeRcp(ieeeSqrt(x)); }
OKRT_DEV V3 normalize(V3 v) { return v * rsq(dot(v, v)); }
OKRT_DEV float length(V3 v) { return ieeeSqrt(dot(v, v)); }
OKRT_DEV float lerp(float a, float b, float t) { return fma_(t, b - a, a); }
OKRT_DEV V3 lerp(V3 a, V3 b, float t) { return mk(lerp(a.x, b.x, t), lerp(a.y, b.y, t), lerp(a.z, b.z, t)); }
OKRT_DEV V3 reflect(V3 i, V3 n) { float d = dot(i, n); return fma3(n, -(d + d), i); }
OKRT_DEV float hlslPow(float x, float y) { return exp2f(y * log2f(x)); }
OKRT_DEV float saturate(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
OKRT_DEV float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
OKRT_DEV bool finite_(float x) { return fabsf(x) <= kFltMax; }

// mul(float4(v,1|0), M) for a matrix uploaded as transpose(G) in glm memory order (== G*v):
// component j runs along memory row j.
OKRT_DEV V3 matPoint(const float* m, V3 v) {
    float r0 = fma_(m[2], v.z, fma_(m[1], v.y, m[0] * v.x)) + m[3];
    float r1 = fma_(m[6], v.z, fma_(m[5], v.y, m[4] * v.x)) + m[7];
    float r2 = fma_(m[10], v.z, fma_(m[9], v.y, m[8] * v.x)) + m[11];
    return mk(r0, r1, r2);
}
OKRT_DEV V3 matDir(const float* m, V3 v) {
    float r0 = fma_(m[2], v.z, fma_(m[1], v.y, m[0] * v.x));
    float r1 = fma_(m[6], v.z, fma_(m[5], v.y, m[4] * v.x));
    float r2 = fma_(m[10], v.z, fma_(m[9], v.y, m[8] * v.x));
    return mk(r0, r1, r2);
}

struct Ray { V3 origin, direction; };

// ---- RNG (GPU-Utilities.hlsli:18-25, 51-78) ------------------------------------------------------
OKRT_DEV uint32_t pcgHash(uint32_t& seed) {
    seed *= 747796405u + 2891336453u;  // one multiply by 3639132858 (quirk Q2)
    seed = ((seed >> ((seed >> 28u) + 4u)) ^ seed) * 277803737u;
    seed = (seed >> 22u) ^ seed;
    return seed;
}
OKRT_DEV float randomFloat(uint32_t& seed) { return (float)pcgHash(seed) / 4294967296.0f; }
OKRT_DEV float randomNormal(uint32_t& seed) {
    float theta = (2 * kPi) * randomFloat(seed);
    float rho = sqrtf(-2 * logf(randomFloat(seed)));
    return rho * cosf(theta);
}
OKRT_DEV V3 randomVector(uint32_t& seed) {
    float x = randomNormal(seed);
    float y = randomNormal(seed);
    float z = randomNormal(seed);
    return normalize(mk(x, y, z));
}
OKRT_DEV float clampedDot(V3 v, V3 u) { return fmaxf(dot(v, u), 0.f); }

// ---- Collision (GPU-Utilities.hlsli:95-198) ---------------------------------------------------------
OKRT_DEV float rayAndSphere(const Ray& ray, V3 pos, float radius) {  // :98-114
    V3 rayToSphere = pos - ray.origin;
    float distToClosestPoint = dot(rayToSphere, ray.direction);
    float magSqrd = dot(rayToSphere, rayToSphere);
    float radSqrd = radius * radius;
    if (distToClosestPoint < 0.f && magSqrd > radSqrd) return -1.f;
    float sideA = fma_(-distToClosestPoint, distToClosestPoint, magSqrd);
    if (sideA > radSqrd) return -1.f;
    float sideB = sqrtf(radSqrd - sideA);
    return magSqrd > radSqrd ? distToClosestPoint - sideB : distToClosestPoint + sideB;
}
// Moeller-Trumbore with the reference's edge convention E1 = p0-p2, E2 = p1-p2 (:117-155).
// Takes p2 and the two edges (what the re-laid-out triangle table stores; the subtraction is the
// same one the shader performs, so precomputing it changes nothing).
OKRT_DEV float rayAndTriangleEdges(V3 ro, V3 rd, V3 p2, V3 E1, V3 E2, float& u, float& v) {
    const float EPSILON = 0.000001f;
    V3 cross1 = cross(rd, E2);
    float determinant = dot(E1, cross1);
    if (determinant < EPSILON && determinant > -EPSILON) return -1.f;
    float inverseDet = ieeeRcp(determinant);
    V3 rayMoved = ro - p2;
    u = dot(rayMoved, cross1) * inverseDet;
    if (u < -EPSILON) return -1.f;
    V3 cross2 = cross(rayMoved, E1);
    v = dot(rd, cross2) * inverseDet;
    if (v < -EPSILON || u + v > 1.0f) return -1.f;
    float t1 = dot(E2, cross2) * inverseDet;
    if (t1 < -EPSILON) return -1.f;
    return t1;
}
// Slab test with a precomputed 1/direction (:183-197).
OKRT_DEV float rayAndAABBDist(V3 ro, V3 inv, V3 bbMin, V3 bbMax) {
    V3 tMin = (bbMin - ro) * inv;
    V3 tMax = (bbMax - ro) * inv;
    float nx = fminf(tMin.x, tMax.x), ny = fminf(tMin.y, tMax.y), nz = fminf(tMin.z, tMax.z);
    float fx = fmaxf(tMin.x, tMax.x), fy = fmaxf(tMin.y, tMax.y), fz = fmaxf(tMin.z, tMax.z);
    float distFar = fminf(fminf(fx, fy), fz);
    float distNear = fmaxf(fmaxf(nx, ny), nz);
    bool didHit = distFar >= distNear && distFar > 0.f;
    return didHit ? distNear : kFltMax;
}
OKRT_DEV V3 rcp3(V3 d) { return mk(ieeeRcp(d.x), ieeeRcp(d.y), ieeeRcp(d.z)); }

// ---- sampling -------------------------------------------------------------------------------------
// D3D11 fixed-function filtering is outside the reference repository; the definition used (and
// restated by the oracle): texel centres at +0.5, exact fp32 weights, UNORM8 = c/255,
// lerp(lerp(t00,t10,fx), lerp(t01,t11,fx), fy). Texture2DArray: WRAP (RayTracer.cpp:99-109),
// slice = round-half-even(idx) clamped. TextureCube: v_cubeid face rule, clamp inside the face.
// c / 255.f for c = 0..255, correctly rounded, without the 10-instruction IEEE division sequence: one Newton step on
// q = c * fl(1/255) is exact for all 256 inputs (checked exhaustively against the division: tests/test_abi.py).
OKRT_DEV float unorm8(uint32_t c) {
    const float r = 1.0f / 255.0f, fc = (float)c;
    const float q = fc * r;
    return fma_(fma_(-q, 255.f, fc), r, q);
}
OKRT_DEV V3 texel(const uint8_t* p) {
    uint32_t w = *reinterpret_cast<const uint32_t*>(p);
    return mk(unorm8(w & 255u), unorm8((w >> 8) & 255u), unorm8((w >> 16) & 255u));
}
// a - floor(a / n) * n for the WRAP address mode; a division by a power of two is an exact multiplication
OKRT_DEV float wrapFloor(float a, int n) {
    const float fn = (float)n;
    // 1 / 2^k by exponent arithmetic (exact): bits(2^-k) = (254 << 23) - bits(2^k)
    const float q = (n & (n - 1)) == 0 ? a * __uint_as_float((254u << 23) - __float_as_uint(fn)) : a / fn;
    return a - floorf(q) * fn;
}
// (int)w % n folded into [0, n): w already lies in [0, n] after wrapFloor except for non-finite / huge coordinates
OKRT_DEV int wrapIndex(float w, int n) {
    int i = (int)w;
    if ((unsigned)i >= (unsigned)n) {
        i %= n;
        if (i < 0) i += n;
    }
    return i;
}
// Texture memory is tiled: 4x4 texels = 64 bytes = one HBM access granule, tiles row-major, edge tiles padded. The
// 2x2 footprint of a bilinear tap falls into one tile in 9 of 16 cases and into two in 6 more; in the row-major
// layout the uploads arrive in it always spans two lines (okrt_set_textures / okrt_set_envmap re-tile on upload).
OKRT_DEV uint32_t tilesAcross(int n) { return ((uint32_t)n + 3u) >> 2; }
OKRT_DEV size_t tiledOffset(int x, int y, uint32_t tilesX) {
    return ((size_t)((uint32_t)y >> 2) * tilesX + ((uint32_t)x >> 2)) * 64 + (((uint32_t)y & 3u) * 4u + ((uint32_t)x & 3u)) * 4u;
}
OKRT_DEV V3 bilerp(V3 t00, V3 t10, V3 t01, V3 t11, float fx, float fy) {
    return lerp(lerp(t00, t10, fx), lerp(t01, t11, fx), fy);
}
OKRT_DEV V3 sampleTexture(const SceneView& s, uint32_t textureIdx, float u, float v, PathCounters& cnt) {  // RaytracerCS.hlsl:133-136
    cnt.texTaps++;
    if (s.numTextures == 0) return mk(0.f, 0.f, 0.f);
    float fidx = (float)textureIdx;
    uint32_t slice = (uint32_t)fminf(rintf(fidx), (float)(s.numTextures - 1));
    if (!finite_(u)) u = 0.f;
    if (!finite_(v)) v = 0.f;
    const int W = (int)s.texW, H = (int)s.texH;
    float x = fma_(u, (float)W, -0.5f), y = fma_(v, (float)H, -0.5f);
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    float wx = wrapFloor(fx0, W), wy = wrapFloor(fy0, H);
    int x0 = wrapIndex(wx, W), y0 = wrapIndex(wy, H);
    int x1 = x0 + 1 == W ? 0 : x0 + 1;
    int y1 = y0 + 1 == H ? 0 : y0 + 1;
    const uint32_t tx = tilesAcross(W);
    const uint8_t* base = s.textures + (size_t)slice * tx * tilesAcross(H) * 64;
    V3 t00 = texel(base + tiledOffset(x0, y0, tx)), t10 = texel(base + tiledOffset(x1, y0, tx));
    V3 t01 = texel(base + tiledOffset(x0, y1, tx)), t11 = texel(base + tiledOffset(x1, y1, tx));
    return bilerp(t00, t10, t01, t11, fx, fy);
}
OKRT_DEV V3 sampleEnvironment(const SceneView& s, V3 d, PathCounters& cnt) {  // RaytracerCS.hlsl:86-88
    cnt.envTaps++;
    if (s.envFace == 0) return mk(0.f, 0.f, 0.f);
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    int face;
    float sc, tc, ma;
    if (az >= ax && az >= ay) {
        ma = az;
        if (d.z >= 0) { face = 4; sc = d.x; tc = -d.y; } else { face = 5; sc = -d.x; tc = -d.y; }
    } else if (ay >= ax) {
        ma = ay;
        if (d.y >= 0) { face = 2; sc = d.x; tc = d.z; } else { face = 3; sc = d.x; tc = -d.z; }
    } else {
        ma = ax;
        if (d.x >= 0) { face = 0; sc = -d.z; tc = -d.y; } else { face = 1; sc = d.z; tc = -d.y; }
    }
    float u = fma_(sc / ma, 0.5f, 0.5f), v = fma_(tc / ma, 0.5f, 0.5f);
    if (!finite_(u)) u = 0.f;
    if (!finite_(v)) v = 0.f;
    const int F = (int)s.envFace;
    float x = fma_(u, (float)F, -0.5f), y = fma_(v, (float)F, -0.5f);
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    int x0 = min(max((int)fx0, 0), F - 1), x1 = min(max((int)fx0 + 1, 0), F - 1);
    int y0 = min(max((int)fy0, 0), F - 1), y1 = min(max((int)fy0 + 1, 0), F - 1);
    const uint32_t tx = tilesAcross(F);
    const uint8_t* base = s.env + (size_t)face * tx * tx * 64;
    V3 t00 = texel(base + tiledOffset(x0, y0, tx)), t10 = texel(base + tiledOffset(x1, y0, tx));
    V3 t01 = texel(base + tiledOffset(x0, y1, tx)), t11 = texel(base + tiledOffset(x1, y1, tx));
    return bilerp(t00, t10, t01, t11, fx, fy);
}

// ---- hit record + shading fetch (RaytracerCS.hlsl:336-392) -------------------------------------------
struct Hit {
    float distance;    // world-space distance of the closest hit, FLT_MAX on a miss
    uint32_t hitIdx;   // sphere index or instance (mesh table) index, UINT_MAX on a miss
    uint32_t hitType;  // 0 sphere, 1 mesh
    uint32_t triIdx;   // index into t0/t1
    float bu, bv;      // barycentrics (u -> p0, v -> p1)
};
struct Surface {
    bool hit;
    float distance;
    V3 worldPosition, worldNormal;
    V3 albedo, emission;  // emission = emissionColour * emissionPower
    float roughness, metallic, specular, transparency, ior;
};

OKRT_DEV V3 bary3(float bx, float by, float bz, const float* a, const float* b, const float* c) {  // GPU-Utilities.hlsli:3-6
    V3 r = ld3(a) * bx;
    r = fma3(ld3(b), by, r);
    return fma3(ld3(c), bz, r);
}
OKRT_DEV V3 sampleNormalMap(const SceneView& s, uint32_t idx, float u, float v, V3 normal, V3 tangent, V3 bitangent, PathCounters& cnt) {  // :167-175
    V3 t = sampleTexture(s, idx, u, v, cnt);
    t = mk(fma_(t.x, 2.f, -1.f), fma_(t.y, 2.f, -1.f), fma_(t.z, 2.f, -1.f));
    V3 r = tangent * t.x;
    r = fma3(bitangent, t.y, r);
    r = fma3(normal, t.z, r);
    return normalize(r);
}
OKRT_DEV void applyMaterial(const SceneView& s, const Material& m, float u, float v, Surface& o, PathCounters& cnt) {  // :138-165
    o.albedo = ld3(m.albedo);
    o.roughness = m.roughness;
    o.metallic = m.metallic;
    o.specular = m.specular;
    o.transparency = m.transparency;
    o.ior = m.ior;
    o.emission = ld3(m.emission) * m.emissionPower;
    if (m.albedoTex != kInvalid) o.albedo = sampleTexture(s, m.albedoTex, u, v, cnt);
    if (m.roughnessTex != kInvalid) o.roughness = clampf(sampleTexture(s, m.roughnessTex, u, v, cnt).x, m.roughness, 1.f);
    if (m.metallicTex != kInvalid) o.metallic = clampf(sampleTexture(s, m.metallicTex, u, v, cnt).x, m.metallic, 1.f);
    if (m.specularTex != kInvalid) o.specular = clampf(sampleTexture(s, m.specularTex, u, v, cnt).x, m.specular, 1.f);
}
OKRT_DEV Surface shadeFetch(const SceneView& s, const Ray& ray, const Hit& h, PathCounters& cnt) {
    Surface o;
    o.hit = h.hitIdx != kInvalid;
    o.distance = h.distance;
    o.worldPosition = fma3(ray.direction, h.distance, ray.origin);  // :337
    o.worldNormal = mk(0.f, 0.f, 0.f);
    o.albedo = o.emission = mk(0.f, 0.f, 0.f);
    o.roughness = o.metallic = o.specular = o.transparency = o.ior = 0.f;
    if (!o.hit) return o;  // quirk Q4: the shader's OOB sphere read returns zeros and is unused
    if (h.hitType == 0) {  // sphere :342-362
        cnt.sphereHits++;
        const SphereRec& sp = s.spheres[h.hitIdx];
        const Material& m = sp.material;
        V3 n = normalize(o.worldPosition - ld3(sp.position));
        float su = fma_(atan2f(n.x, n.z), 1.f / (2 * kPi), 0.5f);  // normalToUV :77-84
        float sv = fma_(n.y, 0.5f, 0.5f);
        if (m.normalMapIdx != kInvalid) {
            V3 tangent = normalize(cross(mk(0.f, 1.f, 0.f), n));
            V3 bitangent = normalize(cross(tangent, n));
            o.worldNormal = sampleNormalMap(s, m.normalMapIdx, su, sv, n, tangent, bitangent, cnt);
        } else {
            o.worldNormal = n;
        }
        applyMaterial(s, m, su, sv, o, cnt);
    } else {  // mesh :364-391
        cnt.meshHits++;
        const MeshRec& mesh = s.meshes[h.hitIdx];
        const Material& m = mesh.material;
        const TriangleInfoRec& ti = s.triInfo[h.triIdx];
        const float4* si = s.shadeInfo + (size_t)h.triIdx * 4;
        const float4 r0 = si[0], r1 = si[1], r2 = si[2], r3 = si[3];
        const float n0[3] = {r0.x, r0.y, r0.z}, n1[3] = {r1.x, r1.y, r1.z}, n2[3] = {r2.x, r2.y, r2.z};
        float bz = 1.f - (h.bu + h.bv);  // :324
        float lu = fma_(r3.y, bz, fma_(r2.w, h.bv, r0.w * h.bu));
        float lv = fma_(r3.z, bz, fma_(r3.x, h.bv, r1.w * h.bu));
        V3 n = normalize(matDir(mesh.M, bary3(h.bu, h.bv, bz, n0, n1, n2)));  // M*n, not inverse-transpose (Q11)
        if (m.normalMapIdx != kInvalid) {
            V3 tangent = normalize(matDir(mesh.M, bary3(h.bu, h.bv, bz, ti.v[0].tangent, ti.v[1].tangent, ti.v[2].tangent)));
            V3 bitangent = normalize(matDir(mesh.M, bary3(h.bu, h.bv, bz, ti.v[0].bitangent, ti.v[1].bitangent, ti.v[2].bitangent)));
            o.worldNormal = sampleNormalMap(s, m.normalMapIdx, lu, lv, n, tangent, bitangent, cnt);
        } else {
            o.worldNormal = n;
        }
        applyMaterial(s, m, lu, lv, o, cnt);
    }
    return o;
}

// ---- BSDF (RaytracerCS.hlsl:100-131, GPU-Utilities.hlsli:85-93) -----------------------------------------
OKRT_DEV float reflectance(float cosine, float reflectionIdx) {  // quirk Q5: (1 + r0)
    float r0 = (1.f - reflectionIdx) / (1.f + reflectionIdx);
    r0 *= r0;
    return fma_(1.f + r0, hlslPow(1.f - cosine, 5.f), r0);
}
OKRT_DEV V3 refractUser(V3 direction, V3 normal, float ratio) {  // quirk Q8
    float cosTheta = fminf(dot(-direction, normal), 1.f);
    V3 perp = fma3(normal, cosTheta, direction) * ratio;
    float k = -sqrtf(fabsf(1.f - dot(perp, perp)));
    return fma3(normal, k, perp);
}
OKRT_DEV V3 findReflectDirection(V3 direction, V3 normal, float roughness, uint32_t& seed) {
    V3 diffuse = normalize(normal + randomVector(seed));
    V3 specular = reflect(direction, normal);
    return normalize(lerp(specular, diffuse, roughness));
}
OKRT_DEV V3 findTransparencyBounce(V3 direction, V3 normal, float ior, uint32_t& seed) {
    bool front = dot(direction, normal) < 0.f;
    float ratio = front ? 1.f / ior : ior;
    if (!front) normal = normal * -1.f;
    float cosTheta = dot(-direction, normal);
    float sinTheta = sqrtf(fma_(-cosTheta, cosTheta, 1.f));
    bool cannotRefract = ratio * sinTheta > 1.f;
    float refl = reflectance(cosTheta, ratio);
    float rnd = randomFloat(seed);  // always consumed: SM5 `||` does not short-circuit (Q6)
    if (cannotRefract | (refl > rnd)) direction = reflect(direction, normal);
    return refractUser(direction, normal, ratio);  // also after reflecting (Q7)
}

// ---- lights (RaytracerCS.hlsl:397-418, 448-487) ----------------------------------------------------------
OKRT_DEV V3 getLighting(const Ray& ray, const Surface& sf, V3 lightPos, float radius, V3 colour, float intensity) {
    float dist = rayAndSphere(ray, lightPos, radius);
    if (dist < 0.f) return mk(0.f, 0.f, 0.f);
    float mod = 1.f;
    if (sf.hit && sf.distance > dist)
        mod *= clampedDot(ray.direction, -sf.worldNormal);
    else if (sf.hit && sf.distance < dist)
        mod *= sf.transparency * clampedDot(ray.direction, -sf.worldNormal);
    return (colour * intensity) * mod;
}
OKRT_DEV void addLights(const SceneView& s, const okrt_render_data& rd, const Ray& ray, const Surface& sf, uint32_t bounce,
                        V3 contribution, V3& light, uint32_t& seed) {
    for (uint32_t p = 0; p < rd.numPointLights; p++) {
        const PointLightRec& L = s.pointLights[p];
        light = fma3(getLighting(ray, sf, ld3(L.position), L.radius, ld3(L.colour), L.intensity), contribution, light);
    }
    for (uint32_t k = 0; k < rd.numSpotLights; k++) {
        const SpotLightRec& L = s.spotLights[k];
        V3 rayToLight = normalize(fma3(randomVector(seed), 0.01f, ld3(L.position)) - ray.origin);
        float cosTheta = dot(rayToLight, -ld3(L.direction));
        if (cosTheta < L.maxAngle) continue;
        light = fma3(getLighting(ray, sf, ld3(L.position), L.radius, ld3(L.colour), L.intensity), contribution, light);
    }
    for (uint32_t d = 0; d < rd.numDirLights; d++) {
        const DirLightRec& L = s.dirLights[d];
        if (clampedDot(ray.direction, -ld3(L.direction)) < L.effectiveAngle) continue;
        float mod = 1.f;
        if (sf.hit && bounce)
            mod *= sf.transparency * clampedDot(ray.direction, -sf.worldNormal);
        else if (!bounce)
            mod = 0.f;  // Q13
        light = fma3((ld3(L.colour) * L.intensity) * contribution, mod, light);
    }
}

// ---- camera (RaytracerCS.hlsl:177-216, 430) ------------------------------------------------------------------
// `frameN` = numAccumulationFrames of the frame this sample belongs to (rd's own value for a single frame).
OKRT_DEV Ray primaryRay(const okrt_render_data& rd, uint32_t frameN, uint32_t px, uint32_t py, uint32_t& seed) {
    seed = px + (py + 74813u) * rd.textureDims[0] * (frameN + 1u);  // :430
    float posx = (float)px, posy = (float)(rd.textureDims[1] - py);                   // H - y (Q10)
    posx += randomFloat(seed) - 0.5f;
    posy += randomFloat(seed) - 0.5f;
    posx /= (float)rd.textureDims[0];
    posy /= (float)rd.textureDims[1];
    posx = fma_(posx, 2.f, -1.f);
    posy = fma_(posy, 2.f, -1.f);
    const float* P = rd.cameraInverseProjectionMatrix;
    V3 pos = mk(posx, posy, rd.cameraNearZ);
    V3 target = matPoint(P, pos);
    V3 tdir = normalize(mk(target.x / target.z, target.y / target.z, target.z / target.z));
    Ray ray;
    ray.origin = ld3(rd.cameraPosition);
    ray.direction = matDir(rd.cameraInverseViewMatrix, tdir);
    // applyDOF :208-216 (always runs: 2 draws). With strength 0 the lens offset is (+-0, +-0, +-0) — cos, sin and sqrt are finite, the
    // basis vectors finite — and origin + (+-0) is the origin itself for every component but a NEGATIVE zero (-0 + +0 = +0): then the
    // two draws are consumed and the trigonometry, whose only use is the sign of those zeros, is not evaluated (same bits; 2-7 % of
    // the camera kernel).
    V3 focusPoint = fma3(ray.direction, rd.dofDistance + rd.cameraNearZ, ray.origin);
    const float* cu = rd.cameraUpDir;
    const float* cr = rd.cameraRightDir;
    const bool noLens = rd.dofStrength == 0.f && __float_as_uint(ray.origin.x) != 0x80000000u && __float_as_uint(ray.origin.y) != 0x80000000u &&
                        __float_as_uint(ray.origin.z) != 0x80000000u &&
                        fabsf(cu[0]) <= kFltMax && fabsf(cu[1]) <= kFltMax && fabsf(cu[2]) <= kFltMax &&
                        fabsf(cr[0]) <= kFltMax && fabsf(cr[1]) <= kFltMax && fabsf(cr[2]) <= kFltMax;  // (NaN fails every comparison)
    if (noLens) {
        (void)pcgHash(seed);
        (void)pcgHash(seed);
    } else {
        float angle = randomFloat(seed) * 2.f * kPi;
        float cx = cosf(angle), cy = sinf(angle);
        float r = sqrtf(randomFloat(seed));
        float jx = (cx * r) * rd.dofStrength, jy = (cy * r) * rd.dofStrength;
        V3 rayOffset = fma3(ld3(rd.cameraUpDir), jy, ld3(rd.cameraRightDir) * jx);
        ray.origin = ray.origin + rayOffset;
    }
    ray.direction = normalize(focusPoint - ray.origin);
    return ray;
}

OKRT_DEV Ray primaryRay(const okrt_render_data& rd, uint32_t px, uint32_t py, uint32_t& seed) {
    return primaryRay(rd, rd.numAccumulationFrames, px, py, seed);
}

// One bounce of the path state after a hit (RaytracerCS.hlsl:495-513).
OKRT_DEV void scatter(const Surface& sf, Ray& ray, V3& contribution, V3& light, uint32_t& seed) {
    float metallicFactor = (sf.metallic * (1.f - sf.roughness)) >= randomFloat(seed) ? 1.f : 0.f;
    float specularFactor = (sf.specular * (1.f - sf.roughness)) >= randomFloat(seed) ? 1.f : 0.f;
    float transparencyFactor = sf.transparency >= randomFloat(seed) ? 1.f : 0.f;
    V3 reflectDir = findReflectDirection(ray.direction, sf.worldNormal, sf.roughness * (1.f - specularFactor), seed);
    V3 refractDir = findTransparencyBounce(ray.direction, sf.worldNormal, sf.ior, seed);
    V3 bounceDir = normalize(lerp(reflectDir, refractDir, transparencyFactor));
    V3 hitPoint = fma3(bounceDir, 0.001f, sf.worldPosition);
    contribution = contribution * lerp(sf.albedo, mk(1.f, 1.f, 1.f), metallicFactor);
    light = fma3(sf.emission * contribution, 1.f - transparencyFactor, light);
    ray.origin = hitPoint;
    ray.direction = bounceDir;
}

OKRT_DEV uint32_t toUnorm8(float x) { return (uint32_t)(saturate(x) * 255.f + 0.5f); }

// accumulate + resolve (RaytracerCS.hlsl:516-536)
OKRT_DEV void writePixel(const okrt_render_data& rd, V3 light, uint32_t bbCount, uint32_t triCount, float4* accum, uint32_t* result, size_t o) {
    uint32_t maxCount = rd.debugMaxCount;
    if (rd.debugMode == 1)
        light = bbCount > maxCount ? mk(1.f, 0.f, 0.f) : mk(1.f, 1.f, 1.f) * ((float)bbCount / (float)maxCount);
    else if (rd.debugMode == 2)
        light = triCount > maxCount ? mk(1.f, 0.f, 0.f) : mk(1.f, 1.f, 1.f) * ((float)triCount / (float)maxCount);
    uint32_t packed;
    if (rd.accumulationEnabled == 1) {
        float4 a = accum[o];
        a.x += light.x; a.y += light.y; a.z += light.z; a.w += 1.f;
        accum[o] = a;
        float n = (float)rd.numAccumulationFrames;
        packed = toUnorm8(a.x / n) | (toUnorm8(a.y / n) << 8) | (toUnorm8(a.z / n) << 16) | (toUnorm8(a.w / n) << 24);
    } else {
        packed = toUnorm8(light.x) | (toUnorm8(light.y) << 8) | (toUnorm8(light.z) << 16) | (255u << 24);
    }
    result[o] = packed;
}

}  // namespace okrt
