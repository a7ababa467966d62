/*
 * okrt_oracle.cpp — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A scalar C++ restatement of the reference's compute-shader path tracer, read from
 *   resources/shaders/RaytracerCS.hlsl      (main :421-537, findClosestHit :218-395, ...)
 *   resources/shaders/GPU-Utilities.hlsli   (RNG :18-25,51-78, refract :85-93, Collision :95-198)
 *   resources/shaders/GPU-Structs.hlsli     (table layouts :7-179)
 * of Olowerr/GPU-Raytracer. Each function cites the lines it follows. Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (libokrt.so) never does.
 *
 * PARITY STATUS: "parity unpinned" for the kernel — the reference ships no tests, golden vectors
 * or runnable kernel (HLSL cs_5_0 through D3D11 only). This file is a reading of
 * the HLSL, pinned by the known-answer tests in tests/ that were derived by hand from the
 * shader text. D3D fixed-function behaviour (bilinear weights, cube edges, UNORM rounding,
 * intrinsic lowering) is not in the reference repository; the choices made here are listed below.
 *
 * Floating-point lowering conventions (what fxc typically emits; applied identically by the HIP
 * kernels so that CPU and GPU agree to the last bit outside libm transcendentals):
 *   dot(a,b)      = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))            (dp3)
 *   cross(a,b).x  = fma(a.y,b.z, -(a.z*b.y))                       (mul + mad)
 *   lerp(a,b,t)   = fma(t, b-a, a)
 *   reflect(i,n)  = fma(n, -(2*dot(i,n)), i)
 *   normalize(v)  = v * (1/sqrt(dot(v,v)))    length(v) = sqrt(dot(v,v))
 *   mul(float4(v,w), M)[j] = fma chain over M's memory row j (see mat_point / mat_dir)
 *   x*y + z       = fma(x,y,z) wherever a product feeds an add directly in the source
 *   pow(x,y)      = exp2(y*log2(x))   (HLSL: NaN for x<0)
 *   min/max       = fminf/fmaxf (a NaN operand yields the other one, as in HLSL)
 *   x/y, sqrt     = IEEE correctly rounded
 * Build: g++ -O2 -mfma -ffp-contract=off (no fast-math), see oracle/Makefile.
 */
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/okrt.h"

namespace {

constexpr uint32_t UINT_MAX_U = ~0u;               // GPU-Structs.hlsli:1
constexpr float FLT_MAX_F = 3.402823466e+38F;      // GPU-Structs.hlsli:2
constexpr float PI_F = 3.14159265f;                // GPU-Structs.hlsli:3
constexpr float AIR_REFRACTION_INDEX = 1.f;        // GPU-Structs.hlsli:4

inline float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 v3(const float* p) { return V3{p[0], p[1], p[2]}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 fma3(V3 a, float s, V3 c) { return {fma_(a.x, s, c.x), fma_(a.y, s, c.y), fma_(a.z, s, c.z)}; }
inline V3 fma3(V3 a, V3 b, V3 c) { return {fma_(a.x, b.x, c.x), fma_(a.y, b.y, c.y), fma_(a.z, b.z, c.z)}; }
inline float dot(V3 a, V3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
inline V3 cross(V3 a, V3 b) {
    return {fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x))};
}
inline float rsq(float x) { return 1.0f / sqrtf(x); }
inline V3 normalize(V3 v) { return v * rsq(dot(v, v)); }
inline float length(V3 v) { return sqrtf(dot(v, v)); }
inline float lerp(float a, float b, float t) { return fma_(t, b - a, a); }
inline V3 lerp(V3 a, V3 b, float t) { return {lerp(a.x, b.x, t), lerp(a.y, b.y, t), lerp(a.z, b.z, t)}; }
inline V3 reflect(V3 i, V3 n) {
    float d = dot(i, n);
    return fma3(n, -(d + d), i);
}
inline float hlsl_pow(float x, float y) { return exp2f(y * log2f(x)); }
inline float saturate(float x) { return fminf(fmaxf(x, 0.f), 1.f); }  // NaN -> 0
inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// ---- table layouts (GPU-Structs.hlsli) ------------------------------------------------------
struct Material {  // :37-71, 68 B
    float albedo[3];
    uint32_t albedoTex;
    float roughness;
    uint32_t roughnessTex;
    float metallic;
    uint32_t metallicTex;
    float specular;
    uint32_t specularTex;
    uint32_t normalMapIdx;
    float emissionColour[3];
    float emissionPower;
    float transparency;
    float indexOfRefraction;
};
struct Sphere {  // :86-93, 84 B
    float position[3];
    Material material;
    float radius;
};
struct Triangle {  // :95-98, 36 B
    float p[3][3];
};
struct VertexInfo {  // :100-106, 44 B
    float normal[3];
    float uv[2];
    float tangent[3];
    float bitangent[3];
};
struct TriangleInfo {  // :108-111, 132 B
    VertexInfo v[3];
};
struct Mesh {  // :113-125, 232 B
    float transformMatrix[16];
    float inverseTransformMatrix[16];
    uint32_t triangleStartIdx, triangleEndIdx;
    float bbMin[3], bbMax[3];
    Material material;
    uint32_t bvhNodeStartIdx;
};
struct DirectionalLight {  // :127-135, 32 B
    float colour[3];
    float intensity;
    float effectiveAngle;
    float direction[3];
};
struct PointLight {  // :137-145, 32 B
    float colour[3];
    float intensity;
    float radius;
    float position[3];
};
struct SpotLight {  // :147-157, 48 B
    float colour[3];
    float intensity;
    float radius;
    float maxAngle;
    float position[3];
    float direction[3];
};
struct BvhNode {  // :159-165, 36 B
    float bbMin[3], bbMax[3];
    uint32_t triStart, triEnd, firstChildIdx;
};
struct OctTreeNode {  // :167-179, 48 B
    float bbMin[3], bbMax[3];
    uint32_t meshesStartIdx, meshesEndIdx, spheresStartIdx, spheresEndIdx, firstChildIdx, numChildren;
};
static_assert(sizeof(Material) == 68 && sizeof(Sphere) == 84 && sizeof(Triangle) == 36, "abi");
static_assert(sizeof(TriangleInfo) == 132 && sizeof(Mesh) == 232 && sizeof(BvhNode) == 36, "abi");
static_assert(sizeof(OctTreeNode) == 48 && sizeof(DirectionalLight) == 32, "abi");
static_assert(sizeof(PointLight) == 32 && sizeof(SpotLight) == 48, "abi");
static_assert(sizeof(okrt_render_data) == 240, "abi");

struct Ray {
    V3 origin, direction;
};

// mul(float4(v,1), M) and mul(float4(v,0), M) for a matrix uploaded as transpose(G) in glm
// memory order: HLSL reads it column_major, so component j = sum_i mem[4j+i] * v_i  (== G*v).
inline V3 mat_point(const float* m, V3 v) {
    auto row = [&](int j) {
        float r = m[4 * j + 0] * v.x;
        r = fma_(m[4 * j + 1], v.y, r);
        r = fma_(m[4 * j + 2], v.z, r);
        return r + m[4 * j + 3];  // * 1.0
    };
    return {row(0), row(1), row(2)};
}
inline V3 mat_dir(const float* m, V3 v) {
    auto row = [&](int j) {
        float r = m[4 * j + 0] * v.x;
        r = fma_(m[4 * j + 1], v.y, r);
        return fma_(m[4 * j + 2], v.z, r);
    };
    return {row(0), row(1), row(2)};
}
inline void mat_point4(const float* m, V3 v, float out[4]) {
    for (int j = 0; j < 4; ++j) {
        float r = m[4 * j + 0] * v.x;
        r = fma_(m[4 * j + 1], v.y, r);
        r = fma_(m[4 * j + 2], v.z, r);
        out[j] = r + m[4 * j + 3];
    }
}

// ---- RNG (GPU-Utilities.hlsli:18-25, 51-78) --------------------------------------------------
inline uint32_t pcg_hash(uint32_t& seed) {
    seed *= 747796405u + 2891336453u;  // sic: precedence makes this one multiply (quirk Q2)
    seed = ((seed >> ((seed >> 28u) + 4u)) ^ seed) * 277803737u;
    seed = (seed >> 22u) ^ seed;
    return seed;
}
inline float randomFloat(uint32_t& seed) { return (float)pcg_hash(seed) / (float)UINT_MAX_U; }
inline float randomFloatNormalDistribution(uint32_t& seed) {
    float theta = (2 * PI_F) * randomFloat(seed);
    float rho = sqrtf(-2 * logf(randomFloat(seed)));
    return rho * cosf(theta);
}
inline V3 getRandomVector(uint32_t& seed) {
    float x = randomFloatNormalDistribution(seed);
    float y = randomFloatNormalDistribution(seed);
    float z = randomFloatNormalDistribution(seed);
    return normalize(v3(x, y, z));
}
inline void randomPointInCircle(uint32_t& seed, float& ox, float& oy) {
    float angle = randomFloat(seed) * 2.f * PI_F;
    float cx = cosf(angle), cy = sinf(angle);
    float r = sqrtf(randomFloat(seed));
    ox = cx * r;
    oy = cy * r;
}
inline float clampedDot(V3 v, V3 u) { return fmaxf(dot(v, u), 0.f); }  // :80-83

// user-defined refract shadowing the intrinsic (GPU-Utilities.hlsli:85-93, quirk Q8)
inline V3 refract_(V3 direction, V3 normal, float refractionRatio) {
    float cosTheta = fminf(dot(-direction, normal), 1.f);
    V3 rayOutPerpendicular = fma3(normal, cosTheta, direction) * refractionRatio;
    float ropLengthSqrd = dot(rayOutPerpendicular, rayOutPerpendicular);
    float k = -sqrtf(fabsf(1.f - ropLengthSqrd));
    return fma3(normal, k, rayOutPerpendicular);
}

// ---- Collision (GPU-Utilities.hlsli:95-198) ---------------------------------------------------
inline float RayAndSphere(const Ray& ray, V3 spherePos, float radius) {  // :98-114
    V3 rayToSphere = spherePos - ray.origin;
    float distToClosestPoint = dot(rayToSphere, ray.direction);
    float rayToSphereMagSqrd = dot(rayToSphere, rayToSphere);
    float sphereRadiusSqrd = radius * radius;
    if (distToClosestPoint < 0.f && rayToSphereMagSqrd > sphereRadiusSqrd) return -1.f;
    float sideA = fma_(-distToClosestPoint, distToClosestPoint, rayToSphereMagSqrd);
    if (sideA > sphereRadiusSqrd) return -1.f;
    float sideB = sqrtf(sphereRadiusSqrd - sideA);
    return rayToSphereMagSqrd > sphereRadiusSqrd ? distToClosestPoint - sideB : distToClosestPoint + sideB;
}
inline float RayAndTriangle(const Ray& ray, V3 p0, V3 p1, V3 p2, float& u, float& v) {  // :117-155
    const float EPSILON = 0.000001f;
    V3 E1 = p0 - p2, E2 = p1 - p2;
    V3 cross1 = cross(ray.direction, E2);
    float determinant = dot(E1, cross1);
    if (determinant < EPSILON && determinant > -EPSILON) return -1.f;
    float inverseDet = 1.f / determinant;
    V3 rayMoved = ray.origin - p2;
    u = dot(rayMoved, cross1) * inverseDet;
    if (u < -EPSILON) return -1.f;
    V3 cross2 = cross(rayMoved, E1);
    v = dot(ray.direction, cross2) * inverseDet;
    if (v < -EPSILON || u + v > 1.0f) return -1.f;
    float t1 = dot(E2, cross2) * inverseDet;
    if (t1 < -EPSILON) return -1.f;
    return t1;
}
inline float RayAndAABBDist(const Ray& ray, const float* bbMin, const float* bbMax) {  // :183-197
    V3 inv = {1.f / ray.direction.x, 1.f / ray.direction.y, 1.f / ray.direction.z};
    V3 tMin = (v3(bbMin) - ray.origin) * inv;
    V3 tMax = (v3(bbMax) - ray.origin) * inv;
    V3 t1 = {fminf(tMin.x, tMax.x), fminf(tMin.y, tMax.y), fminf(tMin.z, tMax.z)};
    V3 t2 = {fmaxf(tMin.x, tMax.x), fmaxf(tMin.y, tMax.y), fmaxf(tMin.z, tMax.z)};
    float distFar = fminf(fminf(t2.x, t2.y), t2.z);
    float distNear = fmaxf(fmaxf(t1.x, t1.y), t1.z);
    bool didHit = distFar >= distNear && distFar > 0.f;
    return didHit ? distNear : FLT_MAX_F;
}

// ---- context --------------------------------------------------------------------------------
struct Counters {
    uint64_t pixels = 0, segments = 0, n_oct = 0, n_mesh = 0, n_node = 0, n_tri = 0, mesh_hits = 0,
             sphere_hits = 0, tex_taps = 0, env_taps = 0;
    void add(const Counters& o) {
        pixels += o.pixels; segments += o.segments; n_oct += o.n_oct; n_mesh += o.n_mesh;
        n_node += o.n_node; n_tri += o.n_tri; mesh_hits += o.mesh_hits; sphere_hits += o.sphere_hits;
        tex_taps += o.tex_taps; env_taps += o.env_taps;
    }
};

}  // namespace

struct okrt_ctx {  // the oracle's context (same opaque name as the product's; separate library)
    std::vector<Triangle> triPos;
    std::vector<TriangleInfo> triInfo;
    std::vector<BvhNode> bvh;
    std::vector<OctTreeNode> oct;
    std::vector<Sphere> spheres;
    std::vector<Mesh> meshes;
    std::vector<DirectionalLight> dirLights;
    std::vector<PointLight> pointLights;
    std::vector<SpotLight> spotLights;
    std::vector<uint8_t> textures;
    uint32_t texCount = 0, texW = 0, texH = 0;
    std::vector<uint8_t> env;
    uint32_t envFace = 0;
    uint32_t W = 0, H = 0;
    uint32_t stripeRows = 0, stripeIndex = 0, stripeStride = 1;
    std::vector<uint32_t> rowMap;  // local row -> global y
    std::vector<float> accum;      // rows*W*4
    std::vector<uint8_t> result;   // rows*W*4
    std::vector<uint32_t> bbImage, triImage;  // per-pixel counters of the last render
    int maxBounces = 1;
    int threads = 0;  // 0 = hardware_concurrency
    // SENSITIVITY MODELS of the unpinned D3D11 sampler (okrt_oracle_set_sampler_model; tests/test_sampler_sensitivity.py): how far
    // can the picture move if the hardware filters with fixed-point weights and across cube edges? Never on by default.
    int samplerWeightBits = 0;   // 0 = exact fp32 weights; n = bilinear weights snapped to n fractional bits (D3D11 requires >= 8)
    bool seamlessCube = false;   // bilinear footprints that leave a cube face take their texels from the adjacent face
    bool countersOn = false;
    Counters counters;
    std::string err;
};

namespace {

struct Payload {  // RaytracerCS.hlsl:10-17
    bool hit;
    float distance;
    Material material;
    V3 worldPosition;
    V3 worldNormal;
};

Material materialCreate() {  // GPU-Structs.hlsli:52-70
    Material m;
    m.albedo[0] = m.albedo[1] = m.albedo[2] = 0.f;
    m.albedoTex = UINT_MAX_U;
    m.roughness = 0.f; m.roughnessTex = UINT_MAX_U;
    m.metallic = 0.f; m.metallicTex = UINT_MAX_U;
    m.specular = 0.f; m.specularTex = UINT_MAX_U;
    m.normalMapIdx = UINT_MAX_U;
    m.emissionColour[0] = m.emissionColour[1] = m.emissionColour[2] = 0.f;
    m.emissionPower = 0.f;
    m.transparency = 0.f;
    m.indexOfRefraction = 0.f;
    return m;
}
inline bool isValidIdx(uint32_t idx) { return idx != UINT_MAX_U; }

// D3D11 fixed-function sampling is not in the reference repository (parity unpinned). Chosen:
// texel centres at +0.5, exact fp32 weights, UNORM8 -> c/255, blend = lerp(lerp(t00,t10,fx),
// lerp(t01,t11,fx), fy), WRAP addressing per RayTracer.cpp:99-109, slice = round-half-even(idx)
// clamped to the array, non-finite coordinates sample as 0.
inline V3 texel(const uint8_t* p) { return {p[0] / 255.f, p[1] / 255.f, p[2] / 255.f}; }
// sensitivity model: a bilinear weight as fixed point with `bits` fractional bits (round to nearest; 1.0 stays representable)
inline float snapWeight(float w, int bits) {
    if (bits <= 0) return w;
    const float s = (float)(1 << bits);
    return floorf(fma_(w, s, 0.5f)) / s;
}
inline int wrapi(int i, int n) {
    int r = i % n;
    return r < 0 ? r + n : r;
}
V3 sampleTexture(const okrt_ctx& c, uint32_t textureIdx, float u, float v, Counters& cnt) {  // :133-136
    cnt.tex_taps++;
    if (c.texCount == 0) return {0, 0, 0};  // unbound SRV reads zero in D3D11
    float fidx = (float)textureIdx;
    uint32_t slice = (uint32_t)std::min<float>(nearbyintf(fidx), (float)(c.texCount - 1));
    if (!std::isfinite(u)) u = 0.f;
    if (!std::isfinite(v)) v = 0.f;
    const int W = (int)c.texW, H = (int)c.texH;
    float x = fma_(u, (float)W, -0.5f), y = fma_(v, (float)H, -0.5f);
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = snapWeight(x - fx0, c.samplerWeightBits), fy = snapWeight(y - fy0, c.samplerWeightBits);
    // wrap in float first so huge coordinates stay in int range
    float wx = fx0 - floorf(fx0 / (float)W) * (float)W;
    float wy = fy0 - floorf(fy0 / (float)H) * (float)H;
    int x0 = wrapi((int)wx, W), y0 = wrapi((int)wy, H);
    int x1 = wrapi(x0 + 1, W), y1 = wrapi(y0 + 1, H);
    const uint8_t* base = c.textures.data() + (size_t)slice * W * H * 4;
    V3 t00 = texel(base + ((size_t)y0 * W + x0) * 4), t10 = texel(base + ((size_t)y0 * W + x1) * 4);
    V3 t01 = texel(base + ((size_t)y1 * W + x0) * 4), t11 = texel(base + ((size_t)y1 * W + x1) * 4);
    return lerp(lerp(t00, t10, fx), lerp(t01, t11, fx), fy);
}

// TextureCube.SampleLevel (RaytracerCS.hlsl:86-88). Face selection follows the AMD v_cubeid rule
// (z wins ties, then y); face coordinates are the D3D/GL table; bilinear with clamp-to-edge
// inside the face (D3D10+ hardware filters seamlessly across edges: differs only within half a
// texel of a face border; covered by the image tolerance).
// major axis / face / face coordinates of a direction (the D3D / GL table; z wins ties, then y)
inline void cubeFace(V3 d, int& face, float& sc, float& tc, float& ma) {
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
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
}
// Sensitivity model (seamlessCube): texel (x, y) of `face`, where x or y may lie one texel outside the face — then the texel of the
// ADJACENT face that the direction through that texel centre falls on (what seamless hardware filtering reads; at a cube corner
// hardware averages the three faces' corner texels, this takes one of them).
V3 cubeTexelSeamless(const okrt_ctx& c, int face, int x, int y) {
    const int F = (int)c.envFace;
    if (x < 0 || x >= F || y < 0 || y >= F) {
        const float sc = (2.f * ((float)x + 0.5f)) / (float)F - 1.f, tc = (2.f * ((float)y + 0.5f)) / (float)F - 1.f;
        V3 d;
        switch (face) {  // inverse of the table in cubeFace with ma = 1
            case 0: d = v3(1.f, -tc, -sc); break;
            case 1: d = v3(-1.f, -tc, sc); break;
            case 2: d = v3(sc, 1.f, tc); break;
            case 3: d = v3(sc, -1.f, -tc); break;
            case 4: d = v3(sc, -tc, 1.f); break;
            default: d = v3(-sc, -tc, -1.f); break;
        }
        float s2, t2, ma;
        cubeFace(d, face, s2, t2, ma);
        x = std::clamp((int)floorf(fma_(s2 / ma, 0.5f, 0.5f) * (float)F), 0, F - 1);
        y = std::clamp((int)floorf(fma_(t2 / ma, 0.5f, 0.5f) * (float)F), 0, F - 1);
    }
    return texel(c.env.data() + ((size_t)face * F * F + (size_t)y * F + x) * 4);
}

V3 getEnvironmentLight(const okrt_ctx& c, V3 d, Counters& cnt) {
    cnt.env_taps++;
    if (c.envFace == 0) return {0, 0, 0};
    int face;
    float sc, tc, ma;
    cubeFace(d, face, sc, tc, ma);
    float u = fma_(sc / ma, 0.5f, 0.5f), v = fma_(tc / ma, 0.5f, 0.5f);
    if (!std::isfinite(u)) u = 0.f;
    if (!std::isfinite(v)) v = 0.f;
    const int F = (int)c.envFace;
    float x = fma_(u, (float)F, -0.5f), y = fma_(v, (float)F, -0.5f);
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = snapWeight(x - fx0, c.samplerWeightBits), fy = snapWeight(y - fy0, c.samplerWeightBits);
    if (c.seamlessCube) {
        const int xi = (int)fx0, yi = (int)fy0;
        V3 t00 = cubeTexelSeamless(c, face, xi, yi), t10 = cubeTexelSeamless(c, face, xi + 1, yi);
        V3 t01 = cubeTexelSeamless(c, face, xi, yi + 1), t11 = cubeTexelSeamless(c, face, xi + 1, yi + 1);
        return lerp(lerp(t00, t10, fx), lerp(t01, t11, fx), fy);
    }
    int x0 = std::clamp((int)fx0, 0, F - 1), x1 = std::clamp((int)fx0 + 1, 0, F - 1);
    int y0 = std::clamp((int)fy0, 0, F - 1), y1 = std::clamp((int)fy0 + 1, 0, F - 1);
    const uint8_t* base = c.env.data() + (size_t)face * F * F * 4;
    V3 t00 = texel(base + ((size_t)y0 * F + x0) * 4), t10 = texel(base + ((size_t)y0 * F + x1) * 4);
    V3 t01 = texel(base + ((size_t)y1 * F + x0) * 4), t11 = texel(base + ((size_t)y1 * F + x1) * 4);
    return lerp(lerp(t00, t10, fx), lerp(t01, t11, fx), fy);
}

float reflectance(float cosine, float reflectionIdx) {  // :100-105 (quirk Q5: 1 + r0)
    float r0 = (1.f - reflectionIdx) / (1.f + reflectionIdx);
    r0 *= r0;
    return fma_(1.f + r0, hlsl_pow(1.f - cosine, 5.f), r0);
}
V3 findReflectDirection(V3 direction, V3 normal, float roughness, uint32_t& seed) {  // :107-112
    V3 diffuseReflection = normalize(normal + getRandomVector(seed));
    V3 specularReflection = reflect(direction, normal);
    return normalize(lerp(specularReflection, diffuseReflection, roughness));
}
V3 findTransparencyBounce(V3 direction, V3 normal, float refractionIdx, uint32_t& seed) {  // :114-131
    bool hitFrontFace = dot(direction, normal) < 0.f;
    float refractionRatio = hitFrontFace ? AIR_REFRACTION_INDEX / refractionIdx : refractionIdx;
    if (!hitFrontFace) normal = normal * -1.f;
    float cos_theta = dot(-direction, normal);
    float sin_theta = sqrtf(fma_(-cos_theta, cos_theta, 1.f));
    bool cannot_refract = refractionRatio * sin_theta > 1.f;
    // SM5 `||` does not short-circuit: the draw is always consumed (quirk Q6)
    float refl = reflectance(cos_theta, refractionRatio);
    float rnd = randomFloat(seed);
    if (cannot_refract | (refl > rnd)) direction = reflect(direction, normal);
    return refract_(direction, normal, refractionRatio);  // also after reflecting (quirk Q7)
}

void findMaterialTextureColours(const okrt_ctx& c, Material& m, float u, float v, Counters& cnt) {  // :138-165
    if (isValidIdx(m.albedoTex)) {
        V3 t = sampleTexture(c, m.albedoTex, u, v, cnt);
        m.albedo[0] = t.x; m.albedo[1] = t.y; m.albedo[2] = t.z;
    }
    if (isValidIdx(m.roughnessTex)) m.roughness = clampf(sampleTexture(c, m.roughnessTex, u, v, cnt).x, m.roughness, 1.f);
    if (isValidIdx(m.metallicTex)) m.metallic = clampf(sampleTexture(c, m.metallicTex, u, v, cnt).x, m.metallic, 1.f);
    if (isValidIdx(m.specularTex)) m.specular = clampf(sampleTexture(c, m.specularTex, u, v, cnt).x, m.specular, 1.f);
}
V3 sampleNormalMap(const okrt_ctx& c, uint32_t idx, float u, float v, V3 normal, V3 tangent, V3 bitangent, Counters& cnt) {  // :167-175
    V3 s = sampleTexture(c, idx, u, v, cnt);
    s = {fma_(s.x, 2.f, -1.f), fma_(s.y, 2.f, -1.f), fma_(s.z, 2.f, -1.f)};
    // mul(s, float3x3(T,B,N)) = s.x*T + s.y*B + s.z*N
    V3 r = tangent * s.x;
    r = fma3(bitangent, s.y, r);
    r = fma3(normal, s.z, r);
    return normalize(r);
}
inline V3 bary3(V3 uvw, const float* a, const float* b, const float* c3) {  // GPU-Utilities.hlsli:3-6
    V3 r = v3(a) * uvw.x;
    r = fma3(v3(b), uvw.y, r);
    return fma3(v3(c3), uvw.z, r);
}

Payload findClosestHit(const okrt_ctx& c, const okrt_render_data& rd, const Ray& ray, uint32_t& bbCheckCount,
                       uint32_t& triCheckCount, Counters& cnt) {  // RaytracerCS.hlsl:218-395
    Payload payload;
    payload.hit = false;
    payload.distance = FLT_MAX_F;
    payload.material = materialCreate();
    payload.worldPosition = {0, 0, 0};
    payload.worldNormal = {0, 0, 0};
    cnt.segments++;

    uint32_t hitIdx = UINT_MAX_U;
    uint32_t hitType = 0;

    for (uint32_t i = 0; i < rd.numSpheres; i++) {  // :232-242
        float distanceToHit = RayAndSphere(ray, v3(c.spheres[i].position), c.spheres[i].radius);
        if (distanceToHit > 0.f && distanceToHit < payload.distance) {
            payload.distance = distanceToHit;
            hitIdx = i;
            hitType = 0;
        }
    }

    constexpr uint32_t OCT_MAX_STACK_SIZE = 20, BVH_MAX_STACK_SIZE = 50;  // :244-250
    uint32_t octStack[OCT_MAX_STACK_SIZE];
    uint32_t octStackSize = 0;
    if (!c.oct.empty()) octStack[octStackSize++] = 0;  // an unbound t5 reads a zero node: nothing to do
    uint32_t bvhStack[BVH_MAX_STACK_SIZE];
    uint32_t bvhStackSize = 0;

    uint32_t triHitIdx = UINT_MAX_U;
    V3 hitBary = {0, 0, 0};

    while (octStackSize > 0) {  // :256-334
        uint32_t octIdx = octStack[--octStackSize];
        const OctTreeNode& octNode = c.oct[octIdx];
        bbCheckCount += 1;
        cnt.n_oct++;
        if (RayAndAABBDist(ray, octNode.bbMin, octNode.bbMax) >= payload.distance) continue;

        for (uint32_t i = octNode.meshesStartIdx; i < octNode.meshesEndIdx; i++) {  // :265-328
            bbCheckCount += 1;
            cnt.n_mesh++;
            const Mesh& mesh = c.meshes[i];
            bvhStack[0] = mesh.bvhNodeStartIdx;
            bvhStackSize = 1;
            Ray localRay;
            localRay.origin = mat_point(mesh.inverseTransformMatrix, ray.origin);
            localRay.direction = normalize(mat_dir(mesh.inverseTransformMatrix, ray.direction));

            while (bvhStackSize > 0) {  // :278-327
                uint32_t nodeIdx = bvhStack[--bvhStackSize];
                const BvhNode& node = c.bvh[nodeIdx];
                bbCheckCount += 1;
                cnt.n_node++;
                if (RayAndAABBDist(localRay, node.bbMin, node.bbMax) >= payload.distance) continue;
                if (isValidIdx(node.firstChildIdx)) {
                    if (bvhStackSize + 2 <= BVH_MAX_STACK_SIZE) {
                        bvhStack[bvhStackSize++] = node.firstChildIdx;
                        bvhStack[bvhStackSize++] = node.firstChildIdx + 1;
                    }
                    continue;
                }
                for (uint32_t j = node.triStart; j < node.triEnd; j++) {  // :295-326
                    const Triangle& tri = c.triPos[j];
                    float bu = 0.f, bv = 0.f;
                    triCheckCount += 1;
                    cnt.n_tri++;
                    float distanceToHit = RayAndTriangle(localRay, v3(tri.p[0]), v3(tri.p[1]), v3(tri.p[2]), bu, bv);
                    if (distanceToHit <= 0.f) continue;
                    // world-space distance for the comparison (:311-314)
                    distanceToHit = length(mat_dir(mesh.transformMatrix, localRay.direction * distanceToHit));
                    if (distanceToHit < payload.distance) {
                        payload.distance = distanceToHit;
                        hitIdx = i;
                        hitType = 1;
                        triHitIdx = j;
                        hitBary.x = bu;
                        hitBary.y = bv;
                        hitBary.z = 1.f - (bu + bv);
                    }
                }
            }
        }
        for (uint32_t i = 0; i < octNode.numChildren; i++)  // :330-333
            if (octStackSize < OCT_MAX_STACK_SIZE) octStack[octStackSize++] = octNode.firstChildIdx + i;
    }

    payload.hit = hitIdx != UINT_MAX_U;
    payload.worldPosition = fma3(ray.direction, payload.distance, ray.origin);
    if (!payload.hit) return payload;  // quirk Q4: the reference reads sphereData[UINT_MAX] (zeros); result unused

    if (hitType == 0) {  // sphere :342-362
        cnt.sphere_hits++;
        const Sphere& s = c.spheres[hitIdx];
        payload.material = s.material;
        V3 hitNormal = normalize(payload.worldPosition - v3(s.position));
        // normalToUV :77-84
        float su = fma_(atan2f(hitNormal.x, hitNormal.z), 1.f / (2 * PI_F), 0.5f);
        float sv = fma_(hitNormal.y, 0.5f, 0.5f);
        if (isValidIdx(payload.material.normalMapIdx)) {
            V3 tangent = normalize(cross(v3(0.f, 1.f, 0.f), hitNormal));
            V3 bitangent = normalize(cross(tangent, hitNormal));
            payload.worldNormal = sampleNormalMap(c, payload.material.normalMapIdx, su, sv, hitNormal, tangent, bitangent, cnt);
        } else {
            payload.worldNormal = hitNormal;
        }
        findMaterialTextureColours(c, payload.material, su, sv, cnt);
    } else {  // mesh :364-391
        cnt.mesh_hits++;
        const Mesh& mesh = c.meshes[hitIdx];
        payload.material = mesh.material;
        const TriangleInfo& ti = c.triInfo[triHitIdx];
        float lu = fma_(ti.v[2].uv[0], hitBary.z, fma_(ti.v[1].uv[0], hitBary.y, ti.v[0].uv[0] * hitBary.x));
        float lv = fma_(ti.v[2].uv[1], hitBary.z, fma_(ti.v[1].uv[1], hitBary.y, ti.v[0].uv[1] * hitBary.x));
        V3 normal = mat_dir(mesh.transformMatrix, bary3(hitBary, ti.v[0].normal, ti.v[1].normal, ti.v[2].normal));
        V3 tangent = mat_dir(mesh.transformMatrix, bary3(hitBary, ti.v[0].tangent, ti.v[1].tangent, ti.v[2].tangent));
        V3 bitangent = mat_dir(mesh.transformMatrix, bary3(hitBary, ti.v[0].bitangent, ti.v[1].bitangent, ti.v[2].bitangent));
        normal = normalize(normal);
        if (isValidIdx(payload.material.normalMapIdx)) {
            tangent = normalize(tangent);
            bitangent = normalize(bitangent);
            payload.worldNormal = sampleNormalMap(c, payload.material.normalMapIdx, lu, lv, normal, tangent, bitangent, cnt);
        } else {
            payload.worldNormal = normal;
        }
        findMaterialTextureColours(c, payload.material, lu, lv, cnt);
    }
    return payload;
}

V3 getLighting(const Ray& ray, const Payload& hp, V3 lightPos, float lightRadius, V3 lightColour, float lightIntensity) {  // :397-418
    float dist = RayAndSphere(ray, lightPos, lightRadius);
    if (dist < 0.f) return {0, 0, 0};
    float lightStrengthModifier = 1.f;
    if (hp.hit && hp.distance > dist)
        lightStrengthModifier *= clampedDot(ray.direction, -hp.worldNormal);
    else if (hp.hit && hp.distance < dist)
        lightStrengthModifier *= hp.material.transparency * clampedDot(ray.direction, -hp.worldNormal);
    return (lightColour * lightIntensity) * lightStrengthModifier;
}

// One sample of one pixel: RaytracerCS.hlsl:421-537 up to (not including) the accumulate.
V3 tracePixel(const okrt_ctx& c, const okrt_render_data& rd, uint32_t px, uint32_t py, uint32_t& bbCheckCount,
              uint32_t& triCheckCount, Counters& cnt) {
    cnt.pixels++;
    uint32_t seed = px + (py + 74813u) * rd.textureDims[0] * (rd.numAccumulationFrames + 1u);  // :430

    // createRay :177-206
    float posx = (float)px, posy = (float)(rd.textureDims[1] - py);
    posx += randomFloat(seed) - 0.5f;
    posy += randomFloat(seed) - 0.5f;
    posx /= (float)rd.textureDims[0];
    posy /= (float)rd.textureDims[1];
    posx = fma_(posx, 2.f, -1.f);
    posy = fma_(posy, 2.f, -1.f);
    float target[4];
    mat_point4(rd.cameraInverseProjectionMatrix, v3(posx, posy, rd.cameraNearZ), target);
    Ray ray;
    ray.origin = v3(rd.cameraPosition);
    V3 tdir = normalize(v3(target[0] / target[2], target[1] / target[2], target[2] / target[2]));
    ray.direction = mat_dir(rd.cameraInverseViewMatrix, tdir);

    {  // applyDOF :208-216
        float jx, jy;
        randomPointInCircle(seed, jx, jy);
        jx *= rd.dofStrength;
        jy *= rd.dofStrength;
        V3 rayOffset = fma3(v3(rd.cameraUpDir), jy, v3(rd.cameraRightDir) * jx);
        V3 focusPoint = fma3(ray.direction, rd.dofDistance + rd.cameraNearZ, ray.origin);
        ray.origin = ray.origin + rayOffset;
        ray.direction = normalize(focusPoint - ray.origin);
    }

    V3 light = {0, 0, 0};
    V3 contribution = {1, 1, 1};

    for (uint32_t i = 0; i <= (uint32_t)c.maxBounces; i++) {  // :444-514
        Payload hitData = findClosestHit(c, rd, ray, bbCheckCount, triCheckCount, cnt);

        for (uint32_t p = 0; p < rd.numPointLights; p++) {  // :448-454
            const PointLight& L = c.pointLights[p];
            V3 lightning = getLighting(ray, hitData, v3(L.position), L.radius, v3(L.colour), L.intensity);
            light = fma3(lightning, contribution, light);
        }
        for (uint32_t s = 0; s < rd.numSpotLights; s++) {  // :456-467
            const SpotLight& L = c.spotLights[s];
            V3 rayToLight = normalize(fma3(getRandomVector(seed), 0.01f, v3(L.position)) - ray.origin);
            float cosTheta = dot(rayToLight, -v3(L.direction));
            if (cosTheta < L.maxAngle) continue;
            V3 lightning = getLighting(ray, hitData, v3(L.position), L.radius, v3(L.colour), L.intensity);
            light = fma3(lightning, contribution, light);
        }
        for (uint32_t d = 0; d < rd.numDirLights; d++) {  // :469-487
            const DirectionalLight& L = c.dirLights[d];
            if (clampedDot(ray.direction, -v3(L.direction)) < L.effectiveAngle) continue;
            float lightStrengthModifier = 1.f;
            if (hitData.hit && i)
                lightStrengthModifier *= hitData.material.transparency * clampedDot(ray.direction, -hitData.worldNormal);
            else if (!i)
                lightStrengthModifier = 0.f;
            light = fma3(((v3(L.colour) * L.intensity) * contribution), lightStrengthModifier, light);
        }

        if (!hitData.hit) {  // :489-493
            light = fma3(getEnvironmentLight(c, ray.direction, cnt), contribution, light);
            break;
        }

        const Material& material = hitData.material;
        float metallicFactor = (material.metallic * (1.f - material.roughness)) >= randomFloat(seed) ? 1.f : 0.f;  // :499
        float specularFactor = (material.specular * (1.f - material.roughness)) >= randomFloat(seed) ? 1.f : 0.f;  // :500
        float transparencyFactor = material.transparency >= randomFloat(seed) ? 1.f : 0.f;                        // :501

        V3 reflectDir = findReflectDirection(ray.direction, hitData.worldNormal, material.roughness * (1.f - specularFactor), seed);
        V3 refractDir = findTransparencyBounce(ray.direction, hitData.worldNormal, material.indexOfRefraction, seed);
        V3 bounceDir = normalize(lerp(reflectDir, refractDir, transparencyFactor));
        V3 hitPoint = fma3(bounceDir, 0.001f, hitData.worldPosition);

        contribution = contribution * lerp(v3(material.albedo), v3(1.f, 1.f, 1.f), metallicFactor);  // :509
        light = fma3(((v3(material.emissionColour) * material.emissionPower) * contribution), 1.f - transparencyFactor, light);  // :510

        ray.origin = hitPoint;
        ray.direction = bounceDir;
    }

    uint32_t debugModeMaxCount = rd.debugMaxCount;  // :516-520
    if (rd.debugMode == 1)
        light = bbCheckCount > debugModeMaxCount ? v3(1.f, 0.f, 0.f) : v3(1.f, 1.f, 1.f) * ((float)bbCheckCount / (float)debugModeMaxCount);
    else if (rd.debugMode == 2)
        light = triCheckCount > debugModeMaxCount ? v3(1.f, 0.f, 0.f) : v3(1.f, 1.f, 1.f) * ((float)triCheckCount / (float)debugModeMaxCount);
    return light;
}

inline uint8_t toUnorm8(float x) {  // D3D11 float -> UNORM: saturate (NaN -> 0), scale, round to nearest
    return (uint8_t)(saturate(x) * 255.f + 0.5f);
}

void buildRowMap(okrt_ctx& c) {
    c.rowMap.clear();
    if (c.stripeRows == 0) return;
    for (uint32_t y = 0; y < c.H; ++y) {
        uint32_t stripe = y / c.stripeRows;
        if (stripe % c.stripeStride == c.stripeIndex) c.rowMap.push_back(y);
    }
    size_t n = (size_t)c.rowMap.size() * c.W;
    c.accum.assign(n * 4, 0.f);
    c.result.assign(n * 4, 0);
    c.bbImage.assign(n, 0);
    c.triImage.assign(n, 0);
}

okrt_status fail(okrt_ctx* c, okrt_status s, const char* msg) {
    if (c) c->err = msg;
    return s;
}

template <class T>
okrt_status setVec(okrt_ctx* c, std::vector<T>& v, const void* data, uint32_t elem, uint32_t count) {
    if (elem != sizeof(T)) return fail(c, OKRT_ERR_INVALID_ARG, "element stride does not match the reference layout");
    if (!data || !count) {
        v.clear();
        return OKRT_OK;
    }
    v.resize(count);
    memcpy(v.data(), data, (size_t)count * sizeof(T));
    return OKRT_OK;
}

}  // namespace

extern "C" {

okrt_status okrt_oracle_create(okrt_ctx** out) {
    if (!out) return OKRT_ERR_INVALID_ARG;
    *out = new okrt_ctx();
    return OKRT_OK;
}
okrt_status okrt_oracle_destroy(okrt_ctx* c) {
    delete c;
    return OKRT_OK;
}
const char* okrt_oracle_last_error(const okrt_ctx* c) { return c ? c->err.c_str() : "null context"; }

okrt_status okrt_oracle_set_table(okrt_ctx* c, okrt_table slot, const void* data, uint32_t elem, uint32_t count) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    switch (slot) {
        case OKRT_TABLE_TRI_POS: return setVec(c, c->triPos, data, elem, count);
        case OKRT_TABLE_TRI_INFO: return setVec(c, c->triInfo, data, elem, count);
        case OKRT_TABLE_BVH: return setVec(c, c->bvh, data, elem, count);
        case OKRT_TABLE_OCT: return setVec(c, c->oct, data, elem, count);
        case OKRT_TABLE_SPHERE: return setVec(c, c->spheres, data, elem, count);
        case OKRT_TABLE_MESH: return setVec(c, c->meshes, data, elem, count);
        case OKRT_TABLE_DIR_LIGHT: return setVec(c, c->dirLights, data, elem, count);
        case OKRT_TABLE_POINT_LIGHT: return setVec(c, c->pointLights, data, elem, count);
        case OKRT_TABLE_SPOT_LIGHT: return setVec(c, c->spotLights, data, elem, count);
        default: return fail(c, OKRT_ERR_INVALID_ARG, "unknown table slot");
    }
}
okrt_status okrt_oracle_set_textures(okrt_ctx* c, const uint8_t* rgba8, uint32_t count, uint32_t w, uint32_t h) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    if (!rgba8 || !count || !w || !h) {
        c->textures.clear();
        c->texCount = c->texW = c->texH = 0;
        return OKRT_OK;
    }
    c->textures.assign(rgba8, rgba8 + (size_t)count * w * h * 4);
    c->texCount = count; c->texW = w; c->texH = h;
    return OKRT_OK;
}
okrt_status okrt_oracle_set_envmap(okrt_ctx* c, const uint8_t* faces, uint32_t face) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    if (!faces || !face) {
        c->env.clear();
        c->envFace = 0;
        return OKRT_OK;
    }
    c->env.assign(faces, faces + (size_t)6 * face * face * 4);
    c->envFace = face;
    return OKRT_OK;
}
// Okay::scaleTexture (DX11.cpp:232-268, Scale/ScalePS.hlsl): one bilinear tap at each output
// texel centre, default sampler after ClearState (LINEAR, CLAMP), RGBA8 UNORM out.
okrt_status okrt_oracle_scale_texture_rgba8(okrt_ctx* c, const uint8_t* src, uint32_t sw, uint32_t sh, uint8_t* dst,
                                            uint32_t dw, uint32_t dh) {
    if (!src || !dst || !sw || !sh || !dw || !dh) return fail(c, OKRT_ERR_INVALID_ARG, "scale: bad argument");
    for (uint32_t oy = 0; oy < dh; ++oy)
        for (uint32_t ox = 0; ox < dw; ++ox) {
            float u = ((float)ox + 0.5f) / (float)dw, v = ((float)oy + 0.5f) / (float)dh;
            float x = fma_(u, (float)sw, -0.5f), y = fma_(v, (float)sh, -0.5f);
            float fx0 = floorf(x), fy0 = floorf(y);
            float fx = x - fx0, fy = y - fy0;
            int x0 = std::clamp((int)fx0, 0, (int)sw - 1), x1 = std::clamp((int)fx0 + 1, 0, (int)sw - 1);
            int y0 = std::clamp((int)fy0, 0, (int)sh - 1), y1 = std::clamp((int)fy0 + 1, 0, (int)sh - 1);
            for (int ch = 0; ch < 4; ++ch) {
                float t00 = src[((size_t)y0 * sw + x0) * 4 + ch] / 255.f, t10 = src[((size_t)y0 * sw + x1) * 4 + ch] / 255.f;
                float t01 = src[((size_t)y1 * sw + x0) * 4 + ch] / 255.f, t11 = src[((size_t)y1 * sw + x1) * 4 + ch] / 255.f;
                dst[((size_t)oy * dw + ox) * 4 + ch] = toUnorm8(lerp(lerp(t00, t10, fx), lerp(t01, t11, fx), fy));
            }
        }
    return OKRT_OK;
}

okrt_status okrt_oracle_resize(okrt_ctx* c, uint32_t w, uint32_t h) {
    if (!c || !w || !h) return fail(c, OKRT_ERR_INVALID_ARG, "resize: zero dimension");
    c->W = w; c->H = h;
    c->stripeRows = h; c->stripeIndex = 0; c->stripeStride = 1;
    buildRowMap(*c);
    return OKRT_OK;
}
okrt_status okrt_oracle_set_tile(okrt_ctx* c, uint32_t stripeRows, uint32_t stripeIndex, uint32_t stripeStride) {
    if (!c || !c->W) return fail(c, OKRT_ERR_STATE, "set_tile before resize");
    if (!stripeRows || !stripeStride || stripeIndex >= stripeStride) return fail(c, OKRT_ERR_INVALID_ARG, "set_tile: bad stripe");
    c->stripeRows = stripeRows; c->stripeIndex = stripeIndex; c->stripeStride = stripeStride;
    buildRowMap(*c);
    return OKRT_OK;
}
okrt_status okrt_oracle_tile_rows(const okrt_ctx* c, uint32_t* out) {
    if (!c || !out) return OKRT_ERR_INVALID_ARG;
    *out = (uint32_t)c->rowMap.size();
    return OKRT_OK;
}
okrt_status okrt_oracle_reset_accumulation(okrt_ctx* c) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    std::fill(c->accum.begin(), c->accum.end(), 0.f);
    return OKRT_OK;
}
okrt_status okrt_oracle_set_option(okrt_ctx* c, okrt_option opt, int64_t value) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    switch (opt) {
        case OKRT_OPT_TRAVERSAL: return OKRT_OK;  // the oracle only has the reference order
        case OKRT_OPT_MAX_BOUNCES:
            if (value < 0 || value > 64) return fail(c, OKRT_ERR_INVALID_ARG, "max_bounces out of range");
            c->maxBounces = (int)value;
            return OKRT_OK;
        case OKRT_OPT_COUNTERS: c->countersOn = value != 0; return OKRT_OK;
        case OKRT_OPT_TIMING: return OKRT_OK;
        case OKRT_OPT_BUILDER: return OKRT_OK;  // no derived structure here
        case OKRT_OPT_DEFER: return OKRT_OK;    // every frame is rendered inside its call
        case OKRT_OPT_QUEUE_MEMORY_MB: return OKRT_OK;
        default: return fail(c, OKRT_ERR_INVALID_ARG, "unknown option");
    }
}
// Oracle-only: the two sensitivity models of the unpinned D3D11 sampler (see okrt_ctx). weight_fraction_bits = 0 and seamless_cube = 0
// restore the oracle's defaults (exact fp32 weights, clamp inside the face).
okrt_status okrt_oracle_set_sampler_model(okrt_ctx* c, int weight_fraction_bits, int seamless_cube) {
    if (!c || weight_fraction_bits < 0 || weight_fraction_bits > 23) return OKRT_ERR_INVALID_ARG;
    c->samplerWeightBits = weight_fraction_bits;
    c->seamlessCube = seamless_cube != 0;
    return OKRT_OK;
}
okrt_status okrt_oracle_set_threads(okrt_ctx* c, int threads) {
    if (!c || threads < 0) return OKRT_ERR_INVALID_ARG;
    c->threads = threads;
    return OKRT_OK;
}

// Renders local rows [row0, row1) only (bounded samples for the CPU-baseline timing);
// okrt_oracle_render renders all owned rows.
okrt_status okrt_oracle_render_rows(okrt_ctx* c, const okrt_render_data* rd, uint32_t row0, uint32_t row1) {
    if (!c || !rd) return OKRT_ERR_INVALID_ARG;
    if (!c->W) return fail(c, OKRT_ERR_STATE, "render before resize");
    if (rd->textureDims[0] != c->W || rd->textureDims[1] != c->H) return fail(c, OKRT_ERR_INVALID_ARG, "textureDims != resize dims");
    if (rd->numSpheres > c->spheres.size() || rd->numDirLights > c->dirLights.size() ||
        rd->numPointLights > c->pointLights.size() || rd->numSpotLights > c->spotLights.size())
        return fail(c, OKRT_ERR_RANGE, "render data counts exceed the uploaded tables");
    row1 = std::min<uint32_t>(row1, (uint32_t)c->rowMap.size());
    if (row0 >= row1) return OKRT_OK;
    int nt = c->threads > 0 ? c->threads : (int)std::thread::hardware_concurrency();
    nt = std::max(1, std::min<int>(nt, (int)(row1 - row0)));
    std::atomic<uint32_t> next{row0};
    std::vector<Counters> partial(nt);
    auto worker = [&](int tid) {
        Counters cnt;
        for (;;) {
            uint32_t l = next.fetch_add(1);
            if (l >= row1) break;
            uint32_t y = c->rowMap[l];
            for (uint32_t x = 0; x < c->W; ++x) {
                uint32_t bb = 0, tri = 0;
                V3 light = tracePixel(*c, *rd, x, y, bb, tri, cnt);
                size_t o = ((size_t)l * c->W + x);
                c->bbImage[o] = bb;
                c->triImage[o] = tri;
                float* acc = &c->accum[o * 4];
                uint8_t* res = &c->result[o * 4];
                if (rd->accumulationEnabled == 1) {  // :526-530
                    acc[0] += light.x; acc[1] += light.y; acc[2] += light.z; acc[3] += 1.f;
                    float n = (float)rd->numAccumulationFrames;
                    for (int k = 0; k < 4; ++k) res[k] = toUnorm8(acc[k] / n);
                } else {  // :531-534
                    res[0] = toUnorm8(light.x); res[1] = toUnorm8(light.y); res[2] = toUnorm8(light.z); res[3] = 255;
                }
            }
        }
        partial[tid] = cnt;
    };
    if (nt == 1) {
        worker(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(worker, t);
        for (auto& t : th) t.join();
    }
    if (c->countersOn)
        for (auto& p : partial) c->counters.add(p);
    return OKRT_OK;
}
okrt_status okrt_oracle_render(okrt_ctx* c, const okrt_render_data* rd) {
    return okrt_oracle_render_rows(c, rd, 0, ~0u);
}
okrt_status okrt_oracle_read_result_rgba8(okrt_ctx* c, uint8_t* dst) {
    if (!c || !dst) return OKRT_ERR_INVALID_ARG;
    memcpy(dst, c->result.data(), c->result.size());
    return OKRT_OK;
}
okrt_status okrt_oracle_read_accum_f32x4(okrt_ctx* c, float* dst) {
    if (!c || !dst) return OKRT_ERR_INVALID_ARG;
    memcpy(dst, c->accum.data(), c->accum.size() * sizeof(float));
    return OKRT_OK;
}
okrt_status okrt_oracle_read_counter_images(okrt_ctx* c, uint32_t* bb, uint32_t* tri) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    if (bb) memcpy(bb, c->bbImage.data(), c->bbImage.size() * 4);
    if (tri) memcpy(tri, c->triImage.data(), c->triImage.size() * 4);
    return OKRT_OK;
}
okrt_status okrt_oracle_get_counters(okrt_ctx* c, okrt_counters* out) {
    if (!c || !out) return OKRT_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    out->pixels = c->counters.pixels; out->segments = c->counters.segments; out->n_oct = c->counters.n_oct;
    out->n_mesh = c->counters.n_mesh; out->n_node = c->counters.n_node; out->n_tri = c->counters.n_tri;
    out->mesh_hits = c->counters.mesh_hits; out->sphere_hits = c->counters.sphere_hits;
    out->tex_taps = c->counters.tex_taps; out->env_taps = c->counters.env_taps;
    return OKRT_OK;
}
okrt_status okrt_oracle_reset_counters(okrt_ctx* c) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    c->counters = Counters();
    return OKRT_OK;
}

okrt_status okrt_oracle_synchronize(okrt_ctx*) { return OKRT_OK; }
// `frames` consecutive frames = `frames` single-frame passes with the host's per-frame pre-increment (RayTracer.cpp:489)
okrt_status okrt_oracle_render_frames(okrt_ctx* c, const okrt_render_data* rd, uint32_t frames) {
    if (!c || !rd) return OKRT_ERR_INVALID_ARG;
    okrt_render_data cur = *rd;
    for (uint32_t f = 0; f < frames; ++f) {
        okrt_status s = okrt_oracle_render(c, &cur);
        if (s != OKRT_OK) return s;
        if (cur.accumulationEnabled == 1) cur.numAccumulationFrames += 1;
    }
    return OKRT_OK;
}
static okrt_status oracle_create_dev(int, okrt_ctx** out) { return okrt_oracle_create(out); }
// The oracle behind the same entry-point table as the product, so tests can drive the C++ host
// layer (gpu-raytracer_amd/host) against it on a machine without a GPU.
const okrt_api* okrt_oracle_get_api(void) {
    static const okrt_api api = {OKRT_ABI_VERSION, oracle_create_dev, okrt_oracle_destroy, okrt_oracle_last_error,
                                 okrt_oracle_set_table, okrt_oracle_set_textures, okrt_oracle_set_envmap, okrt_oracle_scale_texture_rgba8,
                                 okrt_oracle_resize, okrt_oracle_set_tile, okrt_oracle_tile_rows, okrt_oracle_reset_accumulation,
                                 okrt_oracle_render, okrt_oracle_synchronize, okrt_oracle_read_result_rgba8, okrt_oracle_read_accum_f32x4,
                                 okrt_oracle_set_option, okrt_oracle_get_counters, okrt_oracle_reset_counters, okrt_oracle_render_frames};
    return &api;
}

// ---- known-answer probes: expose the leaf functions so tests/ can pin them one by one --------
uint32_t okrt_oracle_kat_pcg_hash(uint32_t* seed) { return pcg_hash(*seed); }
float okrt_oracle_kat_random_float(uint32_t* seed) { return randomFloat(*seed); }
void okrt_oracle_kat_random_vector(uint32_t* seed, float out[3]) {
    V3 v = getRandomVector(*seed);
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
}
void okrt_oracle_kat_point_in_circle(uint32_t* seed, float out[2]) { randomPointInCircle(*seed, out[0], out[1]); }
float okrt_oracle_kat_ray_sphere(const float o[3], const float d[3], const float pos[3], float radius) {
    Ray r{v3(o), v3(d)};
    return RayAndSphere(r, v3(pos), radius);
}
float okrt_oracle_kat_ray_triangle(const float o[3], const float d[3], const float tri[9], float uv[2]) {
    Ray r{v3(o), v3(d)};
    uv[0] = uv[1] = 0.f;
    return RayAndTriangle(r, v3(tri), v3(tri + 3), v3(tri + 6), uv[0], uv[1]);
}
float okrt_oracle_kat_ray_aabb(const float o[3], const float d[3], const float bmin[3], const float bmax[3]) {
    Ray r{v3(o), v3(d)};
    return RayAndAABBDist(r, bmin, bmax);
}
float okrt_oracle_kat_reflectance(float cosine, float idx) { return reflectance(cosine, idx); }
void okrt_oracle_kat_refract(const float d[3], const float n[3], float ratio, float out[3]) {
    V3 r = refract_(v3(d), v3(n), ratio);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void okrt_oracle_kat_transparency_bounce(const float d[3], const float n[3], float ior, uint32_t* seed, float out[3]) {
    V3 r = findTransparencyBounce(v3(d), v3(n), ior, *seed);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void okrt_oracle_kat_reflect_direction(const float d[3], const float n[3], float rough, uint32_t* seed, float out[3]) {
    V3 r = findReflectDirection(v3(d), v3(n), rough, *seed);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void okrt_oracle_kat_sample_texture(okrt_ctx* c, uint32_t idx, float u, float v, float out[3]) {
    Counters cnt;
    V3 r = sampleTexture(*c, idx, u, v, cnt);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void okrt_oracle_kat_sample_env(okrt_ctx* c, const float d[3], float out[3]) {
    Counters cnt;
    V3 r = getEnvironmentLight(*c, v3(d), cnt);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
// closest hit of one ray through the uploaded tables: out = {distance, bu, bv}, ids = {hitIdx, hitType, triIdx}
void okrt_oracle_kat_closest_hit(okrt_ctx* c, const okrt_render_data* rd, const float o[3], const float d[3], float out[3], uint32_t ids[3]) {
    // findClosestHit returns the shaded payload; the raw hit is re-derived here with the same loop order
    Ray ray{v3(o), v3(d)};
    uint32_t bb = 0, tri = 0;
    Counters cnt;
    Payload p = findClosestHit(*c, *rd, ray, bb, tri, cnt);
    out[0] = p.distance;
    out[1] = p.worldPosition.x;
    out[2] = p.worldNormal.x;
    ids[0] = p.hit ? 1u : 0u;
    ids[1] = bb;
    ids[2] = tri;
}
void okrt_oracle_kat_primary_ray(const okrt_render_data* rd, uint32_t px, uint32_t py, float origin[3], float dir[3], uint32_t* seedOut) {
    // createRay + applyDOF only
    uint32_t seed = px + (py + 74813u) * rd->textureDims[0] * (rd->numAccumulationFrames + 1u);
    float posx = (float)px, posy = (float)(rd->textureDims[1] - py);
    posx += randomFloat(seed) - 0.5f;
    posy += randomFloat(seed) - 0.5f;
    posx /= (float)rd->textureDims[0];
    posy /= (float)rd->textureDims[1];
    posx = fma_(posx, 2.f, -1.f);
    posy = fma_(posy, 2.f, -1.f);
    float target[4];
    mat_point4(rd->cameraInverseProjectionMatrix, v3(posx, posy, rd->cameraNearZ), target);
    V3 o = v3(rd->cameraPosition);
    V3 tdir = normalize(v3(target[0] / target[2], target[1] / target[2], target[2] / target[2]));
    V3 d = mat_dir(rd->cameraInverseViewMatrix, tdir);
    float jx, jy;
    randomPointInCircle(seed, jx, jy);
    jx *= rd->dofStrength;
    jy *= rd->dofStrength;
    V3 rayOffset = fma3(v3(rd->cameraUpDir), jy, v3(rd->cameraRightDir) * jx);
    V3 focusPoint = fma3(d, rd->dofDistance + rd->cameraNearZ, o);
    o = o + rayOffset;
    d = normalize(focusPoint - o);
    origin[0] = o.x; origin[1] = o.y; origin[2] = o.z;
    dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
    *seedOut = seed;
}

}  // extern "C"
