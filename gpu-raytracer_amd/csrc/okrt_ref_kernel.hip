// okrt_ref_kernel.hip — the reference-ORDER path-tracing kernel for gfx950.
//
// One thread per pixel, the whole path in one launch, traversing the uploaded tables exactly as the
// reference's compute shader does (RaytracerCS.hlsl:218-334): spheres, then the octree stack, the
// instances of each octree node, each instance's BVH with the right child popped first, every
// triangle of a leaf. Its purpose is parity and accounting, not speed: visit order, tie-breaking
// and the bbCheckCount / triCheckCount debug counters are those of the reference, which makes its
// counters the layout-independent "algorithmic bytes" numerator of the roofline.
// The fast traversal (okrt_fast_trace.hip) is checked against this kernel and the CPU oracle.
//
// Launch shape: 256-thread blocks = 4 wave64, each wave an 8x8 pixel tile (the reference's 16x9
// groups are 2.25 waves; quirk Q16), blocks cover 16x16 pixels; every pixel of the owned rows is
// rendered (the reference truncates to multiples of 16x9; quirk Q15).
#include "okrt_kernels.h"

namespace okrt {

namespace {

constexpr uint32_t kOctStack = 20;  // RaytracerCS.hlsl:244
constexpr uint32_t kBvhStack = 50;  // RaytracerCS.hlsl:249

OKRT_DEV Hit closestHitReferenceOrder(const SceneView& s, const okrt_render_data& rd, const Ray& ray, uint32_t& bbCheckCount,
                                      uint32_t& triCheckCount, PathCounters& cnt) {
    Hit h;
    h.distance = kFltMax;
    h.hitIdx = kInvalid;
    h.hitType = 0;
    h.triIdx = kInvalid;
    h.bu = h.bv = 0.f;
    cnt.segments++;

    for (uint32_t i = 0; i < rd.numSpheres; i++) {  // :232-242
        float d = rayAndSphere(ray, ld3(s.spheres[i].position), s.spheres[i].radius);
        if (d > 0.f && d < h.distance) {
            h.distance = d;
            h.hitIdx = i;
            h.hitType = 0;
        }
    }

    uint32_t octStack[kOctStack];
    uint32_t octStackSize = 0;
    if (s.numOct) octStack[octStackSize++] = 0;
    uint32_t bvhStack[kBvhStack];
    const V3 invDirWorld = rcp3(ray.direction);

    while (octStackSize > 0) {  // :256-334
        const OctNodeRec& on = s.oct[octStack[--octStackSize]];
        bbCheckCount += 1;
        cnt.nOct++;
        if (rayAndAABBDist(ray.origin, invDirWorld, ld3(on.bbMin), ld3(on.bbMax)) >= h.distance) continue;

        const uint32_t mEnd = on.meshesEnd;
        for (uint32_t i = on.meshesStart; i < mEnd; i++) {  // :265-328
            bbCheckCount += 1;
            cnt.nMesh++;
            const MeshRec& mesh = s.meshes[i];
            uint32_t bvhStackSize = 1;
            bvhStack[0] = mesh.bvhNodeStartIdx;
            const V3 lo = matPoint(mesh.Minv, ray.origin);
            const V3 ld = normalize(matDir(mesh.Minv, ray.direction));
            const V3 linv = rcp3(ld);

            while (bvhStackSize > 0) {  // :278-327
                const BvhNodeRec& node = s.bvh[bvhStack[--bvhStackSize]];
                bbCheckCount += 1;
                cnt.nNode++;
                if (rayAndAABBDist(lo, linv, ld3(node.bbMin), ld3(node.bbMax)) >= h.distance) continue;
                const uint32_t first = node.firstChild;
                if (first != kInvalid) {
                    if (bvhStackSize + 2 <= kBvhStack) {
                        bvhStack[bvhStackSize++] = first;
                        bvhStack[bvhStackSize++] = first + 1;
                    }
                    continue;
                }
                const uint32_t tEnd = node.triEnd;
                for (uint32_t j = node.triStart; j < tEnd; j++) {  // :295-326
                    const float* p = s.triPos[j].p;
                    V3 p2 = ld3(p + 6);
                    float bu = 0.f, bv = 0.f;
                    triCheckCount += 1;
                    cnt.nTri++;
                    float d = rayAndTriangleEdges(lo, ld, p2, ld3(p) - p2, ld3(p + 3) - p2, bu, bv);
                    if (d <= 0.f) continue;
                    d = length(matDir(mesh.M, ld * d));  // world-space distance (:311-314)
                    if (d < h.distance) {
                        h.distance = d;
                        h.hitIdx = i;
                        h.hitType = 1;
                        h.triIdx = j;
                        h.bu = bu;
                        h.bv = bv;
                    }
                }
            }
        }
        for (uint32_t i = 0; i < on.numChildren; i++)  // :330-333
            if (octStackSize < kOctStack) octStack[octStackSize++] = on.firstChild + i;
    }
    return h;
}

__global__ __launch_bounds__(256) void renderReferenceOrder(SceneView s, okrt_render_data rd, TileView tile, float4* accum,
                                                           uint32_t* result, unsigned long long* counters) {
    const uint32_t t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const uint32_t px = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const uint32_t lrow = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool active = px < tile.W && lrow < tile.rows;

    PathCounters cnt = {};
    if (active) {
        const uint32_t py = tile.globalRow(lrow);
        uint32_t seed;
        Ray ray = primaryRay(rd, px, py, seed);
        V3 light = mk(0.f, 0.f, 0.f), contribution = mk(1.f, 1.f, 1.f);
        uint32_t bb = 0, tri = 0;
        for (uint32_t i = 0; i <= s.maxBounces; i++) {  // RaytracerCS.hlsl:444-514
            Hit h = closestHitReferenceOrder(s, rd, ray, bb, tri, cnt);
            Surface sf = shadeFetch(s, ray, h, cnt);
            addLights(s, rd, ray, sf, i, contribution, light, seed);
            if (!sf.hit) {
                light = fma3(sampleEnvironment(s, ray.direction, cnt), contribution, light);
                break;
            }
            scatter(sf, ray, contribution, light, seed);
        }
        writePixel(rd, light, bb, tri, accum, result, (size_t)lrow * tile.W + px);
    }

    if (counters) {  // wave-level sums, one atomic per counter per wave
        uint32_t vals[kNumCounters] = {active ? 1u : 0u, cnt.segments, cnt.nOct, cnt.nMesh, cnt.nNode, cnt.nTri,
                                       cnt.meshHits, cnt.sphereHits, cnt.texTaps, cnt.envTaps, cnt.fastNodes, cnt.fastTris};
#pragma unroll
        for (int k = 0; k < kNumCounters; ++k) {
            uint32_t v = vals[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && v) atomicAdd(&counters[k], (unsigned long long)v);
        }
    }
}

// Okay::scaleTexture (DX11.cpp:232-268 + Scale/ScalePS.hlsl): one bilinear tap at the output texel
// centre, default sampler after ClearState (LINEAR, CLAMP), RGBA8 UNORM render target.
__global__ __launch_bounds__(256) void scaleTextureKernel(const uint8_t* src, uint32_t sw, uint32_t sh, uint8_t* dst, uint32_t dw, uint32_t dh) {
    const uint32_t ox = blockIdx.x * 64 + (threadIdx.x & 63), oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= dw || oy >= dh) return;
    float u = ((float)ox + 0.5f) / (float)dw, v = ((float)oy + 0.5f) / (float)dh;
    float x = fma_(u, (float)sw, -0.5f), y = fma_(v, (float)sh, -0.5f);
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    int x0 = min(max((int)fx0, 0), (int)sw - 1), x1 = min(max((int)fx0 + 1, 0), (int)sw - 1);
    int y0 = min(max((int)fy0, 0), (int)sh - 1), y1 = min(max((int)fy0 + 1, 0), (int)sh - 1);
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);
    uint32_t c00 = s32[(size_t)y0 * sw + x0], c10 = s32[(size_t)y0 * sw + x1], c01 = s32[(size_t)y1 * sw + x0], c11 = s32[(size_t)y1 * sw + x1];
    uint32_t out = 0;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        float t00 = (float)((c00 >> (8 * ch)) & 255u) / 255.f, t10 = (float)((c10 >> (8 * ch)) & 255u) / 255.f;
        float t01 = (float)((c01 >> (8 * ch)) & 255u) / 255.f, t11 = (float)((c11 >> (8 * ch)) & 255u) / 255.f;
        out |= toUnorm8(lerp(lerp(t00, t10, fx), lerp(t01, t11, fx), fy)) << (8 * ch);
    }
    reinterpret_cast<uint32_t*>(dst)[(size_t)oy * dw + ox] = out;
}

}  // namespace

void launchReferenceOrder(const SceneView& s, const okrt_render_data& rd, const TileView& t, float4* accum, uint32_t* result,
                          unsigned long long* counters, hipStream_t stream) {
    dim3 grid((t.W + 15) / 16, (t.rows + 15) / 16);
    hipLaunchKernelGGL(renderReferenceOrder, grid, dim3(256), 0, stream, s, rd, t, accum, result, counters);
}

// Row-major RGBA8 slices -> 4x4-texel tiles (okrt_device.h: tiledOffset). One thread per texel of the padded grid.
__global__ __launch_bounds__(256) void tileTexturesKernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t slices, uint32_t W, uint32_t H) {
    const uint32_t tx = (W + 3) >> 2, ty = (H + 3) >> 2;
    const size_t perSlice = (size_t)tx * ty * 16, i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= perSlice * slices) return;
    const uint32_t slice = (uint32_t)(i / perSlice);
    const size_t r = i - (size_t)slice * perSlice;
    const uint32_t tile = (uint32_t)(r >> 4), in = (uint32_t)(r & 15);
    const uint32_t x = (tile % tx) * 4 + (in & 3), y = (tile / tx) * 4 + (in >> 2);
    dst[i] = (x < W && y < H) ? src[((size_t)slice * H + y) * W + x] : 0u;
}

void launchTileTextures(const void* src, void* dst, uint32_t slices, uint32_t W, uint32_t H, hipStream_t stream) {
    const size_t n = (size_t)((W + 3) >> 2) * ((H + 3) >> 2) * 16 * slices;
    if (n) hipLaunchKernelGGL(tileTexturesKernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const uint32_t*)src, (uint32_t*)dst, slices, W, H);
}

void launchScaleTexture(const uint8_t* src, uint32_t sw, uint32_t sh, uint8_t* dst, uint32_t dw, uint32_t dh, hipStream_t stream) {
    dim3 grid((dw + 63) / 64, (dh + 3) / 4);
    hipLaunchKernelGGL(scaleTextureKernel, grid, dim3(256), 0, stream, src, sw, sh, dst, dw, dh);
}

// ---- known-answer probes of the device leaf functions (tests/test_gpu_kats.py) ----------------------------------------
// One thread per record: `in` holds inFloats floats per record, `out` outFloats. Integer arguments and results travel
// as bit patterns. The same inputs go through the CPU checker's probes of the same functions; everything here that contains no
// transcendental must agree bit for bit.
__global__ __launch_bounds__(256) void katKernel(SceneView s, uint32_t op, uint32_t n, const float* __restrict__ in, uint32_t inFloats,
                                                 float* __restrict__ out, uint32_t outFloats) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* a = in + (size_t)i * inFloats;
    float* o = out + (size_t)i * outFloats;
    PathCounters cnt = {};
    switch (op) {
        case OKRT_KAT_PCG_HASH: {  // seed -> hash (== new seed)
            uint32_t seed = __float_as_uint(a[0]);
            o[0] = __uint_as_float(pcgHash(seed));
            o[1] = __uint_as_float(seed);
            break;
        }
        case OKRT_KAT_RANDOM_FLOAT: {
            uint32_t seed = __float_as_uint(a[0]);
            o[0] = randomFloat(seed);
            o[1] = __uint_as_float(seed);
            break;
        }
        case OKRT_KAT_RAY_SPHERE: {  // o[3] d[3] pos[3] radius -> t
            Ray r;
            r.origin = ld3(a);
            r.direction = ld3(a + 3);
            o[0] = rayAndSphere(r, ld3(a + 6), a[9]);
            break;
        }
        case OKRT_KAT_RAY_TRIANGLE: {  // o[3] d[3] p0 p1 p2 -> t u v
            const V3 p0 = ld3(a + 6), p1 = ld3(a + 9), p2 = ld3(a + 12);
            float u = 0.f, v = 0.f;
            o[0] = rayAndTriangleEdges(ld3(a), ld3(a + 3), p2, p0 - p2, p1 - p2, u, v);
            o[1] = u;
            o[2] = v;
            break;
        }
        case OKRT_KAT_RAY_AABB: {  // o[3] d[3] min[3] max[3] -> distance
            o[0] = rayAndAABBDist(ld3(a), rcp3(ld3(a + 3)), ld3(a + 6), ld3(a + 9));
            break;
        }
        case OKRT_KAT_REFRACT: {  // d[3] n[3] ratio -> out[3]
            const V3 r = refractUser(ld3(a), ld3(a + 3), a[6]);
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            break;
        }
        case OKRT_KAT_SAMPLE_TEXTURE: {  // idx(bits) u v -> rgb
            const V3 r = sampleTexture(s, __float_as_uint(a[0]), a[1], a[2], cnt);
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            break;
        }
        case OKRT_KAT_SAMPLE_ENV: {  // d[3] -> rgb
            const V3 r = sampleEnvironment(s, ld3(a), cnt);
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            break;
        }
        case OKRT_KAT_RANDOM_VECTOR: {  // seed -> v[3], new seed   (Box-Muller: log, cos)
            uint32_t seed = __float_as_uint(a[0]);
            const V3 r = randomVector(seed);
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            o[3] = __uint_as_float(seed);
            break;
        }
        case OKRT_KAT_REFLECTANCE: {  // cosine, index -> Schlick with (1 + r0) (quirk Q5; pow = exp2(y log2 x))
            o[0] = reflectance(a[0], a[1]);
            break;
        }
        case OKRT_KAT_REFLECT_DIRECTION: {  // d[3] n[3] roughness seed -> out[3], new seed
            uint32_t seed = __float_as_uint(a[7]);
            const V3 r = findReflectDirection(ld3(a), ld3(a + 3), a[6], seed);
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            o[3] = __uint_as_float(seed);
            break;
        }
        case OKRT_KAT_TRANSPARENCY_BOUNCE: {  // d[3] n[3] ior seed -> out[3], new seed   (quirks Q6-Q8)
            uint32_t seed = __float_as_uint(a[7]);
            const V3 r = findTransparencyBounce(ld3(a), ld3(a + 3), a[6], seed);
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            o[3] = __uint_as_float(seed);
            break;
        }
        case OKRT_KAT_PRIMARY_RAY: {  // px(bits) py(bits) RenderData[60] -> origin[3], direction[3], seed after createRay + applyDOF
            okrt_render_data rd;
            float* rf = reinterpret_cast<float*>(&rd);
            for (int k = 0; k < 60; ++k) rf[k] = a[2 + k];
            uint32_t seed = 0;
            const Ray r = primaryRay(rd, __float_as_uint(a[0]), __float_as_uint(a[1]), seed);
            o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
            o[3] = r.direction.x; o[4] = r.direction.y; o[5] = r.direction.z;
            o[6] = __uint_as_float(seed);
            break;
        }
        case OKRT_KAT_CLOSEST_HIT: {  // o[3] d[3] numSpheres(bits) -> distance, hit?, bbCheckCount, triCheckCount, hitIdx, triIdx
            okrt_render_data rd = {};
            rd.numSpheres = __float_as_uint(a[6]);
            Ray r;
            r.origin = ld3(a);
            r.direction = ld3(a + 3);
            uint32_t bb = 0, tri = 0;
            const Hit h = closestHitReferenceOrder(s, rd, r, bb, tri, cnt);
            o[0] = h.distance;
            o[1] = __uint_as_float(h.hitIdx != kInvalid ? 1u : 0u);
            o[2] = __uint_as_float(bb);
            o[3] = __uint_as_float(tri);
            o[4] = __uint_as_float(h.hitIdx);
            o[5] = __uint_as_float(h.triIdx);
            break;
        }
        case OKRT_KAT_SEGMENT_RAY: {  // px(bits) py(bits) k(bits) RenderData[60] -> alive?, origin[3], direction[3], seed
            // the ray that enters path segment k of this pixel, following the reference-order path (k = 0: the camera ray)
            okrt_render_data rd;
            float* rf = reinterpret_cast<float*>(&rd);
            for (int j = 0; j < 60; ++j) rf[j] = a[3 + j];
            const uint32_t k = __float_as_uint(a[2]);
            uint32_t seed = 0;
            Ray r = primaryRay(rd, __float_as_uint(a[0]), __float_as_uint(a[1]), seed);
            V3 light = mk(0.f, 0.f, 0.f), contribution = mk(1.f, 1.f, 1.f);
            uint32_t bb = 0, tri = 0, alive = 1u;
            for (uint32_t i = 0; i < k && alive; i++) {  // RaytracerCS.hlsl:444-514
                Hit h = closestHitReferenceOrder(s, rd, r, bb, tri, cnt);
                Surface sf = shadeFetch(s, r, h, cnt);
                addLights(s, rd, r, sf, i, contribution, light, seed);
                if (!sf.hit) alive = 0u;
                else scatter(sf, r, contribution, light, seed);
            }
            o[0] = __uint_as_float(alive);
            o[1] = r.origin.x; o[2] = r.origin.y; o[3] = r.origin.z;
            o[4] = r.direction.x; o[5] = r.direction.y; o[6] = r.direction.z;
            o[7] = __uint_as_float(seed);
            break;
        }
        case OKRT_KAT_IEEE_SHORTCUTS: {  // first bit pattern, count -> patterns where ieeeRcp / ieeeSqrt differ from the compiler's 1/x / sqrtf
            uint32_t badRcp = 0, badSqrt = 0, firstRcp = 0, firstSqrt = 0;
            const uint32_t first = __float_as_uint(a[0]), count = __float_as_uint(a[1]);
            for (uint32_t j = 0; j < count; ++j) {
                const float x = __uint_as_float(first + j);
                const float q0 = 1.0f / x, q1 = ieeeRcp(x), r0 = sqrtf(x), r1 = ieeeSqrt(x);
                const bool sameQ = __float_as_uint(q0) == __float_as_uint(q1) || (q0 != q0 && q1 != q1);
                const bool sameR = __float_as_uint(r0) == __float_as_uint(r1) || (r0 != r0 && r1 != r1);
                if (!sameQ && !badRcp++) firstRcp = first + j;
                if (!sameR && !badSqrt++) firstSqrt = first + j;
            }
            o[0] = __uint_as_float(badRcp); o[1] = __uint_as_float(badSqrt);
            o[2] = __uint_as_float(firstRcp); o[3] = __uint_as_float(firstSqrt);
            break;
        }
        default: break;
    }
}

void launchKat(const SceneView& s, uint32_t op, uint32_t n, const float* in, uint32_t inFloats, float* out, uint32_t outFloats, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(katKernel, dim3((n + 255) / 256), dim3(256), 0, stream, s, op, n, in, inFloats, out, outFloats);
}

}  // namespace okrt
