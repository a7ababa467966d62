// okrt_kernels.h — launch interface between the context (okrt_context.hip) and the kernels.
#pragma once
#include <string>

#include "okrt_device.h"

namespace okrt {

// The rows this context owns (okrt_set_tile): local row l <-> global y.
struct TileView {
    uint32_t W, H;        // full frame
    uint32_t rows;        // owned rows
    uint32_t stripeRows, stripeIndex, stripeStride;
    __host__ __device__ uint32_t globalRow(uint32_t l) const {
        return ((l / stripeRows) * stripeStride + stripeIndex) * stripeRows + (l % stripeRows);
    }
};

constexpr int kNumCounters = 12;  // layout of okrt_counters

// Reference-order megakernel (okrt_ref_kernel.hip).
void launchReferenceOrder(const SceneView& s, const okrt_render_data& rd, const TileView& t, float4* accum, uint32_t* result,
                          unsigned long long* counters, hipStream_t stream);

// One bilinear tap per output texel (okrt_ref_kernel.hip).
void launchScaleTexture(const uint8_t* src, uint32_t sw, uint32_t sh, uint8_t* dst, uint32_t dw, uint32_t dh, hipStream_t stream);

// Row-major RGBA8 slices -> the tiled layout the samplers read (okrt_ref_kernel.hip).
void launchTileTextures(const void* src, void* dst, uint32_t slices, uint32_t W, uint32_t H, hipStream_t stream);

// Known-answer probes of the device leaf functions (okrt_ref_kernel.hip, okrt_debug_kat).
void launchKat(const SceneView& s, uint32_t op, uint32_t n, const float* in, uint32_t inFloats, float* out, uint32_t outFloats, hipStream_t stream);

// What okrt_mgpu.hip needs of a context (okrt_context.hip owns the struct): pending frames executed, then the frame view.
struct CtxFrame {
    int device;
    hipStream_t stream;
    float4* accum;   // owned rows, packed
    TileView tile;
};
okrt_status ctxFrame(okrt_ctx* c, CtxFrame* out);           // flushes deferred frames; OKRT_ERR_STATE before okrt_resize
okrt_status ctxFail(okrt_ctx* c, okrt_status s, const std::string& msg);

}  // namespace okrt
