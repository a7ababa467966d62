// okrt_mgpu.hip — the multi-GPU exchange of the tile split (include/okrt.h, "multi-GPU"): gather of the per-rank
// accumulation tiles to one root GPU with RCCL point-to-point calls over xGMI, and the de-interleave kernel that turns
// the gathered tiles into the full frame. The reference is single-GPU (one immediate context, DX11.cpp:12-19); this is
// the one exchange step the image-space split needs (pixels are independent: RaytracerCS.hlsl:430,528,536).
//
// RCCL is bound at run time (dlopen of librccl.so.1): libokrt.so has no link-time dependency on it, and a process that
// already carries an RCCL (PyTorch) is not handed a second copy unless it asks for a communicator here.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "okrt_kernels.h"

using namespace okrt;

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            r.error = std::string("cannot load librccl.so.1: ") + dlerror();
            return;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(r.lib, n);
            if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + n;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return r;
}

thread_local std::string g_commCreateError;

// tiles[r] at r * strideRows rows -> the full frame. One thread per pixel of the frame.
__global__ __launch_bounds__(256) void deinterleaveKernel(const float4* __restrict__ tiles, float4* __restrict__ full, uint32_t W, uint32_t H,
                                                         uint32_t nranks, uint32_t stripeRows, uint32_t strideRows) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)W * H) return;
    const uint32_t y = (uint32_t)(i / W), x = (uint32_t)(i - (size_t)y * W);
    const uint32_t stripe = y / stripeRows, rank = stripe % nranks;
    const uint32_t local = (stripe / nranks) * stripeRows + (y - stripe * stripeRows);
    full[i] = tiles[((size_t)rank * strideRows + local) * W + x];
}

uint32_t rowsOf(uint32_t H, uint32_t stripeRows, uint32_t idx, uint32_t stride) {
    uint32_t rows = 0;
    for (uint32_t s = idx; (uint64_t)s * stripeRows < H; s += stride) rows += (H - s * stripeRows) < stripeRows ? (H - s * stripeRows) : stripeRows;
    return rows;
}

}  // namespace

struct okrt_comm {
    ncclComm_t comm = nullptr;  // null for a communicator of one rank
    uint32_t rank = 0, nranks = 1;
    int device = 0;
    float4* staging = nullptr;  // root: nranks tiles of maxRows rows
    size_t stagingBytes = 0;
    float4* full = nullptr;     // root: library-owned full frame
    size_t fullBytes = 0;
    std::string err;
};

namespace {

okrt_status commFail(okrt_comm* c, okrt_status s, const std::string& msg) {
    if (c) c->err = msg;
    else g_commCreateError = msg;
    return s;
}

bool ncclOk(okrt_comm* c, ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return true;
    commFail(c, OKRT_ERR_DEVICE, std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error"));
    return false;
}

// Validates that the context's tile is the rank's share of an nranks-way split; returns its geometry.
okrt_status tileOf(okrt_ctx* ctx, okrt_comm* comm, CtxFrame* f) {
    okrt_status s = ctxFrame(ctx, f);
    if (s != OKRT_OK) return s;
    if (f->tile.stripeStride != comm->nranks || f->tile.stripeIndex != comm->rank)
        return ctxFail(ctx, OKRT_ERR_STATE, "mgpu gather: the context's tile is not okrt_set_tile(stripe_rows, rank, nranks) of this communicator");
    if (f->device != comm->device) return ctxFail(ctx, OKRT_ERR_STATE, "mgpu gather: communicator and context live on different devices");
    return OKRT_OK;
}

okrt_status ensureRootBuffers(okrt_ctx* ctx, okrt_comm* comm, const CtxFrame& f, bool needFull) {
    const uint32_t maxRows = rowsOf(f.tile.H, f.tile.stripeRows, 0, comm->nranks);  // rank 0 owns the most rows
    const size_t stagingBytes = (size_t)comm->nranks * maxRows * f.tile.W * sizeof(float4);
    const size_t fullBytes = (size_t)f.tile.H * f.tile.W * sizeof(float4);
    if (comm->stagingBytes < stagingBytes) {
        if (comm->staging) hipFree(comm->staging);
        comm->staging = nullptr;
        comm->stagingBytes = 0;
        if (hipMalloc((void**)&comm->staging, stagingBytes) != hipSuccess) return ctxFail(ctx, OKRT_ERR_OOM, "mgpu gather: staging allocation failed");
        comm->stagingBytes = stagingBytes;
    }
    if (needFull && comm->fullBytes < fullBytes) {
        if (comm->full) hipFree(comm->full);
        comm->full = nullptr;
        comm->fullBytes = 0;
        if (hipMalloc((void**)&comm->full, fullBytes) != hipSuccess) return ctxFail(ctx, OKRT_ERR_OOM, "mgpu gather: frame allocation failed");
        comm->fullBytes = fullBytes;
    }
    return OKRT_OK;
}

// The sends / receives of one rank (inside the caller's RCCL group) and, on the root, the copy of its own tile.
okrt_status postRank(okrt_ctx* ctx, okrt_comm* comm, const CtxFrame& f, uint32_t root) {
    const uint32_t maxRows = rowsOf(f.tile.H, f.tile.stripeRows, 0, comm->nranks);
    const size_t rowFloats = (size_t)f.tile.W * 4;
    if (comm->rank == root) {
        if (hipMemcpyAsync(comm->staging + (size_t)root * maxRows * f.tile.W, f.accum, (size_t)f.tile.rows * rowFloats * sizeof(float), hipMemcpyDeviceToDevice,
                           f.stream) != hipSuccess)
            return ctxFail(ctx, OKRT_ERR_DEVICE, "mgpu gather: copy of the root's own tile failed");
        for (uint32_t r = 0; r < comm->nranks; ++r) {
            if (r == root) continue;
            const uint32_t rows = rowsOf(f.tile.H, f.tile.stripeRows, r, comm->nranks);
            if (rows && !ncclOk(comm, rccl().Recv(comm->staging + (size_t)r * maxRows * f.tile.W, rows * rowFloats, ncclFloat, (int)r, comm->comm, f.stream), "ncclRecv"))
                return ctxFail(ctx, OKRT_ERR_DEVICE, comm->err);
        }
    } else if (f.tile.rows) {
        if (!ncclOk(comm, rccl().Send(f.accum, f.tile.rows * rowFloats, ncclFloat, (int)root, comm->comm, f.stream), "ncclSend"))
            return ctxFail(ctx, OKRT_ERR_DEVICE, comm->err);
    }
    return OKRT_OK;
}

okrt_status finishRoot(okrt_ctx* ctx, okrt_comm* comm, const CtxFrame& f, void* fullFrameDev) {
    const uint32_t maxRows = rowsOf(f.tile.H, f.tile.stripeRows, 0, comm->nranks);
    float4* dst = fullFrameDev ? (float4*)fullFrameDev : comm->full;
    const size_t px = (size_t)f.tile.W * f.tile.H;
    hipLaunchKernelGGL(deinterleaveKernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, f.stream, comm->staging, dst, f.tile.W, f.tile.H, comm->nranks,
                       f.tile.stripeRows, maxRows);
    if (hipGetLastError() != hipSuccess) return ctxFail(ctx, OKRT_ERR_DEVICE, "mgpu gather: de-interleave launch failed");
    return OKRT_OK;
}

}  // namespace

extern "C" {

const char* okrt_comm_last_error(const okrt_comm* comm) { return comm ? comm->err.c_str() : g_commCreateError.c_str(); }

okrt_status okrt_comm_unique_id(uint8_t out_id[OKRT_COMM_ID_BYTES]) {
    static_assert(OKRT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!out_id) return OKRT_ERR_INVALID_ARG;
    Rccl& r = rccl();
    if (!r.error.empty()) return commFail(nullptr, OKRT_ERR_DEVICE, r.error);
    ncclUniqueId id;
    if (!ncclOk(nullptr, r.GetUniqueId(&id), "ncclGetUniqueId")) return OKRT_ERR_DEVICE;
    std::memcpy(out_id, id.internal, OKRT_COMM_ID_BYTES);
    return OKRT_OK;
}

okrt_status okrt_comm_create(okrt_ctx* ctx, const uint8_t id[OKRT_COMM_ID_BYTES], uint32_t rank, uint32_t nranks, okrt_comm** out) {
    if (!ctx || !out || !nranks || rank >= nranks || (nranks > 1 && !id)) return commFail(nullptr, OKRT_ERR_INVALID_ARG, "comm_create: bad argument");
    *out = nullptr;
    CtxFrame f;
    okrt_status s = ctxFrame(ctx, &f);
    if (s != OKRT_OK) return commFail(nullptr, s, "comm_create: the context needs a frame size first (okrt_resize)");
    okrt_comm* c = new okrt_comm();
    c->rank = rank;
    c->nranks = nranks;
    c->device = f.device;
    if (nranks > 1) {
        Rccl& r = rccl();
        if (!r.error.empty()) {
            delete c;
            return commFail(nullptr, OKRT_ERR_DEVICE, r.error);
        }
        ncclUniqueId uid;
        std::memcpy(uid.internal, id, OKRT_COMM_ID_BYTES);
        if (hipSetDevice(f.device) != hipSuccess || !ncclOk(c, r.CommInitRank(&c->comm, (int)nranks, uid, (int)rank), "ncclCommInitRank")) {
            g_commCreateError = c->err.empty() ? "hipSetDevice failed" : c->err;
            delete c;
            return OKRT_ERR_DEVICE;
        }
    }
    *out = c;
    return OKRT_OK;
}

okrt_status okrt_comm_create_all(okrt_ctx* const* ctxs, uint32_t n, okrt_comm** out) {
    if (!ctxs || !out || !n) return commFail(nullptr, OKRT_ERR_INVALID_ARG, "comm_create_all: bad argument");
    std::vector<int> devs(n);
    for (uint32_t r = 0; r < n; ++r) {
        out[r] = nullptr;
        CtxFrame f;
        okrt_status s = ctxs[r] ? ctxFrame(ctxs[r], &f) : OKRT_ERR_INVALID_ARG;
        if (s != OKRT_OK) return commFail(nullptr, s, "comm_create_all: context " + std::to_string(r) + " is missing or has no frame size yet");
        devs[r] = f.device;
        for (uint32_t q = 0; q < r; ++q)
            if (devs[q] == devs[r]) return commFail(nullptr, OKRT_ERR_INVALID_ARG, "comm_create_all: two contexts share device " + std::to_string(devs[r]));
    }
    std::vector<ncclComm_t> comms(n, nullptr);
    if (n > 1) {
        Rccl& r = rccl();
        if (!r.error.empty()) return commFail(nullptr, OKRT_ERR_DEVICE, r.error);
        if (!ncclOk(nullptr, r.CommInitAll(comms.data(), (int)n, devs.data()), "ncclCommInitAll")) return OKRT_ERR_DEVICE;
    }
    for (uint32_t r = 0; r < n; ++r) {
        okrt_comm* c = new okrt_comm();
        c->comm = comms[r];
        c->rank = r;
        c->nranks = n;
        c->device = devs[r];
        out[r] = c;
    }
    return OKRT_OK;
}

okrt_status okrt_comm_destroy(okrt_comm* c) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    if (c->comm) rccl().CommDestroy(c->comm);
    if (c->staging) hipFree(c->staging);
    if (c->full) hipFree(c->full);
    delete c;
    return OKRT_OK;
}

okrt_status okrt_mgpu_gather_accum_all(okrt_ctx* const* ctxs, okrt_comm* const* comms, uint32_t n, uint32_t root, void* fullFrameDev) {
    if (!ctxs || !comms || !n) return OKRT_ERR_INVALID_ARG;
    std::vector<CtxFrame> fr(n);
    uint32_t rootSlot = n;
    for (uint32_t k = 0; k < n; ++k) {
        if (!ctxs[k] || !comms[k]) return OKRT_ERR_INVALID_ARG;
        if (root >= comms[k]->nranks) return ctxFail(ctxs[k], OKRT_ERR_INVALID_ARG, "mgpu gather: root out of range");
        okrt_status s = tileOf(ctxs[k], comms[k], &fr[k]);
        if (s != OKRT_OK) return s;
        if (fr[k].tile.W != fr[0].tile.W || fr[k].tile.H != fr[0].tile.H || fr[k].tile.stripeRows != fr[0].tile.stripeRows)
            return ctxFail(ctxs[k], OKRT_ERR_STATE, "mgpu gather: ranks disagree about the frame size or the stripe height");
        if (comms[k]->rank == root) rootSlot = k;
    }
    if (rootSlot < n) {
        if (hipSetDevice(fr[rootSlot].device) != hipSuccess) return ctxFail(ctxs[rootSlot], OKRT_ERR_DEVICE, "hipSetDevice");
        okrt_status s = ensureRootBuffers(ctxs[rootSlot], comms[rootSlot], fr[rootSlot], fullFrameDev == nullptr);
        if (s != OKRT_OK) return s;
    }
    const bool multi = comms[0]->nranks > 1;
    if (multi && !ncclOk(comms[0], rccl().GroupStart(), "ncclGroupStart")) return ctxFail(ctxs[0], OKRT_ERR_DEVICE, comms[0]->err);
    okrt_status st = OKRT_OK;
    for (uint32_t k = 0; k < n && st == OKRT_OK; ++k) {
        if (hipSetDevice(fr[k].device) != hipSuccess) st = ctxFail(ctxs[k], OKRT_ERR_DEVICE, "hipSetDevice");
        else st = postRank(ctxs[k], comms[k], fr[k], root);
    }
    if (multi && !ncclOk(comms[0], rccl().GroupEnd(), "ncclGroupEnd") && st == OKRT_OK) st = ctxFail(ctxs[0], OKRT_ERR_DEVICE, comms[0]->err);
    if (st != OKRT_OK) return st;
    if (rootSlot < n) {
        if (hipSetDevice(fr[rootSlot].device) != hipSuccess) return ctxFail(ctxs[rootSlot], OKRT_ERR_DEVICE, "hipSetDevice");
        return finishRoot(ctxs[rootSlot], comms[rootSlot], fr[rootSlot], fullFrameDev);
    }
    return OKRT_OK;
}

okrt_status okrt_mgpu_gather_accum(okrt_ctx* ctx, okrt_comm* comm, uint32_t root, void* fullFrameDev) {
    return okrt_mgpu_gather_accum_all(&ctx, &comm, 1, root, fullFrameDev);
}

okrt_status okrt_mgpu_read_gathered(okrt_ctx* ctx, okrt_comm* comm, float* dst) {
    if (!ctx || !comm || !dst) return OKRT_ERR_INVALID_ARG;
    CtxFrame f;
    okrt_status s = ctxFrame(ctx, &f);
    if (s != OKRT_OK) return s;
    const size_t bytes = (size_t)f.tile.W * f.tile.H * sizeof(float4);
    if (!comm->full || comm->fullBytes < bytes) return ctxFail(ctx, OKRT_ERR_STATE, "read_gathered: no library-owned frame on this rank (gather with full_frame_dev = NULL on the root first)");
    if (hipSetDevice(f.device) != hipSuccess || hipMemcpyAsync(dst, comm->full, bytes, hipMemcpyDeviceToHost, f.stream) != hipSuccess ||
        hipStreamSynchronize(f.stream) != hipSuccess)
        return ctxFail(ctx, OKRT_ERR_DEVICE, "read_gathered: copy failed");
    return OKRT_OK;
}

okrt_status okrt_mgpu_deinterleave(okrt_ctx* ctx, const void* tilesDev, uint32_t nranks, uint32_t stripeRows, uint32_t tileStrideRows, void* fullFrameDev) {
    if (!ctx || !tilesDev || !fullFrameDev || !nranks || !stripeRows) return OKRT_ERR_INVALID_ARG;
    CtxFrame f;
    okrt_status s = ctxFrame(ctx, &f);
    if (s != OKRT_OK) return s;
    if (tileStrideRows < rowsOf(f.tile.H, stripeRows, 0, nranks)) return ctxFail(ctx, OKRT_ERR_INVALID_ARG, "deinterleave: tile_stride_rows is smaller than rank 0's tile");
    if (hipSetDevice(f.device) != hipSuccess) return ctxFail(ctx, OKRT_ERR_DEVICE, "hipSetDevice");
    const size_t px = (size_t)f.tile.W * f.tile.H;
    hipLaunchKernelGGL(deinterleaveKernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, f.stream, (const float4*)tilesDev, (float4*)fullFrameDev, f.tile.W,
                       f.tile.H, nranks, stripeRows, tileStrideRows);
    if (hipGetLastError() != hipSuccess) return ctxFail(ctx, OKRT_ERR_DEVICE, "deinterleave: launch failed");
    return OKRT_OK;
}

}  // extern "C"
