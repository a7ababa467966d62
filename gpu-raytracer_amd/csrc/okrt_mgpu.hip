// okrt_mgpu.hip — the multi-GPU exchange of the tile split (include/okrt.h, "multi-GPU"): gather of the per-rank
// accumulation tiles to one root GPU with RCCL point-to-point calls over xGMI, and the de-interleave kernel that turns
// the gathered tiles into the full frame. The reference is single-GPU (one immediate context, DX11.cpp:12-19); this is
// the one exchange step the image-space split needs (pixels are independent: RaytracerCS.hlsl:430,528,536).
//
// RCCL is bound at run time (dlopen of librccl.so.1): libokrt.so has no link-time dependency on it, and a process that
// already carries an RCCL (PyTorch) is not handed a second copy unless it asks for a communicator here.
//
// Two transports sit under the same entry points:
//   rccl  one ncclSend per peer / one ncclRecv per peer on the root inside ONE group, on the contexts' streams (default);
//   copy  okrt_comm_create_all only (one process owns every context): the root's stream waits on an event of each peer's
//         stream and pulls the tile with hipMemcpyPeerAsync (SDMA over xGMI, or a plain device copy when two contexts
//         share a GPU). Chosen when two contexts share a device — RCCL refuses duplicate devices — which is how an
//         N-rank split is rehearsed on fewer GPUs, or with OKRT_MGPU_TRANSPORT=copy.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
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
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
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
        r.Reduce = (decltype(r.Reduce))sym("ncclReduce");
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

// Samples split (okrt_mgpu_reduce_accum with the copy transport): full += part, element by element.
__global__ __launch_bounds__(256) void addFramesKernel(float4* __restrict__ full, const float4* __restrict__ part, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 a = full[i];
    const float4 b = part[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    full[i] = a;
}

uint32_t rowsOf(uint32_t H, uint32_t stripeRows, uint32_t idx, uint32_t stride) {
    uint32_t rows = 0;
    for (uint32_t s = idx; (uint64_t)s * stripeRows < H; s += stride) rows += (H - s * stripeRows) < stripeRows ? (H - s * stripeRows) : stripeRows;
    return rows;
}

}  // namespace

struct okrt_comm {
    ncclComm_t comm = nullptr;  // null for a communicator of one rank and for the copy transport
    uint32_t rank = 0, nranks = 1;
    int device = 0;
    bool copyTransport = false;
    hipEvent_t ready = nullptr, done = nullptr;  // copy transport: "my frames are rendered" / "the root has my tile"
    hipEvent_t xBegin = nullptr, xEnd = nullptr;  // root: around the last gather / reduce on its stream (okrt_comm_exchange_ms)
    bool xRecorded = false;
    float4* staging = nullptr;  // root: nranks tiles of maxRows rows (gather) or one full frame (reduce, copy transport)
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

// Validates that the context's tile is the rank's share of an nranks-way split (or, for the samples split, the whole
// frame); returns its geometry.
okrt_status tileOf(okrt_ctx* ctx, okrt_comm* comm, CtxFrame* f, bool wholeFrame) {
    okrt_status s = ctxFrame(ctx, f);
    if (s != OKRT_OK) return s;
    if (wholeFrame) {
        if (f->tile.rows != f->tile.H) return ctxFail(ctx, OKRT_ERR_STATE, "mgpu reduce: the samples split needs the full-frame tile on every rank");
    } else if (f->tile.stripeStride != comm->nranks || f->tile.stripeIndex != comm->rank) {
        return ctxFail(ctx, OKRT_ERR_STATE, "mgpu gather: the context's tile is not okrt_set_tile(stripe_rows, rank, nranks) of this communicator");
    }
    if (f->device != comm->device) return ctxFail(ctx, OKRT_ERR_STATE, "mgpu gather: communicator and context live on different devices");
    return OKRT_OK;
}

okrt_status ensureRootBuffers(okrt_ctx* ctx, okrt_comm* comm, size_t stagingBytes, size_t fullBytes) {
    if (comm->stagingBytes < stagingBytes) {
        if (comm->staging) hipFree(comm->staging);
        comm->staging = nullptr;
        comm->stagingBytes = 0;
        if (hipMalloc((void**)&comm->staging, stagingBytes) != hipSuccess) {
            (void)hipGetLastError();
            return ctxFail(ctx, OKRT_ERR_OOM, "mgpu gather: staging allocation failed");
        }
        comm->stagingBytes = stagingBytes;
    }
    if (comm->fullBytes < fullBytes) {
        if (comm->full) hipFree(comm->full);
        comm->full = nullptr;
        comm->fullBytes = 0;
        if (hipMalloc((void**)&comm->full, fullBytes) != hipSuccess) {
            (void)hipGetLastError();
            return ctxFail(ctx, OKRT_ERR_OOM, "mgpu gather: frame allocation failed");
        }
        comm->fullBytes = fullBytes;
    }
    return OKRT_OK;
}

// What the two collectives share: every context and communicator checked BEFORE anything is posted (nothing that can
// fail for a reason other than the transport itself is left inside the RCCL group), the root found, its buffers sized.
struct Plan {
    std::vector<CtxFrame> fr;
    uint32_t rootSlot, maxRows;
    bool copy;
};

okrt_status planCollective(okrt_ctx* const* ctxs, okrt_comm* const* comms, uint32_t n, uint32_t root, bool wholeFrame, bool needFull, Plan* p) {
    if (!ctxs || !comms || !n) return OKRT_ERR_INVALID_ARG;
    for (uint32_t k = 0; k < n; ++k)
        if (!ctxs[k] || !comms[k]) return OKRT_ERR_INVALID_ARG;
    const uint32_t nranks = comms[0]->nranks;
    // one thread drives either ONE rank of the communicator (one process per GPU) or ALL of them (one process, n GPUs):
    // with a subset the group would post sends without their receive and hang the peers
    if (n != 1 && n != nranks) return ctxFail(ctxs[0], OKRT_ERR_INVALID_ARG, "mgpu: pass one rank or all " + std::to_string(nranks) + " ranks of the communicator");
    p->fr.resize(n);
    p->rootSlot = n;
    p->copy = comms[0]->copyTransport;
    if (p->copy && n != nranks) return ctxFail(ctxs[0], OKRT_ERR_INVALID_ARG, "mgpu: the copy transport needs all ranks in one call (okrt_mgpu_*_all)");
    std::vector<char> seen(nranks, 0);
    for (uint32_t k = 0; k < n; ++k) {
        if (comms[k]->nranks != nranks || comms[k]->copyTransport != p->copy) return ctxFail(ctxs[k], OKRT_ERR_INVALID_ARG, "mgpu: communicators of different groups");
        if (root >= nranks) return ctxFail(ctxs[k], OKRT_ERR_INVALID_ARG, "mgpu gather: root out of range");
        if (seen[comms[k]->rank]++) return ctxFail(ctxs[k], OKRT_ERR_INVALID_ARG, "mgpu: rank " + std::to_string(comms[k]->rank) + " passed twice");
        okrt_status s = tileOf(ctxs[k], comms[k], &p->fr[k], wholeFrame);
        if (s != OKRT_OK) return s;
        if (p->fr[k].tile.W != p->fr[0].tile.W || p->fr[k].tile.H != p->fr[0].tile.H || (!wholeFrame && p->fr[k].tile.stripeRows != p->fr[0].tile.stripeRows))
            return ctxFail(ctxs[k], OKRT_ERR_STATE, "mgpu gather: ranks disagree about the frame size or the stripe height");
        if (comms[k]->rank == root) p->rootSlot = k;
    }
    const TileView& t = p->fr[0].tile;
    p->maxRows = wholeFrame ? t.H : rowsOf(t.H, t.stripeRows, 0, nranks);  // rank 0 owns the most rows
    if (p->rootSlot < n) {
        if (hipSetDevice(p->fr[p->rootSlot].device) != hipSuccess) return ctxFail(ctxs[p->rootSlot], OKRT_ERR_DEVICE, "hipSetDevice");
        const size_t frameBytes = (size_t)t.H * t.W * sizeof(float4);
        const size_t staging = wholeFrame ? (p->copy ? frameBytes : 0) : (size_t)nranks * p->maxRows * t.W * sizeof(float4);
        okrt_status s = ensureRootBuffers(ctxs[p->rootSlot], comms[p->rootSlot], staging, needFull ? frameBytes : 0);
        if (s != OKRT_OK) return s;
    }
    return OKRT_OK;
}

// copy transport: the root's stream pulls `bytes` from a peer's buffer once the peer's stream has reached this point.
okrt_status pullFromPeer(okrt_ctx* rootCtx, const CtxFrame& rootF, okrt_comm* peer, const CtxFrame& peerF, void* dst, const void* src, size_t bytes) {
    hipError_t e = hipSetDevice(peerF.device);
    if (e == hipSuccess) e = hipEventRecord(peer->ready, peerF.stream);
    if (e == hipSuccess) e = hipSetDevice(rootF.device);
    if (e == hipSuccess) e = hipStreamWaitEvent(rootF.stream, peer->ready, 0);
    if (e == hipSuccess)
        e = peerF.device == rootF.device ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, rootF.stream)
                                         : hipMemcpyPeerAsync(dst, rootF.device, src, peerF.device, bytes, rootF.stream);
    if (e != hipSuccess) return ctxFail(rootCtx, OKRT_ERR_DEVICE, std::string("mgpu copy transport: ") + hipGetErrorString(e));
    return OKRT_OK;
}
// ... and afterwards every peer's stream waits until the root holds its data (it may render into that buffer again).
okrt_status releasePeers(okrt_ctx* rootCtx, const Plan& p, okrt_comm* const* comms) {
    const CtxFrame& rf = p.fr[p.rootSlot];
    hipError_t e = hipSetDevice(rf.device);
    if (e == hipSuccess) e = hipEventRecord(comms[p.rootSlot]->done, rf.stream);
    for (uint32_t k = 0; k < p.fr.size() && e == hipSuccess; ++k) {
        if (k == p.rootSlot) continue;
        e = hipSetDevice(p.fr[k].device);
        if (e == hipSuccess) e = hipStreamWaitEvent(p.fr[k].stream, comms[p.rootSlot]->done, 0);
    }
    if (e != hipSuccess) return ctxFail(rootCtx, OKRT_ERR_DEVICE, std::string("mgpu copy transport: ") + hipGetErrorString(e));
    return OKRT_OK;
}

okrt_status deinterleaveOnRoot(okrt_ctx* ctx, okrt_comm* comm, const CtxFrame& f, uint32_t maxRows, void* fullFrameDev) {
    float4* dst = fullFrameDev ? (float4*)fullFrameDev : comm->full;
    const size_t px = (size_t)f.tile.W * f.tile.H;
    if (hipSetDevice(f.device) != hipSuccess) return ctxFail(ctx, OKRT_ERR_DEVICE, "hipSetDevice");
    hipLaunchKernelGGL(deinterleaveKernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, f.stream, comm->staging, dst, f.tile.W, f.tile.H, comm->nranks,
                       f.tile.stripeRows, maxRows);
    if (hipGetLastError() != hipSuccess) return ctxFail(ctx, OKRT_ERR_DEVICE, "mgpu gather: de-interleave launch failed");
    return OKRT_OK;
}

okrt_comm* newComm(uint32_t rank, uint32_t nranks, int device) {
    okrt_comm* c = new okrt_comm();
    c->rank = rank;
    c->nranks = nranks;
    c->device = device;
    return c;
}

}  // namespace

extern "C" {

const char* okrt_comm_last_error(const okrt_comm* comm) { return comm ? comm->err.c_str() : g_commCreateError.c_str(); }

const char* okrt_comm_transport(const okrt_comm* comm) {
    if (!comm) return "";
    return comm->nranks == 1 ? "single" : comm->copyTransport ? "copy" : "rccl";
}

// okrt_comm_exchange_ms: the root brackets an exchange with two events on its own stream — from "my frames are rendered" (the
// exchange is enqueued behind them) to "the full frame is assembled": the peers' remaining render time it has to wait for, the
// transfers and the de-interleave. Timing failures never fail the exchange.
static void exchangeBegin(okrt_comm* c, const CtxFrame& f) {
    c->xRecorded = false;
    if (hipSetDevice(f.device) != hipSuccess) return;
    if (!c->xBegin && (hipEventCreate(&c->xBegin) != hipSuccess || hipEventCreate(&c->xEnd) != hipSuccess)) {
        (void)hipGetLastError();
        return;
    }
    if (hipEventRecord(c->xBegin, f.stream) != hipSuccess) (void)hipGetLastError();
}
static void exchangeEnd(okrt_comm* c, const CtxFrame& f) {
    if (!c->xBegin || !c->xEnd || hipSetDevice(f.device) != hipSuccess) return;
    if (hipEventRecord(c->xEnd, f.stream) == hipSuccess) c->xRecorded = true;
    else (void)hipGetLastError();
}

okrt_status okrt_comm_exchange_ms(okrt_comm* c, float* out_ms) {
    if (!c || !out_ms) return OKRT_ERR_INVALID_ARG;
    *out_ms = 0.f;
    if (!c->xRecorded) return commFail(c, OKRT_ERR_STATE, "exchange_ms: this rank was not the root of a gather / reduce yet");
    if (hipSetDevice(c->device) != hipSuccess || hipEventSynchronize(c->xEnd) != hipSuccess || hipEventElapsedTime(out_ms, c->xBegin, c->xEnd) != hipSuccess) {
        (void)hipGetLastError();
        return commFail(c, OKRT_ERR_DEVICE, "exchange_ms: reading the events failed");
    }
    return OKRT_OK;
}

okrt_status okrt_device_count(int* out_count) {
    if (!out_count) return OKRT_ERR_INVALID_ARG;
    *out_count = 0;
    if (hipGetDeviceCount(out_count) != hipSuccess) {
        (void)hipGetLastError();
        *out_count = 0;
        return OKRT_ERR_DEVICE;
    }
    return OKRT_OK;
}

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
    okrt_comm* c = newComm(rank, nranks, f.device);
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
    bool shared = false;
    for (uint32_t r = 0; r < n; ++r) {
        out[r] = nullptr;
        CtxFrame f;
        okrt_status s = ctxs[r] ? ctxFrame(ctxs[r], &f) : OKRT_ERR_INVALID_ARG;
        if (s != OKRT_OK) return commFail(nullptr, s, "comm_create_all: context " + std::to_string(r) + " is missing or has no frame size yet");
        devs[r] = f.device;
        for (uint32_t q = 0; q < r; ++q) {
            if (ctxs[q] == ctxs[r]) return commFail(nullptr, OKRT_ERR_INVALID_ARG, "comm_create_all: context " + std::to_string(r) + " passed twice");
            shared = shared || devs[q] == devs[r];
        }
    }
    // RCCL cannot place two ranks on one device: contexts that share a GPU (an N-rank split rehearsed on fewer GPUs)
    // are joined by the copy transport instead; OKRT_MGPU_TRANSPORT=copy selects it for distinct devices too
    const char* env = getenv("OKRT_MGPU_TRANSPORT");
    const bool copy = n > 1 && (shared || (env && std::strcmp(env, "copy") == 0));
    std::vector<ncclComm_t> comms(n, nullptr);
    if (n > 1 && !copy) {
        Rccl& r = rccl();
        if (!r.error.empty()) return commFail(nullptr, OKRT_ERR_DEVICE, r.error);
        if (!ncclOk(nullptr, r.CommInitAll(comms.data(), (int)n, devs.data()), "ncclCommInitAll")) return OKRT_ERR_DEVICE;
    }
    for (uint32_t r = 0; r < n; ++r) {
        okrt_comm* c = newComm(r, n, devs[r]);
        c->comm = comms[r];
        c->copyTransport = copy;
        if (copy) {
            hipError_t e = hipSetDevice(devs[r]);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
            if (e != hipSuccess) {
                commFail(nullptr, OKRT_ERR_DEVICE, std::string("comm_create_all: ") + hipGetErrorString(e));
                delete c;
                for (uint32_t q = 0; q < r; ++q) {
                    okrt_comm_destroy(out[q]);
                    out[q] = nullptr;
                }
                return OKRT_ERR_DEVICE;
            }
        }
        out[r] = c;
    }
    return OKRT_OK;
}

okrt_status okrt_comm_destroy(okrt_comm* c) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    if (c->comm) rccl().CommDestroy(c->comm);
    if (c->ready) hipEventDestroy(c->ready);
    if (c->done) hipEventDestroy(c->done);
    if (c->xBegin) hipEventDestroy(c->xBegin);
    if (c->xEnd) hipEventDestroy(c->xEnd);
    if (c->staging) hipFree(c->staging);
    if (c->full) hipFree(c->full);
    delete c;
    return OKRT_OK;
}

okrt_status okrt_mgpu_gather_accum_all(okrt_ctx* const* ctxs, okrt_comm* const* comms, uint32_t n, uint32_t root, void* fullFrameDev) {
    Plan p;
    okrt_status st = planCollective(ctxs, comms, n, root, false, fullFrameDev == nullptr, &p);
    if (st != OKRT_OK) return st;
    const uint32_t nranks = comms[0]->nranks;
    const TileView& t0 = p.fr[0].tile;
    const size_t rowBytes = (size_t)t0.W * sizeof(float4);
    if (p.rootSlot < n) exchangeBegin(comms[p.rootSlot], p.fr[p.rootSlot]);
    if (p.rootSlot < n) {  // the root's own tile: a device copy, outside any group
        const CtxFrame& rf = p.fr[p.rootSlot];
        if (rf.tile.rows && (hipSetDevice(rf.device) != hipSuccess ||
                             hipMemcpyAsync(comms[p.rootSlot]->staging + (size_t)root * p.maxRows * t0.W, rf.accum, rf.tile.rows * rowBytes, hipMemcpyDeviceToDevice, rf.stream) != hipSuccess))
            return ctxFail(ctxs[p.rootSlot], OKRT_ERR_DEVICE, "mgpu gather: copy of the root's own tile failed");
    }
    if (nranks > 1 && p.copy) {
        for (uint32_t k = 0; k < n; ++k) {
            if (k == p.rootSlot || !p.fr[k].tile.rows) continue;
            st = pullFromPeer(ctxs[p.rootSlot], p.fr[p.rootSlot], comms[k], p.fr[k], comms[p.rootSlot]->staging + (size_t)comms[k]->rank * p.maxRows * t0.W, p.fr[k].accum,
                              p.fr[k].tile.rows * rowBytes);
            if (st != OKRT_OK) return st;
        }
    } else if (nranks > 1) {
        // ONE group around every send and receive this thread posts; only RCCL calls inside it
        if (!ncclOk(comms[0], rccl().GroupStart(), "ncclGroupStart")) return ctxFail(ctxs[0], OKRT_ERR_DEVICE, comms[0]->err);
        for (uint32_t k = 0; k < n && st == OKRT_OK; ++k) {
            okrt_comm* c = comms[k];
            const CtxFrame& f = p.fr[k];
            (void)hipSetDevice(f.device);  // (RCCL takes the device from the communicator; this keeps the thread's current device in step)
            if (k == p.rootSlot) {
                for (uint32_t r = 0; r < nranks && st == OKRT_OK; ++r) {
                    if (r == root) continue;
                    const uint32_t rows = rowsOf(t0.H, t0.stripeRows, r, nranks);
                    if (rows && !ncclOk(c, rccl().Recv(c->staging + (size_t)r * p.maxRows * t0.W, (size_t)rows * t0.W * 4, ncclFloat, (int)r, c->comm, f.stream), "ncclRecv"))
                        st = ctxFail(ctxs[k], OKRT_ERR_DEVICE, c->err);
                }
            } else if (f.tile.rows) {
                if (!ncclOk(c, rccl().Send(f.accum, (size_t)f.tile.rows * t0.W * 4, ncclFloat, (int)root, c->comm, f.stream), "ncclSend")) st = ctxFail(ctxs[k], OKRT_ERR_DEVICE, c->err);
            }
        }
        if (!ncclOk(comms[0], rccl().GroupEnd(), "ncclGroupEnd") && st == OKRT_OK) st = ctxFail(ctxs[0], OKRT_ERR_DEVICE, comms[0]->err);
        if (st != OKRT_OK) return st;
    }
    if (p.rootSlot < n) {
        st = deinterleaveOnRoot(ctxs[p.rootSlot], comms[p.rootSlot], p.fr[p.rootSlot], p.maxRows, fullFrameDev);
        if (st == OKRT_OK && nranks > 1 && p.copy) st = releasePeers(ctxs[p.rootSlot], p, comms);
        if (st == OKRT_OK) exchangeEnd(comms[p.rootSlot], p.fr[p.rootSlot]);
    }
    return st;
}

okrt_status okrt_mgpu_gather_accum(okrt_ctx* ctx, okrt_comm* comm, uint32_t root, void* fullFrameDev) {
    return okrt_mgpu_gather_accum_all(&ctx, &comm, 1, root, fullFrameDev);
}

// Samples split: every rank holds a whole-frame accumulation of its own block of frame indices; their SUM lands on the
// root (the library-owned frame, or full_frame_dev). One ncclReduce per rank inside one group; copy transport: the root
// pulls each peer's frame into staging and adds it.
okrt_status okrt_mgpu_reduce_accum_all(okrt_ctx* const* ctxs, okrt_comm* const* comms, uint32_t n, uint32_t root, void* fullFrameDev) {
    Plan p;
    okrt_status st = planCollective(ctxs, comms, n, root, true, fullFrameDev == nullptr, &p);
    if (st != OKRT_OK) return st;
    const uint32_t nranks = comms[0]->nranks;
    const TileView& t0 = p.fr[0].tile;
    const size_t px = (size_t)t0.W * t0.H;
    float4* dst = nullptr;
    if (p.rootSlot < n) dst = fullFrameDev ? (float4*)fullFrameDev : comms[p.rootSlot]->full;
    if (p.rootSlot < n) exchangeBegin(comms[p.rootSlot], p.fr[p.rootSlot]);
    if (nranks == 1 || p.copy) {
        const CtxFrame& rf = p.fr[p.rootSlot];
        if (hipSetDevice(rf.device) != hipSuccess || hipMemcpyAsync(dst, rf.accum, px * sizeof(float4), hipMemcpyDeviceToDevice, rf.stream) != hipSuccess)
            return ctxFail(ctxs[p.rootSlot], OKRT_ERR_DEVICE, "mgpu reduce: copy of the root's own frame failed");
        for (uint32_t k = 0; k < n && nranks > 1; ++k) {  // rank order: the sum is deterministic
            if (k == p.rootSlot) continue;
            st = pullFromPeer(ctxs[p.rootSlot], rf, comms[k], p.fr[k], comms[p.rootSlot]->staging, p.fr[k].accum, px * sizeof(float4));
            if (st != OKRT_OK) return st;
            hipLaunchKernelGGL(addFramesKernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, rf.stream, dst, comms[p.rootSlot]->staging, px);
            if (hipGetLastError() != hipSuccess) return ctxFail(ctxs[p.rootSlot], OKRT_ERR_DEVICE, "mgpu reduce: launch failed");
        }
        st = nranks > 1 ? releasePeers(ctxs[p.rootSlot], p, comms) : OKRT_OK;
        if (st == OKRT_OK) exchangeEnd(comms[p.rootSlot], rf);
        return st;
    }
    if (!ncclOk(comms[0], rccl().GroupStart(), "ncclGroupStart")) return ctxFail(ctxs[0], OKRT_ERR_DEVICE, comms[0]->err);
    for (uint32_t k = 0; k < n && st == OKRT_OK; ++k)
        if (hipSetDevice(p.fr[k].device) != hipSuccess) st = ctxFail(ctxs[k], OKRT_ERR_DEVICE, "mgpu reduce: hipSetDevice(" + std::to_string(p.fr[k].device) + ") failed");
        else if (!ncclOk(comms[k], rccl().Reduce(p.fr[k].accum, k == p.rootSlot ? (void*)dst : nullptr, px * 4, ncclFloat, ncclSum, (int)root, comms[k]->comm, p.fr[k].stream), "ncclReduce"))
            st = ctxFail(ctxs[k], OKRT_ERR_DEVICE, comms[k]->err);
    if (!ncclOk(comms[0], rccl().GroupEnd(), "ncclGroupEnd") && st == OKRT_OK) st = ctxFail(ctxs[0], OKRT_ERR_DEVICE, comms[0]->err);
    if (st == OKRT_OK && p.rootSlot < n) exchangeEnd(comms[p.rootSlot], p.fr[p.rootSlot]);
    return st;
}

okrt_status okrt_mgpu_reduce_accum(okrt_ctx* ctx, okrt_comm* comm, uint32_t root, void* fullFrameDev) {
    return okrt_mgpu_reduce_accum_all(&ctx, &comm, 1, root, fullFrameDev);
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
