// DirectX/DX11.cpp (compat) — see DX11.h.
#include "DX11.h"

#include <cstring>

#include <algorithm>
#include <vector>

namespace {
const okrt_api* g_api = nullptr;
int g_device = 0;
bool g_peek = false;
}  // namespace

namespace okay_compat {
void setDeviceApi(const okrt_api* api, int device) {
    g_api = api;
    g_device = device;
}
const okrt_api* deviceApi() { return g_api ? g_api : okrt_get_api(); }
int deviceOrdinal() { return g_device; }
void setReadbackWithoutFlush(bool on) { g_peek = on; }
}  // namespace okay_compat

namespace Okay {

void initiateDX11() {}
void shutdownDX11() {}

void getCPUTextureData(ID3D11Texture2D* pSourceTexture, void** ppOutData) {
    if (!ppOutData) return;
    *ppOutData = nullptr;
    if (!pSourceTexture) return;
    // the caller sizes its use of the buffer by the TEXTURE's description (Application.cpp:666-671); the pixels come from
    // the context, whose frame can differ between RenderTexture::resize() and RayTracer::onResize(): read at the
    // context's size, hand out the description's size, copy what overlaps
    const uint32_t W = pSourceTexture->desc.Width, H = pSourceTexture->desc.Height;
    const size_t bytes = (size_t)W * H * 4u;
    unsigned char* data = new unsigned char[bytes ? bytes : 1];
    std::memset(data, 0, bytes);
    *ppOutData = data;
    const std::shared_ptr<okay_compat::PixelSource> src = pSourceTexture->pixels;
    if (!src || !src->ctx || !src->api || pSourceTexture->desc.Format != DXGI_FORMAT_R8G8B8A8_UNORM) return;
    uint32_t cw = W, ch = H, rows = 0;  // (a device layer older than ABI 3 cannot say: the description is trusted for the width)
    if ((src->api->frame_size && src->api->frame_size(src->ctx, &cw, &ch) != OKRT_OK) || src->api->tile_rows(src->ctx, &rows) != OKRT_OK || !cw || !rows) return;
    const bool peek = g_peek && src->api->peek_result_rgba8;
    if (cw == W && rows == H) {  // the usual case: straight into the caller's buffer
        if (peek) src->api->peek_result_rgba8(src->ctx, data, nullptr);
        else src->api->read_result_rgba8(src->ctx, data);
        return;
    }
    std::vector<unsigned char> frame((size_t)cw * rows * 4u);
    if ((peek ? src->api->peek_result_rgba8(src->ctx, frame.data(), nullptr) : src->api->read_result_rgba8(src->ctx, frame.data())) != OKRT_OK) return;
    const size_t rowBytes = (size_t)std::min(cw, W) * 4u;
    for (uint32_t y = 0; y < std::min(rows, H); ++y) std::memcpy(data + (size_t)y * W * 4u, frame.data() + (size_t)y * cw * 4u, rowBytes);
}

}  // namespace Okay
