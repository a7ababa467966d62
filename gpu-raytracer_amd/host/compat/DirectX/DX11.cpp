// DirectX/DX11.cpp (compat) — see DX11.h.
#include "DX11.h"

#include <cstring>

namespace {
const okrt_api* g_api = nullptr;
int g_device = 0;
}  // namespace

namespace okay_compat {
void setDeviceApi(const okrt_api* api, int device) {
    g_api = api;
    g_device = device;
}
const okrt_api* deviceApi() { return g_api ? g_api : okrt_get_api(); }
int deviceOrdinal() { return g_device; }
}  // namespace okay_compat

namespace Okay {

void initiateDX11() {}
void shutdownDX11() {}

void getCPUTextureData(ID3D11Texture2D* pSourceTexture, void** ppOutData) {
    if (!ppOutData) return;
    *ppOutData = nullptr;
    if (!pSourceTexture) return;
    const size_t bytes = (size_t)pSourceTexture->desc.Width * pSourceTexture->desc.Height * 4u;
    unsigned char* data = new unsigned char[bytes ? bytes : 1];
    std::memset(data, 0, bytes);
    if (pSourceTexture->ctx && pSourceTexture->api && pSourceTexture->desc.Format == DXGI_FORMAT_R8G8B8A8_UNORM)
        pSourceTexture->api->read_result_rgba8(pSourceTexture->ctx, data);
    *ppOutData = data;
}

}  // namespace Okay
