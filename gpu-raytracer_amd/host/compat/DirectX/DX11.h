// DirectX/DX11.h (compat) — the `Okay::` device functions the compute path and Application::saveScreenshot use
// (reference source/DirectX/DX11.h:10-36), over the okrt C ABI. Buffer / shader / swap-chain creation has no counterpart:
// tables go through GPUStorage -> okrt_set_table, kernels are compiled ahead of time.
#pragma once
#include <d3d11.h>

#include <string>
#include <vector>

class Texture;

namespace Okay {
void initiateDX11();   // DX11.cpp:21-42: nothing to create; contexts are made by RayTracer::initiate
void shutdownDX11();
// DX11.cpp:196-230: *ppOutData = new unsigned char[W * H * 4] (release with OKAY_DELETE_ARRAY) holding the RGBA8 pixels
// of the texture, row 0 = top; zeros when no RayTracer has rendered into it. The buffer has the TEXTURE's size; pixels are
// read at the rendering context's size and the overlap is copied (they differ between RenderTexture::resize and
// RayTracer::onResize).
void getCPUTextureData(ID3D11Texture2D* pSourceTexture, void** ppOutData);
}  // namespace Okay

// Which device layer the facade classes drive: okrt_get_api() of libokrt.so unless a host (or a test) installs another
// implementation of the same table before the first RayTracer is initiated.
namespace okay_compat {
void setDeviceApi(const okrt_api* api, int device = 0);
const okrt_api* deviceApi();
int deviceOrdinal();
// true: Okay::getCPUTextureData returns the newest RESOLVED picture (okrt_peek_result_rgba8) instead of executing the
// deferred frames first — for a loop that reads the texture back every frame to present it; screenshots keep `false`.
void setReadbackWithoutFlush(bool on);
}  // namespace okay_compat
