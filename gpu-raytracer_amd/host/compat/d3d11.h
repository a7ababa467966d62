// d3d11.h (compat) — the few Direct3D 11 names that appear in the PUBLIC interfaces of the reference's RayTracer,
// GPUStorage and RenderTexture classes and in Application::saveScreenshot (Application.cpp:661-674), as plain C++
// types over the okrt C ABI. Not a Direct3D implementation: there is no device, no context, no swap chain.
#pragma once
#include <cstdint>
#include <memory>

#include "okrt.h"

typedef unsigned int UINT;
typedef void* HWND;

enum DXGI_FORMAT : uint32_t {  // the three formats RenderTexture.h:17-23 names, with their dxgiformat.h values
    DXGI_FORMAT_UNKNOWN = 0,
    DXGI_FORMAT_R32G32B32A32_FLOAT = 2,
    DXGI_FORMAT_R8G8B8A8_UNORM = 28,
    DXGI_FORMAT_R8_UNORM = 61,
};

struct D3D11_TEXTURE2D_DESC {
    UINT Width = 0, Height = 0, MipLevels = 1, ArraySize = 1;
    DXGI_FORMAT Format = DXGI_FORMAT_UNKNOWN;
    UINT BindFlags = 0;
};

// Where a texture's pixels live once a RayTracer renders into it. Shared by the texture and that RayTracer, so either
// may be destroyed first: the RayTracer clears the link it holds, never reaching through a texture pointer.
namespace okay_compat {
struct PixelSource {
    const okrt_api* api = nullptr;
    okrt_ctx* ctx = nullptr;
};
}  // namespace okay_compat

// A 2D texture = its description + (once a RayTracer renders into it) the okrt context that holds the pixels.
struct ID3D11Texture2D {
    D3D11_TEXTURE2D_DESC desc;
    std::shared_ptr<okay_compat::PixelSource> pixels;
    void GetDesc(D3D11_TEXTURE2D_DESC* out) const { *out = desc; }
};
// Views are handles only: the compute path binds tables by okrt slot, not by view.
struct ID3D11ShaderResourceView { uint32_t slot = ~0u; };
struct ID3D11UnorderedAccessView { ID3D11Texture2D* texture = nullptr; };
struct ID3D11RenderTargetView { ID3D11Texture2D* texture = nullptr; };
struct ID3D11DepthStencilView { ID3D11Texture2D* texture = nullptr; };
struct ID3D11Buffer;
struct ID3D11Device;
struct ID3D11DeviceContext;
struct ID3D11ComputeShader;
struct IDXGISwapChain;
