// DirectX/RenderTexture.h (compat) — stands in for the reference's RenderTexture (source/DirectX/RenderTexture.h:33-70).
// Here a render texture is only a DESCRIPTION (width, height, format) plus the handle objects its getters hand out; the
// pixels live in the okrt context of whichever RayTracer renders into it (okrt_resize allocates result + accumulation:
// RayTracer.cpp:73,494-501). Every public name and signature of the original is available; the bodies are in
// RenderTexture.cpp.
#pragma once
#include "DX11.h"
#include "Utilities.h"

#include "glm/glm.hpp"

// bit positions and masks of the `flags` argument, and the three formats the reference instantiates
enum TextureFlagsBitPos : uint32_t { B_RENDER = 1, B_SHADER_READ, B_SHADER_WRITE, B_DEPTH };
enum TextureFlags : uint32_t { ALL = 1u, RENDER = 1u << B_RENDER, SHADER_READ = 1u << B_SHADER_READ, SHADER_WRITE = 1u << B_SHADER_WRITE, DEPTH = 1u << B_DEPTH };
enum TextureFormat : uint32_t { INVALID = 0u, F_8X1 = DXGI_FORMAT_R8_UNORM, F_8X4 = DXGI_FORMAT_R8G8B8A8_UNORM, F_32X4 = DXGI_FORMAT_R32G32B32A32_FLOAT };

class RenderTexture {
    // what a texture owns: the colour description, an optional depth description, one handle per view kind
    struct Handles {
        ID3D11Texture2D* colour = nullptr;
        ID3D11Texture2D* depth = nullptr;
        ID3D11RenderTargetView* rtv = nullptr;
        ID3D11ShaderResourceView* srv = nullptr;
        ID3D11UnorderedAccessView* uav = nullptr;
        ID3D11DepthStencilView* dsv = nullptr;
        bool ownsColour = false;
    } m;
    void makeViews(bool rtv, bool srv, bool uav, bool depth);

public:
    RenderTexture() = default;
    RenderTexture(ID3D11Texture2D* pDX11Texture, bool createDepthTexture = true) { initiate(pDX11Texture, createDepthTexture); }
    RenderTexture(uint32_t width, uint32_t height, TextureFormat format, uint32_t flags = TextureFlags::ALL) { initiate(width, height, format, flags); }
    ~RenderTexture() { shutdown(); }
    RenderTexture(const RenderTexture&) = delete;
    RenderTexture& operator=(const RenderTexture&) = delete;

    // lifetime
    void initiate(ID3D11Texture2D* pDX11Texture, bool createDepthTexture = true);
    void initiate(uint32_t width, uint32_t height, TextureFormat format, uint32_t flags = TextureFlags::ALL);
    void resize(uint32_t width, uint32_t height);  // keeps the handles; a bound RayTracer follows with onResize()
    void shutdown();

    // the handles (pointer-to-pointer like the original so call sites dereference the same way)
    ID3D11UnorderedAccessView* const* getUAV() const { return &m.uav; }
    ID3D11ShaderResourceView* const* getSRV() const { return &m.srv; }
    ID3D11RenderTargetView* const* getRTV() const { return &m.rtv; }
    ID3D11DepthStencilView* const* getDSV() const { return &m.dsv; }
    ID3D11Texture2D* const* getBuffer() const { return &m.colour; }
    glm::uvec2 getDimensions() const;

    static uint32_t getTextureFlagsFromDX11(uint32_t bindFlags, bool depth = false);
};
