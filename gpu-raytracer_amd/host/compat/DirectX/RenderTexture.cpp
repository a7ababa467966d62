// DirectX/RenderTexture.cpp (compat) — see RenderTexture.h.
#include "RenderTexture.h"

namespace {
// D3D11_BIND_* values (d3d11.h): shader resource, render target, depth stencil, unordered access
constexpr uint32_t kBindSRV = 0x8u, kBindRTV = 0x20u, kBindDSV = 0x40u, kBindUAV = 0x80u;
}  // namespace

uint32_t RenderTexture::getTextureFlagsFromDX11(uint32_t bindFlags, bool depth) {
    uint32_t flags = 0;
    flags |= (bindFlags & kBindRTV) ? (uint32_t)TextureFlags::RENDER : 0u;
    flags |= (bindFlags & kBindSRV) ? (uint32_t)TextureFlags::SHADER_READ : 0u;
    flags |= (bindFlags & kBindUAV) ? (uint32_t)TextureFlags::SHADER_WRITE : 0u;
    flags |= (depth || (bindFlags & kBindDSV)) ? (uint32_t)TextureFlags::DEPTH : 0u;
    return flags;
}

void RenderTexture::makeViews(bool rtv, bool srv, bool uav, bool depth) {
    if (rtv) m.rtv = new ID3D11RenderTargetView{m.colour};
    if (srv) m.srv = new ID3D11ShaderResourceView();
    if (uav) m.uav = new ID3D11UnorderedAccessView{m.colour};
    if (depth) {
        m.depth = new ID3D11Texture2D();
        m.depth->desc = m.colour->desc;
        m.dsv = new ID3D11DepthStencilView{m.depth};
    }
}

void RenderTexture::initiate(uint32_t width, uint32_t height, TextureFormat format, uint32_t flags) {
    shutdown();
    m.colour = new ID3D11Texture2D();
    m.ownsColour = true;
    m.colour->desc.Width = width;
    m.colour->desc.Height = height;
    m.colour->desc.Format = (DXGI_FORMAT)format;
    const bool all = flags == TextureFlags::ALL;
    makeViews(all || (flags & TextureFlags::RENDER), all || (flags & TextureFlags::SHADER_READ), all || (flags & TextureFlags::SHADER_WRITE),
              all || (flags & TextureFlags::DEPTH));
}

void RenderTexture::initiate(ID3D11Texture2D* pDX11Texture, bool createDepthTexture) {
    shutdown();
    m.colour = pDX11Texture;  // not owned: the caller's texture (the swap-chain back buffer in the reference)
    if (m.colour) makeViews(true, true, true, createDepthTexture);
}

void RenderTexture::shutdown() {
    if (m.ownsColour) delete m.colour;
    delete m.depth;
    delete m.rtv;
    delete m.srv;
    delete m.uav;
    delete m.dsv;
    m = Handles();
}

void RenderTexture::resize(uint32_t width, uint32_t height) {
    if (!m.colour) return;
    m.colour->desc.Width = width;
    m.colour->desc.Height = height;
    if (m.depth) m.depth->desc = m.colour->desc;
}

glm::uvec2 RenderTexture::getDimensions() const {
    D3D11_TEXTURE2D_DESC desc{};
    if (m.colour) m.colour->GetDesc(&desc);
    return glm::uvec2(desc.Width, desc.Height);
}
