// DirectX/RenderTexture.cpp (compat) — see RenderTexture.h.
#include "RenderTexture.h"

uint32_t RenderTexture::getTextureFlagsFromDX11(uint32_t bindFlags, bool depth) {
    // D3D11_BIND_SHADER_RESOURCE 0x8, RENDER_TARGET 0x20, DEPTH_STENCIL 0x40, UNORDERED_ACCESS 0x80 (d3d11.h)
    uint32_t flags = 0;
    if (bindFlags & 0x20u) flags |= TextureFlags::RENDER;
    if (bindFlags & 0x8u) flags |= TextureFlags::SHADER_READ;
    if (bindFlags & 0x80u) flags |= TextureFlags::SHADER_WRITE;
    if (depth || (bindFlags & 0x40u)) flags |= TextureFlags::DEPTH;
    return flags;
}

RenderTexture::RenderTexture() : m_pBuffer(nullptr), m_pRTV(nullptr), m_pSRV(nullptr), m_pUAV(nullptr), m_pDepthBuffer(nullptr), m_pDSV(nullptr), m_ownsBuffer(false) {}

RenderTexture::RenderTexture(uint32_t width, uint32_t height, TextureFormat format, uint32_t flags) : RenderTexture() { initiate(width, height, format, flags); }

RenderTexture::RenderTexture(ID3D11Texture2D* pDX11Texture, bool createDepthTexture) : RenderTexture() { initiate(pDX11Texture, createDepthTexture); }

RenderTexture::~RenderTexture() { shutdown(); }

void RenderTexture::initiate(uint32_t width, uint32_t height, TextureFormat format, uint32_t flags) {
    shutdown();
    m_pBuffer = new ID3D11Texture2D();
    m_ownsBuffer = true;
    m_pBuffer->desc.Width = width;
    m_pBuffer->desc.Height = height;
    m_pBuffer->desc.Format = (DXGI_FORMAT)format;
    const bool all = flags == TextureFlags::ALL;
    if (all || (flags & TextureFlags::RENDER)) m_pRTV = new ID3D11RenderTargetView{m_pBuffer};
    if (all || (flags & TextureFlags::SHADER_READ)) m_pSRV = new ID3D11ShaderResourceView();
    if (all || (flags & TextureFlags::SHADER_WRITE)) m_pUAV = new ID3D11UnorderedAccessView{m_pBuffer};
    if (all || (flags & TextureFlags::DEPTH)) {
        m_pDepthBuffer = new ID3D11Texture2D();
        m_pDepthBuffer->desc = m_pBuffer->desc;
        m_pDSV = new ID3D11DepthStencilView{m_pDepthBuffer};
    }
}

void RenderTexture::initiate(ID3D11Texture2D* pDX11Texture, bool createDepthTexture) {
    shutdown();
    m_pBuffer = pDX11Texture;
    m_ownsBuffer = false;
    if (!m_pBuffer) return;
    m_pRTV = new ID3D11RenderTargetView{m_pBuffer};
    m_pSRV = new ID3D11ShaderResourceView();
    m_pUAV = new ID3D11UnorderedAccessView{m_pBuffer};
    if (createDepthTexture) {
        m_pDepthBuffer = new ID3D11Texture2D();
        m_pDepthBuffer->desc = m_pBuffer->desc;
        m_pDSV = new ID3D11DepthStencilView{m_pDepthBuffer};
    }
}

void RenderTexture::shutdown() {
    if (m_ownsBuffer) OKAY_DELETE(m_pBuffer);
    m_pBuffer = nullptr;
    OKAY_DELETE(m_pRTV);
    OKAY_DELETE(m_pSRV);
    OKAY_DELETE(m_pUAV);
    OKAY_DELETE(m_pDepthBuffer);
    OKAY_DELETE(m_pDSV);
    m_ownsBuffer = false;
}

void RenderTexture::resize(uint32_t width, uint32_t height) {
    if (!m_pBuffer) return;
    // the views keep pointing at the same description; a RayTracer bound to it re-reads the size in onResize()
    m_pBuffer->desc.Width = width;
    m_pBuffer->desc.Height = height;
    if (m_pDepthBuffer) m_pDepthBuffer->desc = m_pBuffer->desc;
}
