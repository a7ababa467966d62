// Graphics/GPUStorage.h (compat) — stands in for the reference's GPUStorage (source/Graphics/GPUStorage.h:5-47): a
// structured buffer of `capacity` elements of `elementByteWidth` bytes. Here it is a host staging copy that is pushed to
// the okrt table slot the storage is bound to — okrt_set_table copies during the call, the way D3D11 copies pSysMem at
// creation (DX11.cpp:113-118). The reference's public names and signatures are all here.
#pragma once
#include <cstring>
#include <vector>

#include "DirectX/DX11.h"

class GPUStorage {
    std::vector<char> m_staging;          // what the table holds (empty once adopted from an upload made elsewhere)
    ID3D11ShaderResourceView m_srv;       // carries the okrt slot this storage feeds
    const okrt_api* m_api = nullptr;
    okrt_ctx* m_ctx = nullptr;
    uint32_t m_elementByteWidth = 0u, m_capacity = 0u;
    okrt_status m_status = OKRT_OK;

    void push() {  // staging -> device table (no-op while unbound)
        if (m_api && m_ctx && m_srv.slot != ~0u) m_status = m_api->set_table(m_ctx, (okrt_table)m_srv.slot, m_staging.data(), m_elementByteWidth, m_capacity);
    }

public:
    GPUStorage() = default;
    GPUStorage(uint32_t elementByteWidth, uint32_t capacity, const void* pData) { initiate(elementByteWidth, capacity, pData); }
    ~GPUStorage() { shutdown(); }

    uint32_t getCapacity() const { return m_capacity; }
    ID3D11ShaderResourceView* getSRV() const { return const_cast<ID3D11ShaderResourceView*>(&m_srv); }

    void initiate(uint32_t elementByteWidth, uint32_t capacity, const void* pData);
    void shutdown();
    void updateRaw(uint32_t newCapacity, const void* pData);

    // Re-sizes when the element count changed (GPUStorage.h:36-37), lets `function(char* mapped)` fill the staging copy
    // (the reference maps the buffer WRITE_DISCARD), then uploads. Without storage the call returns like a failed Map.
    template <typename UpdateFunction>
    void update(uint32_t newCapacity, UpdateFunction function) {
        if (newCapacity && newCapacity != m_capacity) initiate(m_elementByteWidth, newCapacity, nullptr);
        if (m_staging.empty()) return;
        function(m_staging.data());
        push();
    }

    // compat only: route this storage to a table slot of a device context (RayTracer does this for t0..t10), or record
    // what a table uploaded elsewhere holds (`adopt`: capacity bookkeeping without a second upload).
    void bind(const okrt_api* api, okrt_ctx* ctx, okrt_table slot);
    void adopt(uint32_t elementByteWidth, uint32_t capacity);
    okrt_status lastStatus() const { return m_status; }
};
