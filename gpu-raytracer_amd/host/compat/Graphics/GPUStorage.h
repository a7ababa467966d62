// Graphics/GPUStorage.h (compat) — the reference's GPUStorage (source/Graphics/GPUStorage.h:5-47): a structured buffer
// of `capacity` elements of `elementByteWidth` bytes. Here: a host staging copy that is pushed to the okrt table slot
// the storage is bound to (okrt_set_table copies during the call, like D3D11 copies pSysMem: DX11.cpp:113-118).
#pragma once
#include <vector>

#include "DirectX/DX11.h"

class GPUStorage {
public:
    GPUStorage();
    GPUStorage(uint32_t elementByteWidth, uint32_t capacity, const void* pData);
    ~GPUStorage();

    void shutdown();
    void initiate(uint32_t elementByteWidth, uint32_t capacity, const void* pData);

    template <typename UpdateFunction>
    void update(uint32_t newCapacity, UpdateFunction function);
    void updateRaw(uint32_t newCapacity, const void* pData);

    inline ID3D11ShaderResourceView* getSRV() const { return const_cast<ID3D11ShaderResourceView*>(&m_srv); }
    inline uint32_t getCapacity() const { return m_capacity; }

    // compat only: route this storage to a table slot of a device context (RayTracer does this for t0..t10), or record
    // what a table uploaded elsewhere holds (`adopt`: capacity bookkeeping without a second upload).
    void bind(const okrt_api* api, okrt_ctx* ctx, okrt_table slot);
    void adopt(uint32_t elementByteWidth, uint32_t capacity);
    okrt_status lastStatus() const { return m_status; }

private:
    void push();

    std::vector<char> m_staging;
    ID3D11ShaderResourceView m_srv;
    const okrt_api* m_api;
    okrt_ctx* m_ctx;
    uint32_t m_capacity;
    uint32_t m_elementByteWidth;
    okrt_status m_status;
};

template <typename UpdateFunction>
inline void GPUStorage::update(uint32_t newCapacity, UpdateFunction function) {
    if (m_capacity != newCapacity && newCapacity) initiate(m_elementByteWidth, newCapacity, nullptr);  // GPUStorage.h:36-37
    if (m_staging.empty()) return;  // the reference's Map fails on a missing buffer and returns (GPUStorage.h:41-42)
    function(m_staging.data());
    push();
}
