// Graphics/GPUStorage.cpp (compat) — see GPUStorage.h.
#include "GPUStorage.h"

void GPUStorage::initiate(uint32_t elementByteWidth, uint32_t capacity, const void* pData) {
    m_elementByteWidth = elementByteWidth;
    m_capacity = capacity;
    m_staging.assign((size_t)elementByteWidth * capacity, 0);
    if (!pData) return;  // sized now, filled by the first update()
    if (!m_staging.empty()) std::memcpy(m_staging.data(), pData, m_staging.size());
    push();
}

void GPUStorage::shutdown() {
    std::vector<char>().swap(m_staging);
    m_capacity = 0u;
}

void GPUStorage::updateRaw(uint32_t newCapacity, const void* pData) {
    update(newCapacity, [this, pData](char* mapped) {
        if (pData) std::memcpy(mapped, pData, m_staging.size());
    });
}

void GPUStorage::bind(const okrt_api* api, okrt_ctx* ctx, okrt_table slot) {
    m_api = api;
    m_ctx = ctx;
    m_srv.slot = (uint32_t)slot;
}

void GPUStorage::adopt(uint32_t elementByteWidth, uint32_t capacity) {
    m_elementByteWidth = elementByteWidth;
    m_capacity = capacity;
    m_staging.clear();
}
