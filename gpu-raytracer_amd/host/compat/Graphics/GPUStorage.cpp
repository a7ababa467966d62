// Graphics/GPUStorage.cpp (compat) — see GPUStorage.h.
#include "GPUStorage.h"

#include <cstring>

GPUStorage::GPUStorage() : m_api(nullptr), m_ctx(nullptr), m_capacity(0u), m_elementByteWidth(0u), m_status(OKRT_OK) {}

GPUStorage::GPUStorage(uint32_t elementByteWidth, uint32_t capacity, const void* pData) : GPUStorage() { initiate(elementByteWidth, capacity, pData); }

GPUStorage::~GPUStorage() { shutdown(); }

void GPUStorage::shutdown() {
    m_staging.clear();
    m_staging.shrink_to_fit();
    m_capacity = 0u;
}

void GPUStorage::initiate(uint32_t elementByteWidth, uint32_t capacity, const void* pData) {
    m_elementByteWidth = elementByteWidth;
    m_capacity = capacity;
    m_staging.assign((size_t)elementByteWidth * capacity, 0);
    if (pData && !m_staging.empty()) std::memcpy(m_staging.data(), pData, m_staging.size());
    if (pData) push();
}

void GPUStorage::updateRaw(uint32_t newCapacity, const void* pData) {
    update(newCapacity, [&](char* pMappedBufferData) {
        if (pData) std::memcpy(pMappedBufferData, pData, (size_t)m_elementByteWidth * m_capacity);
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

void GPUStorage::push() {
    if (!m_api || !m_ctx || m_srv.slot == ~0u) return;
    m_status = m_api->set_table(m_ctx, (okrt_table)m_srv.slot, m_staging.data(), m_elementByteWidth, m_capacity);
}
