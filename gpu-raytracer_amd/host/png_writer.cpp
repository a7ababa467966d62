// png_writer.cpp — see png_writer.h.
#include "png_writer.h"

#include <cstdio>
#include <vector>

namespace okh {

namespace {

uint32_t crc32(const uint8_t* p, size_t n, uint32_t crc = 0) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}

void be32(std::vector<uint8_t>& v, uint32_t x) {
    for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s));
}

void chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
    be32(out, (uint32_t)data.size());
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    be32(out, crc32(out.data() + start, out.size() - start));
}

}  // namespace

bool writePng(const std::string& path, uint32_t width, uint32_t height, const uint8_t* rgba8, std::string* error) {
    if (!width || !height || !rgba8) {
        if (error) *error = "writePng: empty image";
        return false;
    }
    // raw scanlines: one filter byte (0 = None) + the row
    std::vector<uint8_t> raw;
    raw.reserve((size_t)height * (width * 4 + 1));
    for (uint32_t y = 0; y < height; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), rgba8 + (size_t)y * width * 4, rgba8 + (size_t)(y + 1) * width * 4);
    }
    // zlib container: CMF/FLG, stored deflate blocks of at most 65535 bytes, Adler-32 of the raw data
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (size_t pos = 0; pos < raw.size();) {
        const size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
        z.push_back(pos + n == raw.size() ? 1 : 0);  // BFINAL, BTYPE = 00
        z.push_back((uint8_t)(n & 0xFF));
        z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF));
        z.push_back((uint8_t)((~n >> 8) & 0xFF));
        for (size_t i = 0; i < n; ++i) {
            a = (a + raw[pos + i]) % 65521u;
            b = (b + a) % 65521u;
        }
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
    }
    be32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    be32(ihdr, width);
    be32(ihdr, height);
    for (uint8_t v : {8, 6, 0, 0, 0}) ihdr.push_back(v);  // bit depth 8, colour type 6 (RGBA), deflate, adaptive filtering, no interlace
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f || std::fwrite(out.data(), 1, out.size(), f) != out.size()) {
        if (f) std::fclose(f);
        if (error) *error = "writePng: cannot write " + path;
        return false;
    }
    std::fclose(f);
    return true;
}

}  // namespace okh
