// png_writer.h — "Save screenshot" (reference Application::saveScreenshot, Application.cpp:661-674: read the target
// texture back and stbi_write_png("render.png", W, H, 4, data, W * 4)). A dependency-free PNG encoder: 8-bit RGBA,
// filter 0 on every row, zlib stream of stored (uncompressed) deflate blocks, CRC-32 / Adler-32 computed here.
#pragma once
#include <cstdint>
#include <string>

namespace okh {

// rgba8: height rows of width * 4 bytes, row 0 = top. Returns false (with `error`) if the file cannot be written.
bool writePng(const std::string& path, uint32_t width, uint32_t height, const uint8_t* rgba8, std::string* error = nullptr);

}  // namespace okh
