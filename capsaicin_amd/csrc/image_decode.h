// image_decode.h — decoders shared between image_decode.cpp (container sniffing, PNG / TGA / PNM) and jpeg_decode.cpp.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace cap
{
// 8-bit RGBA, rows top to bottom; false = not a JPEG this build decodes (or a corrupt one)
bool decode_jpeg(const uint8_t* data, size_t size, std::vector<uint8_t>* rgba, uint32_t* w, uint32_t* h, uint64_t max_pixels);
}  // namespace cap
