// crt_png.cpp — RGBA8 PNG writer behind crt_write_png (include/crt.h).
// Replaces the reference's stbi_write_png(path, nx, ny, 4, rgb, 0) call
// (CudaTest/src/core/render.h:153).  Own encoder on top of zlib (deflate + crc32): the
// decoded pixels are what the reference writes; the compressed bytes are not stb's.
#include "../../include/crt.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <vector>

namespace {

void put_u32(std::vector<unsigned char>& v, uint32_t x) {
    v.push_back((unsigned char)(x >> 24));
    v.push_back((unsigned char)(x >> 16));
    v.push_back((unsigned char)(x >> 8));
    v.push_back((unsigned char)x);
}

void put_chunk(std::vector<unsigned char>& out, const char tag[4], const unsigned char* data, size_t n) {
    put_u32(out, (uint32_t)n);
    size_t start = out.size();
    out.insert(out.end(), tag, tag + 4);
    if (n) out.insert(out.end(), data, data + n);
    uint32_t crc = (uint32_t)crc32(0L, out.data() + start, (uInt)(n + 4));
    put_u32(out, crc);
}

}  // namespace

extern "C" int crt_write_png_rgba8(const char* path, int nx, int ny, const unsigned char* rgba) {
    if (!path || !rgba || nx <= 0 || ny <= 0) return CRT_ERR_INVALID;
    const size_t stride = (size_t)nx * 4;
    std::vector<unsigned char> raw((stride + 1) * (size_t)ny);
    for (int y = 0; y < ny; ++y) {
        raw[(stride + 1) * y] = 0;   // filter type 0 (None)
        memcpy(&raw[(stride + 1) * y + 1], rgba + stride * y, stride);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 8) != Z_OK) return CRT_ERR_IO;

    std::vector<unsigned char> out;
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    out.insert(out.end(), sig, sig + 8);
    std::vector<unsigned char> ihdr;
    put_u32(ihdr, (uint32_t)nx);
    put_u32(ihdr, (uint32_t)ny);
    ihdr.push_back(8);   // bit depth
    ihdr.push_back(6);   // colour type RGBA
    ihdr.push_back(0);
    ihdr.push_back(0);
    ihdr.push_back(0);
    put_chunk(out, "IHDR", ihdr.data(), ihdr.size());
    put_chunk(out, "IDAT", z.data(), zlen);
    put_chunk(out, "IEND", nullptr, 0);

    FILE* f = fopen(path, "wb");
    if (!f) return CRT_ERR_IO;
    size_t w = fwrite(out.data(), 1, out.size(), f);
    int rc = fclose(f);
    return (w == out.size() && rc == 0) ? CRT_OK : CRT_ERR_IO;
}

extern "C" int crt_write_png(const char* path, int nx, int ny, const float* rgb) {
    if (!path || !rgb || nx <= 0 || ny <= 0) return CRT_ERR_INVALID;
    std::vector<unsigned char> rgba((size_t)nx * ny * 4);
    crt_to_rgba8(rgb, nx, ny, rgba.data());
    return crt_write_png_rgba8(path, nx, ny, rgba.data());
}
