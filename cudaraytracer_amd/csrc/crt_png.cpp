// crt_png.cpp — RGBA8 PNG writer behind crt_write_png (include/crt.h).
// Replaces the reference's stbi_write_png(path, nx, ny, 4, rgb, 0) call (CudaTest/src/core/render.h:153)
// and produces the SAME FILE, byte for byte: the reference's encoder is the stb_image_write.h it vendors
// (CudaTest/stb-master/stb_image_write.h, third party, public domain), whose PNG path is a fixed procedure —
//   * per row the filter (None / Sub / Up / Average / Paeth, with the first row's missing neighbours taken
//     as in that file) whose output has the smallest sum of absolute signed bytes, the first such on ties
//     (stb_image_write.h:1092-1165);
//   * one zlib stream in one IDAT chunk: header 78 5e, a single fixed-Huffman block, LZ77 matches found
//     through a 16384-bucket hash of 3-byte prefixes with at most 2 x 8 positions per bucket (the older half
//     is dropped when a bucket is full), the longest match wins and among equals the most recent, a match is
//     abandoned for a literal when the next position has a strictly longer one ("lazy matching"), only the
//     first position of a match enters the table; stored blocks instead when that is not smaller; Adler-32
//     (stb_image_write.h:895-1021);
//   * IHDR / IDAT / IEND with CRC-32.
// This file restates that procedure (own code, own data structures).  tests/test_png.py pins it against
// the reference's encoder itself (oracle/_ref/libref_stb.so, built from the reference's file where it lies)
// and against fixtures that encoder produced (tests/golden/stb_png_*.npz).
#include "../../include/crt.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

namespace {

using Bytes = std::vector<unsigned char>;

// ---------------------------------------------------------------- CRC-32 (PNG chunks), Adler-32 (zlib)
uint32_t crc32_of(const unsigned char* p, size_t n) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? (0xedb88320u ^ (c >> 1)) : (c >> 1);
            table[i] = c;
        }
        ready = true;
    }
    uint32_t c = 0xffffffffu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xffu] ^ (c >> 8);
    return c ^ 0xffffffffu;
}

uint32_t adler32_of(const unsigned char* p, size_t n) {
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < n;) {
        const size_t stop = (n - i > 5552) ? i + 5552 : n;   // keeps the sums below 2^32
        for (; i < stop; ++i) { a += p[i]; b += a; }
        a %= 65521u;
        b %= 65521u;
    }
    return (b << 16) | a;
}

// ---------------------------------------------------------------- deflate, fixed Huffman codes (RFC 1951 3.2.6)
struct BitWriter {
    Bytes& out;
    uint32_t acc = 0;
    int n = 0;
    explicit BitWriter(Bytes& o) : out(o) {}
    void bits(uint32_t value, int count) {   // least significant bit first
        acc |= value << n;
        n += count;
        while (n >= 8) {
            out.push_back((unsigned char)(acc & 0xffu));
            acc >>= 8;
            n -= 8;
        }
    }
    void code(uint32_t value, int count) {   // Huffman codes go out most significant bit first
        uint32_t r = 0;
        for (int k = 0; k < count; ++k) r |= ((value >> k) & 1u) << (count - 1 - k);
        bits(r, count);
    }
    void symbol(int s) {                     // literal / length alphabet
        if (s <= 143) code(0x30u + (uint32_t)s, 8);
        else if (s <= 255) code(0x190u + (uint32_t)(s - 144), 9);
        else if (s <= 279) code((uint32_t)(s - 256), 7);
        else code(0xc0u + (uint32_t)(s - 280), 8);
    }
    void pad() {
        if (n > 0) bits(0u, 8 - n);
    }
};

const int kLengthBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const int kLengthExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const int kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const int kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

constexpr int kBuckets = 16384;
constexpr int kQuality = 8;          // stbi_write_png_compression_level's default; a bucket holds at most 2 * kQuality positions

uint32_t prefix_hash(const unsigned char* p) {   // stb_image_write.h:867-877 (the bucket of a 3-byte prefix must be theirs)
    uint32_t h = (uint32_t)p[0] + ((uint32_t)p[1] << 8) + ((uint32_t)p[2] << 16);
    h ^= h << 3;
    h += h >> 5;
    h ^= h << 4;
    h += h >> 17;
    h ^= h << 25;
    h += h >> 6;
    return h;
}

int match_length(const unsigned char* a, const unsigned char* b, int limit) {
    if (limit > 258) limit = 258;
    int k = 0;
    while (k < limit && a[k] == b[k]) ++k;
    return k;
}

// The zlib stream stbi_zlib_compress(data, n, &len, 8) returns.
Bytes zlib_stream(const unsigned char* data, int n) {
    Bytes out;
    out.push_back(0x78);
    out.push_back(0x5e);
    BitWriter w(out);
    w.bits(1u, 1);   // last block
    w.bits(1u, 2);   // fixed Huffman codes
    std::vector<std::vector<int>> bucket(kBuckets);   // positions, oldest first
    int i = 0;
    while (i < n - 3) {
        std::vector<int>& here = bucket[prefix_hash(data + i) & (kBuckets - 1)];
        int best = 3, at = -1;
        for (int pos : here)
            if (pos > i - 32768) {
                const int len = match_length(data + pos, data + i, n - i);
                if (len >= best) { best = len; at = pos; }      // equal lengths: the later position
            }
        if ((int)here.size() == 2 * kQuality) here.erase(here.begin(), here.begin() + kQuality);
        here.push_back(i);
        if (at >= 0) {   // does the next position have a strictly longer match?  then this byte goes out as a literal
            const std::vector<int>& next = bucket[prefix_hash(data + i + 1) & (kBuckets - 1)];
            for (int pos : next)
                if (pos > i - 32767 && match_length(data + pos, data + i + 1, n - i - 1) > best) {
                    at = -1;
                    break;
                }
        }
        if (at >= 0) {
            const int dist = i - at;
            int k = 28;
            while (best < kLengthBase[k]) --k;
            w.symbol(257 + k);
            if (kLengthExtra[k]) w.bits((uint32_t)(best - kLengthBase[k]), kLengthExtra[k]);
            int d = 29;
            while (dist < kDistBase[d]) --d;
            w.code((uint32_t)d, 5);
            if (kDistExtra[d]) w.bits((uint32_t)(dist - kDistBase[d]), kDistExtra[d]);
            i += best;
        } else {
            w.symbol(data[i]);
            ++i;
        }
    }
    for (; i < n; ++i) w.symbol(data[i]);
    w.symbol(256);
    w.pad();
    const int blocks = (n + 32766) / 32767;
    if ((long long)out.size() > (long long)n + 2 + 5LL * blocks) {   // not smaller than stored blocks: store
        out.resize(2);
        for (int j = 0; j < n;) {
            const int len = (n - j > 32767) ? 32767 : n - j;
            out.push_back((unsigned char)((n - j == len) ? 1 : 0));
            out.push_back((unsigned char)(len & 0xff));
            out.push_back((unsigned char)(len >> 8));
            out.push_back((unsigned char)(~len & 0xff));
            out.push_back((unsigned char)((~len >> 8) & 0xff));
            out.insert(out.end(), data + j, data + j + len);
            j += len;
        }
    }
    const uint32_t a = adler32_of(data, (size_t)n);
    out.push_back((unsigned char)(a >> 24));
    out.push_back((unsigned char)(a >> 16));
    out.push_back((unsigned char)(a >> 8));
    out.push_back((unsigned char)a);
    return out;
}

// ---------------------------------------------------------------- PNG row filters
int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return (pb <= pc) ? b : c;
}

// Row `y` of `px` (4 bytes per pixel) under filter `type`; the first row has no row above: Up copies, Average halves the
// left neighbour, Paeth predicts the left neighbour (the first pixel of any row has no left neighbour: predicted from above alone).
void filter_row(const unsigned char* px, int width, int y, int type, unsigned char* out) {
    const int n = 4, bytes = width * n;
    const unsigned char* z = px + (size_t)bytes * y;
    const unsigned char* up = (y > 0) ? z - bytes : nullptr;
    for (int i = 0; i < bytes; ++i) {
        const int left = (i >= n) ? z[i - n] : 0;
        const int above = up ? up[i] : 0;
        const int diag = (up && i >= n) ? up[i - n] : 0;
        int pred = 0;
        switch (type) {
            case 1: pred = left; break;
            case 2: pred = above; break;
            case 3: pred = (i >= n) ? ((left + above) >> 1) : (above >> 1); break;
            case 4: pred = (i >= n) ? paeth(left, above, diag) : paeth(0, above, 0); break;
            default: pred = 0; break;
        }
        out[i] = (unsigned char)(z[i] - pred);
    }
}

Bytes png_file(const unsigned char* rgba, int nx, int ny) {
    const int bytes = nx * 4;
    Bytes raw((size_t)(bytes + 1) * ny), line((size_t)bytes), best_line((size_t)bytes);
    for (int y = 0; y < ny; ++y) {
        int best_type = 0;
        long best_sum = 0x7fffffffL;
        for (int type = 0; type < 5; ++type) {
            filter_row(rgba, nx, y, type, line.data());
            long sum = 0;
            for (int i = 0; i < bytes; ++i) sum += abs((int)(signed char)line[i]);
            if (sum < best_sum) { best_sum = sum; best_type = type; best_line = line; }
        }
        raw[(size_t)(bytes + 1) * y] = (unsigned char)best_type;
        memcpy(&raw[(size_t)(bytes + 1) * y + 1], best_line.data(), (size_t)bytes);
    }
    const Bytes z = zlib_stream(raw.data(), (int)raw.size());

    Bytes out;
    auto u32 = [&](uint32_t v) {
        out.push_back((unsigned char)(v >> 24));
        out.push_back((unsigned char)(v >> 16));
        out.push_back((unsigned char)(v >> 8));
        out.push_back((unsigned char)v);
    };
    auto chunk = [&](const char tag[4], const unsigned char* data, size_t n) {
        u32((uint32_t)n);
        const size_t start = out.size();
        out.insert(out.end(), tag, tag + 4);
        if (n) out.insert(out.end(), data, data + n);
        u32(crc32_of(out.data() + start, n + 4));
    };
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    out.insert(out.end(), sig, sig + 8);
    unsigned char ihdr[13] = {(unsigned char)(nx >> 24), (unsigned char)(nx >> 16), (unsigned char)(nx >> 8), (unsigned char)nx,
                              (unsigned char)(ny >> 24), (unsigned char)(ny >> 16), (unsigned char)(ny >> 8), (unsigned char)ny,
                              8, 6, 0, 0, 0};   // 8 bits per channel, RGBA
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), z.size());
    chunk("IEND", nullptr, 0);
    return out;
}

bool size_ok(int nx, int ny) {   // the filtered image is indexed with int, as in the reference's encoder
    return nx > 0 && ny > 0 && (long long)(nx * 4LL + 1) * ny <= 0x7fffffffLL;
}

}  // namespace

extern "C" long long crt_encode_png_rgba8(const unsigned char* rgba, int nx, int ny, unsigned char* out, long long capacity) {
    if (!rgba || !size_ok(nx, ny) || capacity < 0) return CRT_ERR_INVALID;
    const Bytes file = png_file(rgba, nx, ny);
    if (out && (long long)file.size() <= capacity) memcpy(out, file.data(), file.size());
    return (long long)file.size();
}

extern "C" int crt_write_png_rgba8(const char* path, int nx, int ny, const unsigned char* rgba) {
    if (!path || !rgba || !size_ok(nx, ny)) return CRT_ERR_INVALID;
    const Bytes file = png_file(rgba, nx, ny);
    FILE* f = fopen(path, "wb");
    if (!f) return CRT_ERR_IO;
    const size_t w = fwrite(file.data(), 1, file.size(), f);
    const int rc = fclose(f);
    return (w == file.size() && rc == 0) ? CRT_OK : CRT_ERR_IO;
}

extern "C" int crt_write_png(const char* path, int nx, int ny, const float* rgb) {
    if (!path || !rgb || nx <= 0 || ny <= 0) return CRT_ERR_INVALID;
    std::vector<unsigned char> rgba((size_t)nx * ny * 4);
    crt_to_rgba8(rgb, nx, ny, rgba.data());
    return crt_write_png_rgba8(path, nx, ny, rgba.data());
}
