// Minimal PNG reader / writer on zlib.  Own code (the reference vendors stb for this).
#include "png_io.h"

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <vector>

namespace {
uint32_t be32(const unsigned char* p) { return (uint32_t)p[0] << 24 | p[1] << 16 | p[2] << 8 | p[3]; }
void put32(std::vector<unsigned char>& v, uint32_t x) {
    v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}
int paeth(int a, int b, int c) {
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
void chunk(std::vector<unsigned char>& out, const char* type, const unsigned char* data, size_t n) {
    put32(out, (uint32_t)n);
    size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    if (n) out.insert(out.end(), data, data + n);
    put32(out, (uint32_t)crc32(0, out.data() + at, (uInt)(n + 4)));
}
}  // namespace

unsigned char* smx_png_load(const char* path, int* w, int* h, int* channels) {
    FILE* f = fopen(path, "rb");
    if (!f) return nullptr;
    std::vector<unsigned char> buf;
    unsigned char tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (buf.size() < 8 || memcmp(buf.data(), sig, 8)) return nullptr;
    size_t pos = 8;
    int W = 0, H = 0, depth = 0, ctype = 0, interlace = 0;
    bool have_ihdr = false;
    std::vector<unsigned char> idat;
    while (pos + 12 <= buf.size()) {
        const uint32_t len = be32(&buf[pos]);
        const unsigned char* type = &buf[pos + 4];
        const unsigned char* data = &buf[pos + 8];
        if ((size_t)len > buf.size() || pos + 12 + (size_t)len > buf.size()) return nullptr;
        if (!have_ihdr) {
            // the first chunk must be a 13-byte IHDR; dimensions are bounded so that the sizes
            // below cannot overflow or exhaust memory on a hostile file
            if (memcmp(type, "IHDR", 4) || len != 13) return nullptr;
            const uint32_t uw = be32(data), uh = be32(data + 4);
            if (uw == 0 || uh == 0 || uw > SMX_PNG_MAX_DIM || uh > SMX_PNG_MAX_DIM) return nullptr;
            W = (int)uw; H = (int)uh;
            depth = data[8]; ctype = data[9]; interlace = data[12];
            have_ihdr = true;
        } else if (!memcmp(type, "IHDR", 4)) {
            return nullptr;
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + len;
    }
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!have_ihdr || depth != 8 || !ch || interlace || idat.empty()) return nullptr;
    const size_t stride = (size_t)W * ch;                  // <= 4 * 65535
    std::vector<unsigned char> raw;
    try {
        raw.resize((stride + 1) * (size_t)H);              // <= ~17 GB bound; bad_alloc is caught
    } catch (const std::exception&) {
        return nullptr;
    }
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK ||
        rawlen != raw.size())
        return nullptr;
    unsigned char* out = (unsigned char*)malloc(stride * (size_t)H);
    if (!out) return nullptr;
    for (int y = 0; y < H; ++y) {
        const unsigned char* in = &raw[(stride + 1) * y];
        unsigned char* cur = out + stride * y;
        const unsigned char* up = y ? out + stride * (y - 1) : nullptr;
        const int ft = in[0];
        for (size_t i = 0; i < stride; ++i) {
            int a = i >= (size_t)ch ? cur[i - ch] : 0;
            int b = up ? up[i] : 0;
            int c = (up && i >= (size_t)ch) ? up[i - ch] : 0;
            int x = in[1 + i];
            switch (ft) {
                case 0: break;
                case 1: x += a; break;
                case 2: x += b; break;
                case 3: x += (a + b) >> 1; break;
                case 4: x += paeth(a, b, c); break;
                default: free(out); return nullptr;
            }
            cur[i] = (unsigned char)x;
        }
    }
    *w = W; *h = H; *channels = ch;
    return out;
}

// 8-bit (bytes_per_sample = 1) or 16-bit big-endian samples (bytes_per_sample = 2, already in file order)
static int png_write_impl(const char* path, int w, int h, int channels, int bytes_per_sample,
                          const unsigned char* data) {
    const int ctype = channels == 1 ? 0 : channels == 3 ? 2 : channels == 2 ? 4 : channels == 4 ? 6 : -1;
    if (ctype < 0 || w <= 0 || h <= 0 || w > SMX_PNG_MAX_DIM || h > SMX_PNG_MAX_DIM || !data) return 0;
    const size_t stride = (size_t)w * channels * bytes_per_sample;
    std::vector<unsigned char> raw((stride + 1) * h);
    for (int y = 0; y < h; ++y) {
        raw[(stride + 1) * y] = 0;  // filter type None
        memcpy(&raw[(stride + 1) * y + 1], data + stride * y, stride);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<unsigned char> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return 0;
    std::vector<unsigned char> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    unsigned char ihdr[13];
    ihdr[0] = w >> 24; ihdr[1] = w >> 16; ihdr[2] = w >> 8; ihdr[3] = w;
    ihdr[4] = h >> 24; ihdr[5] = h >> 16; ihdr[6] = h >> 8; ihdr[7] = h;
    ihdr[8] = (unsigned char)(8 * bytes_per_sample); ihdr[9] = (unsigned char)ctype; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk(out, "IHDR", ihdr, 13);
    chunk(out, "IDAT", comp.data(), clen);
    chunk(out, "IEND", nullptr, 0);
    FILE* f = fopen(path, "wb");
    if (!f) return 0;
    size_t wr = fwrite(out.data(), 1, out.size(), f);
    const int closed = fclose(f);
    return wr == out.size() && closed == 0;
}

int smx_png_write(const char* path, int w, int h, int channels, const unsigned char* data) {
    return png_write_impl(path, w, h, channels, 1, data);
}

// 16-bit gray PNG (KITTI disparity convention: value = disparity * 256, 0 = invalid)
int smx_png_write_gray16(const char* path, int w, int h, const unsigned short* data) {
    if (w <= 0 || h <= 0 || !data) return 0;
    std::vector<unsigned char> be((size_t)w * h * 2);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        be[2 * i] = (unsigned char)(data[i] >> 8);
        be[2 * i + 1] = (unsigned char)(data[i] & 0xFF);
    }
    return png_write_impl(path, w, h, 1, 2, be.data());
}

// Portable float map (Middlebury disparity convention): "Pf", little-endian (scale -1), rows bottom-up
int smx_pfm_write(const char* path, int w, int h, const float* data) {
    if (w <= 0 || h <= 0 || !data) return 0;
    FILE* f = fopen(path, "wb");
    if (!f) return 0;
    bool ok = fprintf(f, "Pf\n%d %d\n-1.0\n", w, h) > 0;
    for (int y = h - 1; y >= 0 && ok; --y) ok = fwrite(data + (size_t)y * w, sizeof(float), (size_t)w, f) == (size_t)w;
    return (fclose(f) == 0) && ok;
}
