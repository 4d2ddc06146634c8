// Minimal PNG reader / writer on zlib.  Own code (the reference vendors stb for this).
#include "png_io.h"

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
    std::vector<unsigned char> idat;
    while (pos + 12 <= buf.size()) {
        uint32_t len = be32(&buf[pos]);
        const unsigned char* type = &buf[pos + 4];
        const unsigned char* data = &buf[pos + 8];
        if (pos + 12 + len > buf.size()) return nullptr;
        if (!memcmp(type, "IHDR", 4)) {
            W = (int)be32(data); H = (int)be32(data + 4);
            depth = data[8]; ctype = data[9]; interlace = data[12];
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + len;
    }
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!W || !H || depth != 8 || !ch || interlace) return nullptr;
    const size_t stride = (size_t)W * ch;
    std::vector<unsigned char> raw((stride + 1) * H);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK ||
        rawlen != raw.size())
        return nullptr;
    unsigned char* out = (unsigned char*)malloc(stride * H);
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

int smx_png_write(const char* path, int w, int h, int channels, const unsigned char* data) {
    const int ctype = channels == 1 ? 0 : channels == 3 ? 2 : channels == 2 ? 4 : channels == 4 ? 6 : -1;
    if (ctype < 0 || w <= 0 || h <= 0) return 0;
    const size_t stride = (size_t)w * channels;
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
    ihdr[8] = 8; ihdr[9] = (unsigned char)ctype; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk(out, "IHDR", ihdr, 13);
    chunk(out, "IDAT", comp.data(), clen);
    chunk(out, "IEND", nullptr, 0);
    FILE* f = fopen(path, "wb");
    if (!f) return 0;
    size_t wr = fwrite(out.data(), 1, out.size(), f);
    fclose(f);
    return wr == out.size();
}
