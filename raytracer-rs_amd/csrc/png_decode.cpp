// png_decode.cpp — PNG -> RGB f32 texels (byte / 256.0), the texture load of
// raytracer_lib/src/scene/texture.rs:35-49 (`image::open(path)?.to_rgb8()` then /256.0).
// Non-interlaced 8-bit grey / grey+alpha / RGB / RGBA / palette images; zlib does the inflate.
#include <zlib.h>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include "scene.hpp"

namespace mi355rt {
namespace {
uint32_t be32(const unsigned char* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
}

bool load_png_rgb(const std::string& path, TextureData& out, std::string& err)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "No such file or directory (os error 2)"; return false; }
    std::vector<unsigned char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    if (buf.size() < 8 || std::memcmp(buf.data(), sig, 8) != 0) { err = "Format error decoding Png: Invalid PNG signature."; return false; }
    uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte;
    size_t p = 8;
    while (p + 12 <= buf.size()) {
        uint32_t len = be32(&buf[p]);
        const unsigned char* type = &buf[p + 4];
        const unsigned char* d = &buf[p + 8];
        if (p + 12 + (size_t)len > buf.size()) { err = "Format error decoding Png: truncated chunk"; return false; }
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) { w = be32(d); h = be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12]; }
        else if (!std::memcmp(type, "PLTE", 4)) plte.assign(d, d + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        p += 12 + (size_t)len;
    }
    if (!w || !h) { err = "Format error decoding Png: missing IHDR"; return false; }
    if (depth != 8 || interlace != 0) { err = "Unsupported PNG (only 8-bit non-interlaced images are handled)"; return false; }
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch) { err = "Format error decoding Png: bad colour type"; return false; }
    size_t stride = (size_t)w * ch;
    std::vector<unsigned char> raw((stride + 1) * h);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) {
        err = "Format error decoding Png: corrupt deflate stream"; return false;
    }
    std::vector<unsigned char> img(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        int ft = raw[(stride + 1) * y];
        const unsigned char* src = &raw[(stride + 1) * y + 1];
        unsigned char* dst = &img[stride * y];
        const unsigned char* up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t x = 0; x < stride; ++x) {
            int a = x >= (size_t)ch ? dst[x - ch] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)ch) ? up[x - ch] : 0;
            int v = src[x];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) / 2; break;
                case 4: v += paeth(a, b, c); break;
                default: err = "Format error decoding Png: bad filter type"; return false;
            }
            dst[x] = (unsigned char)v;
        }
    }
    out.width = w; out.height = h;
    out.rgb.resize((size_t)w * h * 3);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        unsigned char r, g, b;
        const unsigned char* px = &img[i * ch];
        if (ctype == 0 || ctype == 4) r = g = b = px[0];
        else if (ctype == 3) {
            if ((size_t)px[0] * 3 + 2 >= plte.size()) { err = "Format error decoding Png: palette index out of range"; return false; }
            r = plte[px[0] * 3]; g = plte[px[0] * 3 + 1]; b = plte[px[0] * 3 + 2];
        } else { r = px[0]; g = px[1]; b = px[2]; }
        out.rgb[3 * i] = (float)r / 256.0f; out.rgb[3 * i + 1] = (float)g / 256.0f; out.rgb[3 * i + 2] = (float)b / 256.0f;
    }
    return true;
}

}  // namespace mi355rt
