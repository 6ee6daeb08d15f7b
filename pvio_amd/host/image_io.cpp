// image_io.cpp -- see image_io.h.
#include "image_io.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace pvio {

namespace {

inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

} // namespace

GrayImage decode_png_gray(const uint8_t *data, size_t size) {
    static const uint8_t magic[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (size < 8 || std::memcmp(data, magic, 8) != 0) throw std::runtime_error("png: bad signature");
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    bool have_ihdr = false, done = false;
    std::vector<uint8_t> idat;
    while (!done && pos + 12 <= size) {
        const uint32_t len = be32(data + pos);
        const uint8_t *type = data + pos + 4, *body = data + pos + 8;
        if (pos + 12 + (size_t)len > size) throw std::runtime_error("png: truncated chunk");
        if (crc32(crc32(0L, Z_NULL, 0), type, len + 4) != be32(body + len)) throw std::runtime_error("png: chunk CRC mismatch");
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) throw std::runtime_error("png: bad IHDR");
            w = be32(body), h = be32(body + 4), depth = body[8], ctype = body[9], interlace = body[12];
            have_ihdr = true;
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            done = true;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || !done || w == 0 || h == 0 || w > (1u << 15) || h > (1u << 15)) throw std::runtime_error("png: missing IHDR / IEND or bad size");
    if (interlace != 0) throw std::runtime_error("png: interlaced files are not supported");
    if (depth != 8 && depth != 16) throw std::runtime_error("png: only bit depth 8 and 16 are supported");
    int channels;
    switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: throw std::runtime_error("png: unsupported colour type (palette?)");
    }
    const size_t bpp = (size_t)channels * depth / 8, row = bpp * w;
    // the header is untrusted: deflate expands at most ~1032 : 1, so a stream this short cannot hold an image this large
    // (refused before the allocation instead of failing in it with bad_alloc)
    if (w == 0 || h == 0 || (row + 1) * (double)h > 1032.0 * (double)idat.size() + 64.0) throw std::runtime_error("png: image size does not match the compressed data");
    std::vector<uint8_t> raw((row + 1) * h);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) throw std::runtime_error("png: inflate failed");
    // undo the scanline filters in place (PNG specification, section 9)
    std::vector<uint8_t> zero(row, 0);
    for (uint32_t y = 0; y < h; ++y) {
        uint8_t *cur = raw.data() + (size_t)y * (row + 1) + 1;
        const uint8_t *up = y ? cur - (row + 1) : zero.data();
        const int ft = cur[-1];
        // one loop per filter type (the branch is per scanline, not per byte); bytes left of the first pixel predict from 0
        switch (ft) {
        case 0: break;
        case 1:
            for (size_t i = bpp; i < row; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
            break;
        case 2:
            for (size_t i = 0; i < row; ++i) cur[i] = (uint8_t)(cur[i] + up[i]);
            break;
        case 3:
            for (size_t i = 0; i < bpp && i < row; ++i) cur[i] = (uint8_t)(cur[i] + (up[i] >> 1));
            for (size_t i = bpp; i < row; ++i) cur[i] = (uint8_t)(cur[i] + ((cur[i - bpp] + up[i]) >> 1));
            break;
        case 4:
            for (size_t i = 0; i < bpp && i < row; ++i) cur[i] = (uint8_t)(cur[i] + paeth(0, up[i], 0));
            for (size_t i = bpp; i < row; ++i) cur[i] = (uint8_t)(cur[i] + paeth(cur[i - bpp], up[i], up[i - bpp]));
            break;
        default: throw std::runtime_error("png: bad filter type");
        }
    }
    GrayImage img;
    img.width = (int)w, img.height = (int)h;
    img.pixels.resize((size_t)w * h);
    const size_t sb = depth / 8; // bytes per sample; the first one is the high byte
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t *cur = raw.data() + (size_t)y * (row + 1) + 1;
        for (uint32_t x = 0; x < w; ++x) {
            const uint8_t *px = cur + (size_t)x * bpp;
            if (channels <= 2) {
                img.pixels[(size_t)y * w + x] = px[0];
            } else {
                const uint32_t r = px[0], g = px[sb], b = px[2 * sb];
                img.pixels[(size_t)y * w + x] = (r == g && g == b) ? (uint8_t)r : (uint8_t)((9798u * r + 19235u * g + 3735u * b + 16384u) >> 15);
            }
        }
    }
    return img;
}

GrayImage read_gray_image(const std::string &filename) {
    FILE *f = std::fopen(filename.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open image " + filename);
    std::vector<uint8_t> buf;
    uint8_t tmp[65536];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    std::fclose(f);
    if (buf.size() >= 8 && buf[0] == 0x89 && buf[1] == 'P') return decode_png_gray(buf.data(), buf.size());
    if (buf.size() >= 2 && buf[0] == 'P' && buf[1] == '5') {
        // P5 <ws> width <ws> height <ws> maxval <single ws> raster; '#' comments run to the end of the line
        size_t pos = 2;
        auto next_int = [&]() {
            for (;;) {
                while (pos < buf.size() && (buf[pos] == ' ' || buf[pos] == '\t' || buf[pos] == '\n' || buf[pos] == '\r')) ++pos;
                if (pos < buf.size() && buf[pos] == '#') {
                    while (pos < buf.size() && buf[pos] != '\n') ++pos;
                    continue;
                }
                break;
            }
            long v = 0;
            bool any = false;
            while (pos < buf.size() && buf[pos] >= '0' && buf[pos] <= '9') v = v * 10 + (buf[pos++] - '0'), any = true;
            if (!any) throw std::runtime_error("pgm: bad header in " + filename);
            return v;
        };
        const long w = next_int(), h = next_int(), maxval = next_int();
        ++pos;
        if (w < 1 || h < 1 || maxval < 1 || maxval > 255 || pos + (size_t)w * h > buf.size()) throw std::runtime_error("pgm: unsupported / truncated " + filename);
        GrayImage img;
        img.width = (int)w, img.height = (int)h;
        img.pixels.assign(buf.begin() + (long)pos, buf.begin() + (long)(pos + (size_t)w * h));
        return img;
    }
    throw std::runtime_error("unsupported image format: " + filename);
}

} // namespace pvio
