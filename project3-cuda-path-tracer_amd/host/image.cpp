// image.cpp -- PNG writer for the headless driver.  The reference hands the bytes to stb_image_write
// (src/image.cpp:35); the pixel conversion is reproduced exactly, the container is written here with
// stored (uncompressed) deflate blocks, so no third-party codec is needed.
#include "image.h"

#include <cassert>
#include <cstdint>
#include <cmath>
#include <cstdio>
#include <iostream>

image::image(int x, int y) : xSize(x), ySize(y), pixels((size_t)x * y) {}

void image::setPixel(int x, int y, const lin::vec3 &pixel) {
    assert(x >= 0 && y >= 0 && x < xSize && y < ySize);
    pixels[(size_t)y * xSize + x] = pixel;
}

std::vector<unsigned char> image::toBytes() const {
    std::vector<unsigned char> bytes((size_t)3 * xSize * ySize);
    for (size_t i = 0; i < pixels.size(); ++i) {
        const float c[3] = {pixels[i].x, pixels[i].y, pixels[i].z};
        for (int k = 0; k < 3; ++k) {
            float v = c[k] < 0.0f ? 0.0f : c[k];   // glm::clamp = min(max(x, 0), 1)
            v = v > 1.0f ? 1.0f : v;
            bytes[3 * i + k] = (unsigned char)(v * 255.f);
        }
    }
    return bytes;
}

namespace {
uint32_t crc_table[256];
void crc_init() {
    for (uint32_t n = 0; n < 256; ++n) {
        uint32_t c = n;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        crc_table[n] = c;
    }
}
uint32_t crc32(uint32_t crc, const unsigned char *p, size_t n) {
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = crc_table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
    return ~crc;
}
void be32(std::vector<unsigned char> &v, uint32_t x) {
    v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}
void chunk(std::vector<unsigned char> &png, const char *tag, const std::vector<unsigned char> &data) {
    be32(png, (uint32_t)data.size());
    const size_t start = png.size();
    png.insert(png.end(), tag, tag + 4);
    png.insert(png.end(), data.begin(), data.end());
    be32(png, crc32(0, &png[start], png.size() - start));
}
}  // namespace

bool image::savePNG(const std::string &baseFilename) {
    const std::vector<unsigned char> rgb = toBytes();
    crc_init();
    std::vector<unsigned char> raw;  // filter byte 0 + scanline
    raw.reserve((size_t)ySize * (3 * xSize + 1));
    for (int y = 0; y < ySize; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb.begin() + (size_t)y * 3 * xSize, rgb.begin() + (size_t)(y + 1) * 3 * xSize);
    }
    std::vector<unsigned char> z;    // zlib stream, stored blocks
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw.size() || off == 0;) {
        const size_t n = raw.size() - off < 65535 ? raw.size() - off : 65535;
        z.push_back(off + n >= raw.size() ? 1 : 0);
        z.push_back(n & 0xff); z.push_back(n >> 8);
        z.push_back(~n & 0xff); z.push_back((~n >> 8) & 0xff);
        for (size_t i = 0; i < n; ++i) {
            a = (a + raw[off + i]) % 65521u;
            b = (b + a) % 65521u;
        }
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        off += n;
        if (n == 0) break;
    }
    be32(z, (b << 16) | a);

    std::vector<unsigned char> png = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<unsigned char> ihdr;
    be32(ihdr, (uint32_t)xSize);
    be32(ihdr, (uint32_t)ySize);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(png, "IHDR", ihdr);
    chunk(png, "IDAT", z);
    chunk(png, "IEND", std::vector<unsigned char>());

    const std::string filename = baseFilename + ".png";
    FILE *f = fopen(filename.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(png.data(), 1, png.size(), f) == png.size();
    fclose(f);
    if (ok) std::cout << "Saved " << filename << "." << std::endl;
    return ok;
}

bool image::saveHDR(const std::string &baseFilename) {
    const std::string filename = baseFilename + ".hdr";
    FILE *f = fopen(filename.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n", ySize, xSize);
    std::vector<unsigned char> row((size_t)4 * xSize);
    bool ok = true;
    for (int y = 0; y < ySize && ok; ++y) {
        for (int x = 0; x < xSize; ++x) {
            const lin::vec3 &p = pixels[(size_t)y * xSize + x];
            const float m = std::fmax(p.x, std::fmax(p.y, p.z));
            unsigned char *o = &row[(size_t)4 * x];
            if (!(m >= 1e-32f)) {
                o[0] = o[1] = o[2] = o[3] = 0;
            } else {                     // shared exponent: value = mantissa/256 * 2^(e-128)
                int e;
                const float scale = std::frexp(m, &e) * 256.0f / m;
                o[0] = (unsigned char)(p.x * scale);
                o[1] = (unsigned char)(p.y * scale);
                o[2] = (unsigned char)(p.z * scale);
                o[3] = (unsigned char)(e + 128);
            }
        }
        ok = fwrite(row.data(), 1, row.size(), f) == row.size();
    }
    fclose(f);
    if (ok) std::cout << "Saved " + filename + "." << std::endl;
    return ok;
}
