// image.h -- float RGB image -> 8-bit PNG (reference src/image.h, src/image.cpp:17-39).
#pragma once
#include <string>
#include <vector>

#include "linalg.h"

class image {
private:
    int xSize;
    int ySize;
    std::vector<lin::vec3> pixels;

public:
    image(int x, int y);
    void setPixel(int x, int y, const lin::vec3 &pixel);
    // clamp to [0,1], x255, truncate to unsigned char (reference src/image.cpp:27-30); writes <base>.png
    bool savePNG(const std::string &baseFilename);
    // Radiance RGBE (.hdr) of the unclamped float pixels (reference src/image.cpp:41-45 via stbi_write_hdr);
    // writes <base>.hdr, flat (non-RLE) scanlines
    bool saveHDR(const std::string &baseFilename);
    // the 8-bit RGB bytes savePNG would encode (row-major, 3 bytes per pixel)
    std::vector<unsigned char> toBytes() const;
};
