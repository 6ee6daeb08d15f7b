// image_io.h -- grayscale image decode for the dataset readers: the cv::imread(filename, cv::IMREAD_GRAYSCALE) of
// pvio-pc/src/euroc_dataset_reader.cpp:71 and tum_dataset_reader.cpp:70 without OpenCV / libpng (neither is installed here).
// PNG: non-interlaced, bit depth 8 or 16, colour types 0 (gray), 2 (RGB), 4 (gray + alpha), 6 (RGBA) -- what EuRoC (8-bit
// gray) and TUM-VI (8- or 16-bit gray) ship; inflate comes from zlib.  16-bit samples keep their high byte (what OpenCV's
// decoder asks libpng for, png_set_strip_16); colour is reduced the way libpng does for OpenCV's png_set_rgb_to_gray(0.299,
// 0.587): (9798 r + 19235 g + 3735 b + 16384) >> 15, r = g = b passing through (the datasets themselves are gray).
// PGM (P5, maxval <= 255) is read too.  Failures throw std::runtime_error: there is no silent fallback.
#pragma once
#include "host_namespace.h"
#include <cstdint>
#include <string>
#include <vector>

namespace pvio {

struct GrayImage {
    int width = 0, height = 0;
    std::vector<uint8_t> pixels; // [height][width]
};

GrayImage decode_png_gray(const uint8_t *data, size_t size);
GrayImage read_gray_image(const std::string &filename); // by magic: PNG or PGM

} // namespace pvio
