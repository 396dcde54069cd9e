// Texture input for map_Kd (the reference calls cv::imread, MTPC/sceneManagement.h:137).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mcpt {
// Decodes a baseline or progressive JFIF file into an 8-bit BGR raster (OpenCV's cv::Mat layout).
bool decode_jpeg_file(const std::string& file, int& width, int& height, std::vector<uint8_t>& bgr, std::string& err);
bool decode_jpeg_memory(const uint8_t* data, size_t size, int& width, int& height, std::vector<uint8_t>& bgr, std::string& err);
}  // namespace mcpt
