// Placeholder until the in-tree JPEG decoder lands: textures are read from a pre-decoded "<name>.ppm"
// raster next to the JPEG (see scene_loader.cpp: load_texture).
#include "jpeg_decoder.hpp"

namespace mcpt {
bool decode_jpeg_memory(const uint8_t*, size_t, int&, int&, std::vector<uint8_t>&, std::string& err)
{
    err = "JPEG decoding not built in; provide <texture>.ppm";
    return false;
}
bool decode_jpeg_file(const std::string&, int&, int&, std::vector<uint8_t>&, std::string& err)
{
    err = "JPEG decoding not built in; provide <texture>.ppm";
    return false;
}
}  // namespace mcpt
