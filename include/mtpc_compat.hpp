// C++ drop-in for the reference's entry point
//     bool render_scene(std::string path, std::string filename, int N_ray_per_pixel)      MTPC/MTPC.cpp:35-68
// Same argument meaning, same files read (<path><filename>.obj/.mtl/.camera) and written
// ("../result/<filename>-SPP<N>.png"), same progress prints; the work is done by libmcpt's HIP kernels.
// The reference always returns true; this returns false when the scene cannot be read or no MI355X is present
// (mcpt_last_error() says why).
#pragma once
#include <string>

#include "mcpt.h"

namespace mtpc {
// knobs the reference keeps as literals or time(NULL): RNG seed, device, resolution override, verbosity
inline mcpt_render_scene_options& options()
{
    static mcpt_render_scene_options o{};
    return o;
}
}  // namespace mtpc

inline bool render_scene(std::string path, std::string filename, int N_ray_per_pixel)
{
    return mcpt_render_scene_opts(path.c_str(), filename.c_str(), N_ray_per_pixel, &mtpc::options(), sizeof(mcpt_render_scene_options), nullptr) == MCPT_OK;
}
