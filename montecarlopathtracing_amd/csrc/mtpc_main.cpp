// Command-line stand-in for the reference's main() (MTPC/MTPC.cpp:71-91), which hard-codes the scene name
// and the sample count: mtpc [path] [filename] [spp] [seed] [width height].
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/mtpc_compat.hpp"

int main(int argc, char** argv)
{
    const std::string path = argc > 1 ? argv[1] : "../scene/";
    const std::string filename = argc > 2 ? argv[2] : "cornell-box";
    const int spp = argc > 3 ? std::atoi(argv[3]) : 25;
    if (argc > 4) mtpc::options().seed = std::strtoull(argv[4], nullptr, 10);
    if (argc > 6) { mtpc::options().width = std::atoi(argv[5]); mtpc::options().height = std::atoi(argv[6]); }
    if (!render_scene(path, filename, spp)) {
        std::fprintf(stderr, "render_scene failed: %s\n", mcpt_last_error());
        return 1;
    }
    return 0;
}
