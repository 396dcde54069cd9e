// The one place libmcpt.so looks at the environment (knobs.hpp).
#include "knobs.hpp"

#include <cstdlib>
#include <cstring>
#include <string>

namespace mcpt {

namespace {

const char* env(const char* name) { return std::getenv(name); }

long long env_ll(const char* name, long long dflt, long long lo, long long hi)
{
    const char* e = env(name);
    if (!e || !*e) return dflt;
    const long long v = std::atoll(e);
    return (v < lo || v > hi) ? dflt : v;
}
double env_d(const char* name, double dflt, double lo, double hi)
{
    const char* e = env(name);
    if (!e || !*e) return dflt;
    const double v = std::atof(e);
    return (!(v >= lo) || !(v <= hi)) ? dflt : v;
}

struct Row { const char* name; const char* dflt; const char* doc; };
// (kept in the order of struct Knobs)
const Row kRows[] = {
    {"MCPT_TRACE_ENGINE", "by scene size", "vote | pool: the closest-hit engine of the fast walk (voting engine: one ray per lane; pool engine: a workgroup's rays resident in LDS)"},
    {"MCPT_POOL_MAX_TRIS", "131072", "largest scene (triangles) the pool engine is picked for"},
    {"MCPT_FINISH_ENGINE", "pool where the pool engine runs", "lane: the one-lane-per-path finishing kernel instead of the pool engine in path mode (A/B runs)"},
    {"MCPT_FINISH_PATHS", "1500000 (pool form) / 500000 (lane form)", "paths left at which the finishing pass takes a chunk over; 0: never"},
    {"MCPT_PRE_TEST_MAX_TRIS", "1048576", "largest scene that gets the fp32 pre-test records of its leaf triangles"},
    {"MCPT_SHORT_KERNEL", "1", "0: the voting engine's deep-stack form (36 entries, 3 waves per SIMD) instead of the 27-entry form at 4"},
    {"MCPT_LOGIC_GRID", "resident size", "blocks of the logic kernel's grid"},
    {"MCPT_TRACE_BLOCK_RAYS", "2048", "a block of the trace engines is started per this many rays"},
    {"MCPT_TRACE_MIN_CHUNK", "256", "ray slots per queue claim, lower bound"},
    {"MCPT_TRACE_MAX_CHUNK", "2048", "ray slots per queue claim, upper bound"},
    {"MCPT_WORKSPACE_GB", "a share of the free HBM", "path-state workspace per frame slot, GiB"},
    {"MCPT_FAST_STACK_LIMIT", "36", "stack entries the culling hierarchy is built to need at most (8..36)"},
    {"MCPT_FAST_LEAF", "4", "most triangles in a leaf of the host-built hierarchy (1..8)"},
    {"MCPT_FAST_CT", "1.6", "SAH cost of a leaf triangle relative to a node"},
    {"MCPT_BUILD_SERIAL", "0", "1: the host SAH builder on one thread (the threaded build gives the same tree)"},
    {"MCPT_NODE_CACHE", "engine default", "nodes of the top of the tree the engines may mirror in LDS"},
    {"MCPT_CLUSTER_LEAF", "1", "MCPT_BUILD_DEVICE_FAST: triangles per leaf of a Morton cluster (1..8)"},
    {"MCPT_CLUSTER_LEVELS", "1", "MCPT_BUILD_DEVICE_FAST: levels of 4-wide nodes built on the GPU (1..5)"},
    {"MCPT_PLOC_CLUSTER", "4096", "MCPT_BUILD_DEVICE_SAH: most triangles in a cluster grown on the GPU"},
    {"MCPT_PLOC_HEIGHT", "from the triangle count", "... tallest cluster (3..24)"},
    {"MCPT_PLOC_RADIUS", "8", "... neighbours looked at on either side of the Morton order (1..64)"},
    {"MCPT_PLOC_LEAF", "4", "... most triangles in a leaf (1..8)"},
    {"MCPT_PLOC_BUDGET", "from the height", "... stack entries a cluster's subtree may need (3..30)"},
    {"MCPT_PLOC_AREA", "16", "... a merge may not exceed 1/this of the scene box's area (0: no bound)"},
    {"MCPT_PLOC_CT", "1.0", "... SAH cost of a triangle"},
    {"MCPT_PLOC_CL", "0.0", "... SAH cost of a leaf"},
    {"MCPT_SLOW_LIST", "1048576", "entries of the deferred-ray list (tests shrink it to force the overflow path)"},
    {"MCPT_TEST_STACK_CAP", "off", "stack entries the trace engines may use (tests force the hand-over to the one-lane walk)"},
    {"MCPT_PRINT_DIAG", "0", "1: phase times and in-kernel counters on stderr"},
    {"MCPT_ALLOW_RUNTIME_MISMATCH", "0", "1: run on a HIP runtime of another release than the one libmcpt.so was compiled against (mcpt_allow_runtime_mismatch)"},
};

std::string make_table()
{
    std::string t;
    for (const Row& r : kRows) { t += r.name; t += " | "; t += r.dflt; t += " | "; t += r.doc; t += "\n"; }
    return t;
}

}  // namespace

Knobs read_knobs()
{
    Knobs k;
    if (const char* e = env("MCPT_TRACE_ENGINE")) k.trace_engine = std::strcmp(e, "pool") == 0 ? 1 : (std::strcmp(e, "vote") == 0 ? 0 : -1);
    k.pool_max_tris = env_ll("MCPT_POOL_MAX_TRIS", k.pool_max_tris, 0, 1ll << 40);
    if (const char* e = env("MCPT_FINISH_ENGINE")) k.finish_engine = std::strcmp(e, "lane") == 0 ? 0 : -1;
    k.finish_paths = env_ll("MCPT_FINISH_PATHS", -1, 0, 1ll << 40);
    k.pre_test_max_tris = env_ll("MCPT_PRE_TEST_MAX_TRIS", k.pre_test_max_tris, 0, 1ll << 40);
    k.short_kernel = (int)env_ll("MCPT_SHORT_KERNEL", 1, 0, 1);
    k.logic_grid = (unsigned)env_ll("MCPT_LOGIC_GRID", 0, 1, 1 << 20);
    k.trace_block_rays = env_ll("MCPT_TRACE_BLOCK_RAYS", 2048, 256, 1ll << 30);
    k.trace_min_chunk = (int)env_ll("MCPT_TRACE_MIN_CHUNK", 256, 64, 1 << 24) / 64 * 64;
    k.trace_max_chunk = (int)env_ll("MCPT_TRACE_MAX_CHUNK", 2048, 64, 1 << 24) / 64 * 64;
    if (k.trace_max_chunk < k.trace_min_chunk) k.trace_max_chunk = k.trace_min_chunk;
    k.workspace_gb = env_d("MCPT_WORKSPACE_GB", 0.0, 0.0100001, 1e6);
    k.fast_stack_limit = (int)env_ll("MCPT_FAST_STACK_LIMIT", 0, 8, 36);
    k.fast_leaf = (int)env_ll("MCPT_FAST_LEAF", 0, 1, 1 << 20);
    k.fast_ct = env_d("MCPT_FAST_CT", 0.0, 1e-9, 1e9);
    k.build_serial = env("MCPT_BUILD_SERIAL") != nullptr;
    k.node_cache = (int)env_ll("MCPT_NODE_CACHE", -1, 0, 1 << 20);
    k.cluster_leaf = (int)env_ll("MCPT_CLUSTER_LEAF", 1, 1, 8);
    k.cluster_levels = (int)env_ll("MCPT_CLUSTER_LEVELS", 1, 1, 5);
    k.ploc_cluster = (int)env_ll("MCPT_PLOC_CLUSTER", 4096, 4, 65536);
    k.ploc_height = (int)env_ll("MCPT_PLOC_HEIGHT", 0, 3, 24);
    k.ploc_radius = (int)env_ll("MCPT_PLOC_RADIUS", 8, 1, 64);
    k.ploc_leaf = (int)env_ll("MCPT_PLOC_LEAF", 0, 1, 8);
    k.ploc_budget = (int)env_ll("MCPT_PLOC_BUDGET", 0, 3, 30);
    k.ploc_area = env_d("MCPT_PLOC_AREA", 16.0, 0.0, 1e30);
    k.ploc_ct = env_d("MCPT_PLOC_CT", 1.0, 1e-30, 1e30);
    k.ploc_cl = env_d("MCPT_PLOC_CL", 0.0, -1e30, 1e30);
    k.slow_list = env_ll("MCPT_SLOW_LIST", 0, 1, 1ll << 31);
    k.test_stack_cap = (int)env_ll("MCPT_TEST_STACK_CAP", 0, 4, 1 << 20);
    k.print_diag = env("MCPT_PRINT_DIAG") != nullptr;
    k.allow_runtime_mismatch = (int)env_ll("MCPT_ALLOW_RUNTIME_MISMATCH", 0, 0, 1 << 30) != 0;
    return k;
}

const char* knobs_table()
{
    static const std::string t = make_table();
    return t.c_str();
}

}  // namespace mcpt
