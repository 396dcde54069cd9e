// Every environment variable libmcpt.so reads, in one place.  The library's API carries handles, and a handle carries its tuning: the
// environment is parsed when a handle is created (mcpt_device_create / mcpt_multi_create read it into the device; a scene's culling
// hierarchy is built with the knobs of the device creation that triggers the build) -- never inside a launch, never into a
// function-local static, so two devices created under different settings keep them.  knobs_table() is the documentation
// (mcpt_knobs_describe; INTEGRATION.md section 7 is checked against it by a test); nothing else in the library calls getenv.
#pragma once
#include <cstdint>

namespace mcpt {

struct Knobs {
    // ---- which engines run
    int trace_engine = -1;              // MCPT_TRACE_ENGINE: -1 by scene size, 0 vote, 1 pool
    long long pool_max_tris = 1ll << 17;   // MCPT_POOL_MAX_TRIS
    int finish_engine = -1;             // MCPT_FINISH_ENGINE: -1 pool form where the pool engine runs, 0 lane (one lane per path)
    long long finish_paths = -1;        // MCPT_FINISH_PATHS: -1 the engine's default
    long long pre_test_max_tris = 1ll << 20;   // MCPT_PRE_TEST_MAX_TRIS
    int short_kernel = 1;               // MCPT_SHORT_KERNEL
    // ---- launch shapes
    unsigned logic_grid = 0;            // MCPT_LOGIC_GRID (0: resident-size grid)
    long long trace_block_rays = 2048;  // MCPT_TRACE_BLOCK_RAYS
    int trace_min_chunk = 256, trace_max_chunk = 2048;   // MCPT_TRACE_MIN_CHUNK / MCPT_TRACE_MAX_CHUNK
    double workspace_gb = 0;            // MCPT_WORKSPACE_GB (0: a share of the free HBM)
    // ---- culling hierarchy (host builder)
    int fast_stack_limit = 0;           // MCPT_FAST_STACK_LIMIT (0: the deep stack)
    int fast_leaf = 0;                  // MCPT_FAST_LEAF (0: default leaf size)
    double fast_ct = 0;                 // MCPT_FAST_CT (0: default SAH cost of a triangle)
    int build_serial = 0;               // MCPT_BUILD_SERIAL
    int node_cache = -1;                // MCPT_NODE_CACHE (-1: the engines' own prefix of the top of the tree)
    // ---- culling hierarchy (device builders)
    int cluster_leaf = 1, cluster_levels = 1;   // MCPT_CLUSTER_LEAF / MCPT_CLUSTER_LEVELS (MCPT_BUILD_DEVICE_FAST)
    int ploc_cluster = 4096, ploc_height = 0, ploc_radius = 8, ploc_leaf = 0, ploc_budget = 0;   // MCPT_PLOC_* (MCPT_BUILD_DEVICE_SAH)
    double ploc_area = 16.0, ploc_ct = 1.0, ploc_cl = 0.0;
    // ---- tests and diagnostics
    long long slow_list = 0;            // MCPT_SLOW_LIST (0: 2^20 entries)
    int test_stack_cap = 0;             // MCPT_TEST_STACK_CAP
    int print_diag = 0;                 // MCPT_PRINT_DIAG
    int allow_runtime_mismatch = 0;     // MCPT_ALLOW_RUNTIME_MISMATCH
};

Knobs read_knobs();                     // the process environment, now
const char* knobs_table();              // one line per variable: name, default, meaning

}  // namespace mcpt
