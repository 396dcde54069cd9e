// TEST INFRASTRUCTURE (part of libmcpt_oracle.so).
// The reference sorts its faces with   sort(scene.f.begin(), scene.f.end(), compare)   (MTPC/MTPC.cpp:44), where
// compare(Face a, Face b) = a.morton_code < b.morton_code (MTPC/sceneManagement.cpp:311-314).  std::sort is not stable, so the
// order of faces that share a key (cornell-box: 1 194 faces, veach-mis: 1 416, up to 380 per key) is whatever the standard
// library's introsort leaves.  The sequence of comparisons and moves of std::sort depends only on the comparator's answers,
// not on the element type, so sorting (key, original index) records with the same comparator through THIS toolchain's
// std::sort gives the permutation a g++/libstdc++ build of the reference produces (the build SURVEY.md section 3.5 measured).
// The oracle's default remains the stable order (deviation D2); this order exists to show what D2 changes and what it does not.
#include <algorithm>
#include <cstdint>
#include <vector>

namespace {
struct Rec { uint32_t morton_code; int32_t index; };
bool compare(Rec a, Rec b) { return a.morton_code < b.morton_code; }       // by value, like the reference's
}

extern "C" void orc_std_sort_order(const uint32_t* keys, int n, int32_t* order)
{
    std::vector<Rec> v(static_cast<size_t>(n));
    for (int i = 0; i < n; i++) { v[i].morton_code = keys[i]; v[i].index = i; }
    std::sort(v.begin(), v.end(), compare);
    for (int i = 0; i < n; i++) order[i] = v[i].index;
}
