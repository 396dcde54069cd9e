// What the vector L1 (TCP) of an MI355X CU sustains for the trace kernel's access shape: every lane of a wave reads 16 bytes
// (global_load_dwordx4) from its own record of a small table (L2-resident, mostly L1-resident), 4 loads per 64-B record like a
// CwNode or 1 load per record, dependent on the previous record (a walk) -- lane-loads per cycle per CU against waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o gather_probe gather_probe.hip && ./gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

template <int LOADS>
__global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ table, unsigned mask, int steps, unsigned* __restrict__ sink)
{
    unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    unsigned acc = 0;
    for (int s = 0; s < steps; s++) {
        const uint4* rec = table + (size_t)(idx & mask) * 4;        // 64-B record
        uint4 a = rec[0];
        acc += a.x;
        unsigned next = a.y;
        if (LOADS >= 2) { uint4 b = rec[1]; acc += b.x; next ^= b.z; }
        if (LOADS >= 4) { uint4 c = rec[2], d = rec[3]; acc += c.x + d.w; next ^= c.y ^ d.z; }
        idx = next * 2654435761u + s;                               // the next record depends on what was loaded
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    int dev = 0; hipDeviceProp_t p; hipGetDeviceProperties(&p, dev);
    const int cus = p.multiProcessorCount;
    for (unsigned records : {1u << 12, 1u << 15, 1u << 18}) {       // 256 KB, 2 MB, 16 MB of 64-B records
        std::vector<uint4> h((size_t)records * 4);
        uint32_t x = 12345;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v.x = x; x = x * 1664525u + 1013904223u; v.y = x; x = x * 1664525u + 1013904223u; v.z = x; v.w = x >> 7; }
        uint4* d; unsigned* sink;
        hipMalloc(&d, h.size() * sizeof(uint4)); hipMalloc(&sink, 64);
        hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice);
        for (int blocks_per_cu : {1, 2, 3, 4, 6, 8}) {
            for (int loads : {1, 4}) {
                const int steps = 4000;
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                auto run = [&]() {
                    if (loads == 1) hipLaunchKernelGGL(k_gather<1>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, d, records - 1, steps, sink);
                    else hipLaunchKernelGGL(k_gather<4>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, d, records - 1, steps, sink);
                };
                run(); hipDeviceSynchronize();
                hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                const double lane_loads = (double)cus * blocks_per_cu * 256 * steps * loads;
                const double cycles = ms * 1e-3 * 2.4e9;
                printf("table %6u KB  waves/SIMD %d  loads/record %d : %.3f ms  %.3f lane-loads per cycle per CU  (%.2f T lane-loads/s, %.1f ns per dependent step)\n",
                       records * 64 / 1024, blocks_per_cu, loads, ms, lane_loads / cycles / cus, lane_loads / (ms * 1e-3) / 1e12, ms * 1e6 / steps);
            }
        }
        hipFree(d); hipFree(sink);
    }
    return 0;
}
