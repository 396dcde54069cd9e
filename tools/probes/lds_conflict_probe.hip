// What SQ_LDS_BANK_CONFLICT counts on gfx950 for accesses that are conflict-free by layout: every kernel below touches LDS as
// [k][lane] -- consecutive lanes on consecutive words (or 8-byte / 16-byte groups) -- in one access width, the widths the pool engine
// (csrc/trace_pool.hpp) uses: 4-byte reads / writes, 8-byte atomics on the class sets, 16-byte reads / writes of the ray groups,
// ds_bpermute.  Under rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS the ratio per kernel says how many "conflict"
// cycles an instruction of that width carries by construction (a 64-lane b128 access is 1 KB: several passes over the banks).
//   hipcc -O3 --offload-arch=gfx950 -o lds_conflict_probe lds_conflict_probe.hip
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d out -o p --output-format csv -- ./lds_conflict_probe
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int KT = 20, ITERS = 2000;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(1024, 1) k_read_b32(unsigned* sink)
{
    __shared__ unsigned a[KT * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < KT * 64; i += 1024) a[i] = i;
    __syncthreads();
    unsigned acc = 0;
    int k = (threadIdx.x >> 6) % KT;
    for (int it = 0; it < ITERS; it++) { acc += *(volatile unsigned*)&a[k * 64 + lane]; k = k + 1 == KT ? 0 : k + 1; }
    if (acc == 0x12345u) sink[0] = acc;
}
__global__ void __launch_bounds__(1024, 1) k_write_b32(unsigned* sink)
{
    __shared__ unsigned a[KT * 64];
    const int lane = threadIdx.x & 63;
    int k = (threadIdx.x >> 6) % KT;
    for (int it = 0; it < ITERS; it++) { *(volatile unsigned*)&a[k * 64 + lane] = it; k = k + 1 == KT ? 0 : k + 1; }
    __syncthreads();
    if (a[threadIdx.x & 1023] == 0x12345u) sink[0] = 1;
}
__global__ void __launch_bounds__(1024, 1) k_read_b64(unsigned* sink)
{
    __shared__ unsigned long long a[KT * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < KT * 64; i += 1024) a[i] = i;
    __syncthreads();
    unsigned long long acc = 0;
    int k = (threadIdx.x >> 6) % KT;
    for (int it = 0; it < ITERS; it++) { acc += *(volatile unsigned long long*)&a[k * 64 + lane]; k = k + 1 == KT ? 0 : k + 1; }
    if (acc == 0x12345u) sink[0] = (unsigned)acc;
}
__global__ void __launch_bounds__(1024, 1) k_atomic_b64(unsigned* sink)
{
    __shared__ unsigned long long a[2 * 64];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 128) a[threadIdx.x] = ~0ull;
    __syncthreads();
    unsigned long long acc = 0;
    for (int it = 0; it < ITERS; it++) {
        acc += __hip_atomic_fetch_and(&a[(it & 1) * 64 + lane], ~(1ull << (it & 63)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_or(&a[(it & 1) * 64 + lane], 1ull << (it & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (acc == 0x12345u) sink[0] = (unsigned)acc;
}
__global__ void __launch_bounds__(1024, 1) k_read_b128(unsigned* sink)
{
    __shared__ uint4 a[KT * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < KT * 64; i += 1024) a[i] = make_uint4(i, i, i, i);
    __syncthreads();
    unsigned acc = 0;
    int k = (threadIdx.x >> 6) % KT;
    for (int it = 0; it < ITERS; it++) {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)((k * 64 + lane) * 16)) : "memory");
        acc += v.x + v.w; k = k + 1 == KT ? 0 : k + 1;
    }
    if (acc == 0x12345u) sink[0] = acc + a[0].x;
}
__global__ void __launch_bounds__(1024, 1) k_write_b128(unsigned* sink)
{
    __shared__ uint4 a[KT * 64];
    const int lane = threadIdx.x & 63;
    int k = (threadIdx.x >> 6) % KT;
    for (int it = 0; it < ITERS; it++) {
        const u32x4 v = {(unsigned)it, (unsigned)lane, (unsigned)k, 7u};
        asm volatile("ds_write_b128 %0, %1" : : "v"((unsigned)((k * 64 + lane) * 16)), "v"(v) : "memory");
        k = k + 1 == KT ? 0 : k + 1;
    }
    __syncthreads();
    if (a[threadIdx.x & 1023].x == 0x12345u) sink[0] = 1;
}
__global__ void __launch_bounds__(1024, 1) k_bpermute(unsigned* sink)
{
    const int lane = threadIdx.x & 63;
    unsigned acc = lane;
    for (int it = 0; it < ITERS; it++) acc = (unsigned)__builtin_amdgcn_ds_bpermute(((lane * 5 + it) & 63) << 2, (int)acc) + 1u;
    if (acc == 0x12345u) sink[0] = acc;
}

// the same widths with a different k per lane (what a step of the pool engine does: every lane has claimed a slot of its own)
__global__ void __launch_bounds__(1024, 1) k_read_b32_mixed(unsigned* sink)
{
    __shared__ unsigned a[KT * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < KT * 64; i += 1024) a[i] = i;
    __syncthreads();
    unsigned acc = 0, h = threadIdx.x * 2654435761u;
    for (int it = 0; it < ITERS; it++) { h = h * 1664525u + 1013904223u; const int k = (h >> 16) % KT; acc += *(volatile unsigned*)&a[k * 64 + lane]; }
    if (acc == 0x12345u) sink[0] = acc;
}
__global__ void __launch_bounds__(1024, 1) k_read_b64_mixed(unsigned* sink)
{
    __shared__ unsigned long long a[KT * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < KT * 64; i += 1024) a[i] = i;
    __syncthreads();
    unsigned long long acc = 0;
    unsigned h = threadIdx.x * 2654435761u;
    for (int it = 0; it < ITERS; it++) { h = h * 1664525u + 1013904223u; const int k = (h >> 16) % KT; acc += *(volatile unsigned long long*)&a[k * 64 + lane]; }
    if (acc == 0x12345u) sink[0] = (unsigned)acc;
}
__global__ void __launch_bounds__(1024, 1) k_read_b128_mixed(unsigned* sink)
{
    __shared__ uint4 a[KT * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < KT * 64; i += 1024) a[i] = make_uint4(i, i, i, i);
    __syncthreads();
    unsigned acc = 0, h = threadIdx.x * 2654435761u;
    for (int it = 0; it < ITERS; it++) {
        h = h * 1664525u + 1013904223u;
        const int k = (h >> 16) % KT;
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)((k * 64 + lane) * 16)) : "memory");
        acc += v.x + v.w;
    }
    if (acc == 0x12345u) sink[0] = acc + a[0].x;
}
__global__ void __launch_bounds__(1024, 1) k_write_b128_mixed(unsigned* sink)
{
    __shared__ uint4 a[KT * 64];
    const int lane = threadIdx.x & 63;
    unsigned h = threadIdx.x * 2654435761u;
    for (int it = 0; it < ITERS; it++) {
        h = h * 1664525u + 1013904223u;
        const int k = (h >> 16) % KT;
        const u32x4 v = {(unsigned)it, (unsigned)lane, (unsigned)k, 7u};
        asm volatile("ds_write_b128 %0, %1" : : "v"((unsigned)((k * 64 + lane) * 16)), "v"(v) : "memory");
    }
    __syncthreads();
    if (a[threadIdx.x & 1023].x == 0x12345u) sink[0] = 1;
}

// a 16-byte group per lane read as two 8-byte halves (ds_read2_b64 offset1:1 -- what the compiler makes of some struct loads)
__global__ void __launch_bounds__(1024, 1) k_read2_b64_of_b128(unsigned* sink)
{
    __shared__ uint4 a[KT * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < KT * 64; i += 1024) a[i] = make_uint4(i, i, i, i);
    __syncthreads();
    unsigned acc = 0;
    int k = (threadIdx.x >> 6) % KT;
    for (int it = 0; it < ITERS; it++) {
        u32x4 v;
        asm volatile("ds_read2_b64 %0, %1 offset1:1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)((k * 64 + lane) * 16)) : "memory");
        acc += v.x + v.w; k = k + 1 == KT ? 0 : k + 1;
    }
    if (acc == 0x12345u) sink[0] = acc + a[0].x;
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    unsigned* sink; hipMalloc(&sink, 64);
    const dim3 g(p.multiProcessorCount), b(1024);
    hipLaunchKernelGGL(k_read_b32, g, b, 0, 0, sink);
    hipLaunchKernelGGL(k_write_b32, g, b, 0, 0, sink);
    hipLaunchKernelGGL(k_read_b64, g, b, 0, 0, sink);
    hipLaunchKernelGGL(k_atomic_b64, g, b, 0, 0, sink);
    hipLaunchKernelGGL(k_read_b128, g, b, 0, 0, sink);
    hipLaunchKernelGGL(k_write_b128, g, b, 0, 0, sink);
    hipLaunchKernelGGL(k_bpermute, g, b, 0, 0, sink);
    hipLaunchKernelGGL(k_read_b32_mixed, g, b, 0, 0, sink);
    hipLaunchKernelGGL(k_read_b64_mixed, g, b, 0, 0, sink);
    hipLaunchKernelGGL(k_read_b128_mixed, g, b, 0, 0, sink);
    hipLaunchKernelGGL(k_write_b128_mixed, g, b, 0, 0, sink);
    hipLaunchKernelGGL(k_read2_b64_of_b128, g, b, 0, 0, sink);
    hipDeviceSynchronize();
    printf("lds_conflict_probe: %d blocks x 1024 threads x %d accesses per kernel: %s\n", p.multiProcessorCount, ITERS, hipGetErrorString(hipGetLastError()));
    hipFree(sink);
    return 0;
}
