// What an MI355X sustains for the logic kernel's memory shape: every path position reads ~140 bytes and writes ~160 bytes of state as
// 8- and 4-byte words per lane, (a) component-major over the whole capacity ([component][cap]: every component its own stream, 2 GB
// apart -- the layout k_wf_logic uses) against (b) component-major inside tiles of 64 positions ([tile][component][64]: a wave's
// loads cover one contiguous 9-KB block).  Same bytes, same 512-/256-byte runs per wave instruction; a resident-size grid of blocks
// striding over the positions like k_wf_logic, 4 blocks per CU.
//   hipcc -O3 --offload-arch=gfx950 -o stream_layout_probe stream_layout_probe.hip && ./stream_layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int ND_IN = 15, NI_IN = 5, ND_OUT = 18, NI_OUT = 3;     // doubles / ints per position, read and written

template <bool TILED>
__global__ void __launch_bounds__(256, 4) k_stream(const char* __restrict__ in, char* __restrict__ out, long long n, long long cap, int extra_valu)
{
    constexpr long long kTileIn = 64ll * (ND_IN * 8 + NI_IN * 4), kTileOut = 64ll * (ND_OUT * 8 + NI_OUT * 4);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        double d[ND_IN]; int w[NI_IN];
        if (TILED) {
            const char* t = in + (i >> 6) * kTileIn;
            const int l = (int)(i & 63);
#pragma unroll
            for (int k = 0; k < ND_IN; k++) d[k] = reinterpret_cast<const double*>(t + k * 512)[l];
#pragma unroll
            for (int k = 0; k < NI_IN; k++) w[k] = reinterpret_cast<const int*>(t + ND_IN * 512 + k * 256)[l];
        } else {
#pragma unroll
            for (int k = 0; k < ND_IN; k++) d[k] = reinterpret_cast<const double*>(in)[k * cap + i];
#pragma unroll
            for (int k = 0; k < NI_IN; k++) w[k] = reinterpret_cast<const int*>(in + ND_IN * 8 * cap)[k * cap + i];
        }
        double s = 0; int ws = 0;
#pragma unroll
        for (int k = 0; k < ND_IN; k++) s += d[k];
#pragma unroll
        for (int k = 0; k < NI_IN; k++) ws += w[k];
        for (int k = 0; k < extra_valu; k++) s = s * 1.0000001 + 0.5;          // stand-in for the shading arithmetic between loads and stores
        if (TILED) {
            char* t = out + (i >> 6) * kTileOut;
            const int l = (int)(i & 63);
#pragma unroll
            for (int k = 0; k < ND_OUT; k++) reinterpret_cast<double*>(t + k * 512)[l] = s + k;
#pragma unroll
            for (int k = 0; k < NI_OUT; k++) reinterpret_cast<int*>(t + ND_OUT * 512 + k * 256)[l] = ws + k;
        } else {
#pragma unroll
            for (int k = 0; k < ND_OUT; k++) reinterpret_cast<double*>(out)[k * cap + i] = s + k;
#pragma unroll
            for (int k = 0; k < NI_OUT; k++) reinterpret_cast<int*>(out + ND_OUT * 8 * cap)[k * cap + i] = ws + k;
        }
    }
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const long long cap = 64ll << 20;                                 // 64 M positions: 9.4 GB in, 10.4 GB out
    const size_t in_bytes = (size_t)cap * (ND_IN * 8 + NI_IN * 4), out_bytes = (size_t)cap * (ND_OUT * 8 + NI_OUT * 4);
    char *in, *out;
    if (hipMalloc(&in, in_bytes) != hipSuccess || hipMalloc(&out, out_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(in, 0, in_bytes); hipMemset(out, 0, out_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%d CUs; per position %d B read, %d B written; %lld M positions\n", cus, ND_IN * 8 + NI_IN * 4, ND_OUT * 8 + NI_OUT * 4, cap >> 20);
    for (int extra : {0, 400, 1200}) {
        for (int tiled = 0; tiled < 2; tiled++) {
            for (long long n : {cap, cap / 8}) {
                float best = 1e30f;
                for (int rep = 0; rep < 4; rep++) {
                    hipEventRecord(e0);
                    if (tiled) hipLaunchKernelGGL(k_stream<true>, dim3(cus * 4), dim3(256), 0, 0, in, out, n, cap, extra);
                    else hipLaunchKernelGGL(k_stream<false>, dim3(cus * 4), dim3(256), 0, 0, in, out, n, cap, extra);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep && ms < best) best = ms;
                }
                const double gb = double(n) * (ND_IN * 8 + NI_IN * 4 + ND_OUT * 8 + NI_OUT * 4) / 1e9;
                printf("extra fp64 ops %4d  %-28s %4lld M positions  %7.3f ms  %6.2f TB/s\n", extra, tiled ? "[tile][component][64]" : "[component][capacity]", n >> 20, best, gb / best);
            }
        }
    }
    return 0;
}
