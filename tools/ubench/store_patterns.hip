// Streaming-store bandwidth of the table gather's output patterns (round 3).  0.477 GB per launch:
//   A  row-major output, 8 XCD slices: block b writes the 256-byte slice (b & 7) of 64 pixels' 2 KB rows  (k_conv2_lut_xcd's pattern)
//   B  slice-major output: block b writes 64 pixels x 256 B CONTIGUOUS inside slice (b & 7)'s own region
//   C  one wave per 2 KB row, fully contiguous (the round-2 mapping)
// each with nontemporal and with plain stores.     hipcc --offload-arch=gfx950 -O3 tools/ubench/store_patterns.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <int PAT, bool NT>
__global__ __launch_bounds__(256) void k_store(v4u* __restrict__ out, long long pixels) {
    const v4u val = {threadIdx.x, blockIdx.x, 3u, 4u};
    long long u0, u1;                                            // two 16-byte units per thread
    if (PAT == 2) {                                              // C: thread = 32 B of a row
        const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
        if (t >= pixels * 64) return;
        u0 = t * 2; u1 = u0 + 1;
    } else {
        const int slice = blockIdx.x & 7, j = threadIdx.x & 7;
        for (int k = 0; k < 2; ++k) {
            const long long pixel = ((long long)(blockIdx.x >> 3) * 2 + k) * 32 + (threadIdx.x >> 3);
            if (pixel >= pixels) return;
            if (PAT == 0) u0 = pixel * 128 + slice * 16 + j; else u0 = (long long)slice * pixels * 16 + pixel * 16 + j;
            u1 = u0 + 8;
            if (NT) { __builtin_nontemporal_store(val, out + u0); __builtin_nontemporal_store(val, out + u1); } else { out[u0] = val; out[u1] = val; }
        }
        return;
    }
    if (NT) { __builtin_nontemporal_store(val, out + u0); __builtin_nontemporal_store(val, out + u1); } else { out[u0] = val; out[u1] = val; }
}
template <int PAT, bool NT> float run(v4u* out, long long pixels) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned blocks = PAT == 2 ? (unsigned)((pixels * 64 + 255) / 256) : 8u * (unsigned)((pixels + 63) / 64);
    float best = 1e9;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_store<PAT, NT>), dim3(blocks), dim3(256), 0, 0, out, pixels);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}
int main() {
    const long long pixels = 3640LL * 64;
    v4u* out; hipMalloc(&out, pixels * 2048);
    const double gb = pixels * 2048 / 1e9;
    printf("A row-major, 8 slices : nt %.4f ms (%.2f TB/s)   plain %.4f ms (%.2f TB/s)\n", run<0, true>(out, pixels), gb / run<0, true>(out, pixels), run<0, false>(out, pixels), gb / run<0, false>(out, pixels));
    printf("B slice-major         : nt %.4f ms (%.2f TB/s)   plain %.4f ms (%.2f TB/s)\n", run<1, true>(out, pixels), gb / run<1, true>(out, pixels), run<1, false>(out, pixels), gb / run<1, false>(out, pixels));
    printf("C wave per 2 KB row   : nt %.4f ms (%.2f TB/s)   plain %.4f ms (%.2f TB/s)\n", run<2, true>(out, pixels), gb / run<2, true>(out, pixels), run<2, false>(out, pixels), gb / run<2, false>(out, pixels));
    return 0;
}
