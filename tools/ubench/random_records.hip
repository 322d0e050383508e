// How fast can the chip read random fixed-size records of a table far larger than L2 + Infinity Cache?  (round 3: the conv1 + conv2 table
// gather reads random 256-byte records of a 363 MB table; would 2304-byte records -- the nine taps of one pattern id stored together --
// be fetched faster?)      hipcc --offload-arch=gfx950 -O3 tools/ubench/random_records.hip -o /tmp/rr && /tmp/rr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// one wave reads `per_wave` records of rec_bytes each (16 bytes per lane per load), indices from idx[]; sums so the loads are not dropped
template <int LOADS>       // loads of 1 KB (64 lanes x 16 B) per record
__global__ void k_read(const float* __restrict__ tab, const unsigned* __restrict__ idx, int per_wave, size_t rec_floats, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < per_wave; ++r) {
        const float* rec = tab + (size_t)idx[wave * per_wave + r] * rec_floats;
#pragma unroll
        for (int l = 0; l < LOADS; ++l) {
            const int off = (l * 64 + lane) * 4;
            if (off * 4 < (int)(rec_floats * 4)) acc += *reinterpret_cast<const f32x4*>(rec + off);
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}
// 8 lanes per 256-byte record (the gather kernel's pattern): a wave reads 8 different records per load pair
__global__ void k_read256(const float* __restrict__ tab, const unsigned* __restrict__ idx, int per_group, float* __restrict__ out) {
    const size_t grp = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int j = threadIdx.x & 7;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < per_group; ++r) {
        const float* rec = tab + (size_t)idx[grp * per_group + r] * 64 + j * 4;
        acc += *reinterpret_cast<const f32x4*>(rec);
        acc += *reinterpret_cast<const f32x4*>(rec + 32);
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}
int main() {
    const size_t table_bytes = 2304ull * 19684 * 8 * 8;        // 2.9 GB: nothing of it stays in the 256 MB Infinity Cache
    float* tab; hipMalloc(&tab, table_bytes); hipMemset(tab, 0, table_bytes);
    float* out; hipMalloc(&out, 4);
    const size_t total_read = 1ull << 30;                       // bytes per launch
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rec_bytes : {256, 512, 1024, 2304, 4096, 16384}) {
        const size_t nrec_tab = table_bytes / rec_bytes, nread = total_read / rec_bytes;
        std::vector<unsigned> h(nread);
        unsigned long long s = 88172645463325252ULL;
        for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (unsigned)(s % nrec_tab); }
        unsigned* idx; hipMalloc(&idx, nread * 4); hipMemcpy(idx, h.data(), nread * 4, hipMemcpyHostToDevice);
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            if (rec_bytes == 256) { const int per = 16; hipLaunchKernelGGL(k_read256, dim3((unsigned)(nread / per * 8 / 256)), dim3(256), 0, 0, tab, idx, per, out); }
            else {
                const int per = 8; const unsigned blocks = (unsigned)(nread / per / 4);
                const size_t rf = rec_bytes / 4;
                if (rec_bytes <= 1024) hipLaunchKernelGGL(k_read<1>, dim3(blocks), dim3(256), 0, 0, tab, idx, per, rf, out);
                else if (rec_bytes <= 4096) hipLaunchKernelGGL(k_read<4>, dim3(blocks), dim3(256), 0, 0, tab, idx, per, rf, out);
                else hipLaunchKernelGGL(k_read<16>, dim3(blocks), dim3(256), 0, 0, tab, idx, per, rf, out);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("random records of %5d B from a %.1f GB table: %.3f ms per GB  = %.2f TB/s\n", rec_bytes, table_bytes / 1e9, best, total_read / best / 1e9);
        hipFree(idx);
    }
    return 0;
}
