// oz_internal.h -- host-side object layouts shared by the translation units of libothellozero_amd.
#pragma once
#include <mutex>
#include <vector>

#include "../../include/othellozero_amd.h"
#include "oz_common.h"

#define OZ_API extern "C" __attribute__((visibility("default")))

int oz_current_device();

// Leaf evaluator: NNetWrapper.predict (Net/NNet.py:70-87) over a device-resident batch.
// d_count lives on the device (filled by the compaction kernel); kernels are launched for
// max_count leaves and exit early beyond *d_count, so no host round trip per step.
struct oz_net {
    int kind = 0;            // 0 = OthelloNN, 1 = integer-hash stub
    int n = 8, C = 512, max_batch = 0, device = 0;
    std::mutex mu;
    // staging of oz_net_predict (allocated at the first call, sized for max_batch): one upload [own | opp | count] and one
    // download [pi | v] per call, through host staging vectors -- no allocation, free or device-wide sync per call
    // (one position end to end 254 -> 176 us; replaying the forward as a hipGraph on top of that measured +-0)
    uint64_t* p_in = nullptr;
    float* p_out = nullptr;
    std::vector<uint64_t> h_in;
    std::vector<float> h_out;
    virtual ~oz_net() {
        if (p_in) { hipSetDevice(device); hipFree(p_in); hipFree(p_out); }
    }
    virtual int check() { return 0; }      // sticky device-side validity flags (f16x2 range)
    virtual const int* flag_device() { return nullptr; }     // the device word check() reads (nullptr: nothing to check)
    virtual int forward_device(const uint64_t* d_own, const uint64_t* d_opp, const int* d_count, int max_count,
                               float* d_pi, float* d_v, hipStream_t s) = 0;
};

int oz_net_forward_device(oz_net* net, const uint64_t* d_own, const uint64_t* d_opp, const int* d_count,
                          int max_count, float* d_pi, float* d_v, hipStream_t s);

// implicit-GEMM geometry of k_gemm_f32 (oz_net.hip): out[M][N] = act((A[M][K] . Wt[N][K]^T) * scale + shift),
// M = *d_count * Hout^2 rows (b, oy, ox); A rows are gathered per 3x3 tap with zero fill
struct GemmGeom {
    int Hin, Hout, pad, Cin, taps;   // taps = 9 (3x3 conv) or 1 (dense: Hin = Hout = 1, pad = 0)
    int N, K;                        // output channels, taps*Cin
    int relu;
    int ksplit;                      // > 1: k loop split over blockIdx.y, raw sums to partial[split] (slab floats apart)
    long long slab;
};
int oz_gemm_f32_launch(const float* in, const float* Wt, const float* scale, const float* shift, float* out,
                       const int* d_count, int max_count, int Hin, int Hout, int pad, int Cin, int taps, int N, int relu,
                       hipStream_t s, float* partial, long long partial_floats, int sizing_count = 0);
