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
    virtual ~oz_net() {}
    virtual int check() { return 0; }      // sticky device-side validity flags (f16x2 range)
    virtual int forward_device(const uint64_t* d_own, const uint64_t* d_opp, const int* d_count, int max_count,
                               float* d_pi, float* d_v, hipStream_t s) = 0;
};

int oz_net_forward_device(oz_net* net, const uint64_t* d_own, const uint64_t* d_opp, const int* d_count,
                          int max_count, float* d_pi, float* d_v, hipStream_t s);
