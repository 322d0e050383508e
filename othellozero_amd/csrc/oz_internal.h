// oz_internal.h -- host-side object layouts shared by the translation units of libothellozero_amd.
#pragma once
#include <deque>
#include <mutex>
#include <vector>

#include "../../include/othellozero_amd.h"
#include "oz_common.h"

#define OZ_API extern "C" __attribute__((visibility("default")))

int oz_current_device();

// HIP-event stopwatch with named slots, for timing launches on the stream they are launched on without a host sync:
// begin / end record a pooled event pair around a launch sequence; collect() waits for every finished pair and folds it into
// the per-slot totals, drain() folds only the pairs whose end event has ALREADY completed (hipEventQuery: no host stall --
// what the drivers call inside their enqueue loops).  begin() returns a stable handle (a running pair number), so a
// collect() / drain() between a begin and its end cannot redirect the end; a pair that is still open is kept.  No event is
// created or destroyed per launch once the pool is warm; a pair whose end() never came is recycled by cancel().
struct OzTimer {
    enum { OPEN = 0, ENDED = 1, CANCELLED = 2, FAILED = 3 };
    struct Pend { hipEvent_t a, b; int slot; int state; };
    std::vector<hipEvent_t> pool;
    std::deque<Pend> pending;           // pairs [base, base + size), in begin order (= stream order on one stream)
    long long base = 0;
    std::vector<double> ms;
    std::vector<long long> count;
    explicit OzTimer(int slots = 1) : ms(slots, 0.0), count(slots, 0) {}
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
    // returns the handle to give to end() / cancel(), or -1 (events unavailable: the launch simply goes untimed)
    long long begin(int slot, hipStream_t s) {
        hipEvent_t a = get(), b = get();
        if (!a || !b) { if (a) pool.push_back(a); if (b) pool.push_back(b); return -1; }
        if (hipEventRecord(a, s) != hipSuccess) { pool.push_back(a); pool.push_back(b); return -1; }
        pending.push_back({a, b, slot, OPEN});
        return base + (long long)pending.size() - 1;
    }
    Pend* find(long long h) { return (h < base || h >= base + (long long)pending.size()) ? nullptr : &pending[(size_t)(h - base)]; }
    void end(long long h, hipStream_t s) {
        if (Pend* p = find(h)) p->state = hipEventRecord(p->b, s) == hipSuccess ? ENDED : FAILED;
    }
    void cancel(long long h) { if (Pend* p = find(h)) p->state = CANCELLED; }
    size_t backlog() const { return pending.size(); }
    // folds finished pairs from the front; wait = true blocks on every ended pair, false stops at the first one still running
    int fold(bool wait) {
        int rc = OZ_OK;
        while (!pending.empty()) {
            Pend& p = pending.front();
            if (p.state == OPEN) break;                                  // begin() without its end() yet: keep it and what follows
            if (p.state == ENDED) {
                if (!wait && hipEventQuery(p.b) != hipSuccess) break;
                float t = 0;
                if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) { ms[p.slot] += t; count[p.slot] += 1; }
                else rc = OZ_ERR_HIP;
            } else if (p.state == FAILED) rc = OZ_ERR_HIP;
            pool.push_back(p.a); pool.push_back(p.b);
            pending.pop_front();
            ++base;
        }
        return rc;
    }
    int collect() { return fold(true); }
    int drain() { return fold(false); }
    void reset() { for (auto& x : ms) x = 0; for (auto& x : count) x = 0; }
    void destroy() {
        for (auto& p : pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
        for (auto e : pool) hipEventDestroy(e);
        pending.clear(); pool.clear();
    }
};

// Persistent exact-key evaluation cache of a network (oz_net_set_eval_cache): (own, opp) -> (pi[n*n], v), the generalisation of the
// reference's per-search `_predict_cache` (othelo_mcts.py:82-88) to every search that uses the network, across batches, games and
// refilled slots.  4-way buckets (the four 16-byte keys of a bucket are one 64-byte line), empty key = (0, 0) (no Othello position is
// empty), a full bucket replaces the way the hash picks.  Lookups (leaf compaction) and inserts (after the network) run in different
// kernels of one stream, so a lookup never sees a half-written entry; `stamp` arbitrates two inserts into one entry within a batch.
// A position's (pi, v) does not depend on the batch it is evaluated in (oz_net.hip), so serving it from the cache changes no bit.
#define OZ_EC_WAYS 4
#define OZ_PREDICT_DIRECT 8
struct EvalCacheDev {
    unsigned long long* keys = nullptr;     // [entries][2]
    float* pi = nullptr;                    // [entries][n2]
    float* v = nullptr;                     // [entries]
    unsigned* stamp = nullptr;              // [entries] batch number of the last insert
    unsigned long long* counters = nullptr; // [0] lookups [1] hits [2] inserts
    unsigned buckets = 0;                   // power of two; entries = buckets * OZ_EC_WAYS; 0 = no cache
    int n2 = 0;
};

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
    // calls of at most OZ_PREDICT_DIRECT positions (the drop-in predict of ONE position, Net/NNet.py:70-87) move no buffer at all: the boards sit in
    // pinned host memory the first kernel reads directly, (pi, v) are written by the heads kernel straight into pinned host memory, and the
    // leaf count is a pointer into a device-resident table of constants -- the two staging copies were 7 us of blit kernels + their launches
    // around an 80 us forward (round 5)
    uint64_t* hp_in = nullptr;              // pinned host: own[OZ_PREDICT_DIRECT] | opp[OZ_PREDICT_DIRECT]
    float* hp_out = nullptr;                // pinned host: pi[OZ_PREDICT_DIRECT][64] | v[OZ_PREDICT_DIRECT]
    int* d_counts = nullptr;                // device: d_counts[i] = i, i = 0 .. OZ_PREDICT_DIRECT
    EvalCacheDev ec;         // oz_net_set_eval_cache; cleared whenever the weights change (oz_net_commit)
    void free_eval_cache() {
        if (!ec.buckets) return;
        hipSetDevice(device);
        hipFree(ec.keys); hipFree(ec.pi); hipFree(ec.v); hipFree(ec.stamp); hipFree(ec.counters);
        ec = EvalCacheDev();
    }
    virtual ~oz_net() {
        if (p_in) { hipSetDevice(device); hipFree(p_in); hipFree(p_out); }
        if (hp_in) { hipSetDevice(device); hipHostFree(hp_in); hipHostFree(hp_out); hipFree(d_counts); }
        free_eval_cache();
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
    // pixel-major row tiles (large batches): a 128-row tile = ONE output pixel of 128 consecutive boards, so which of the
    // nine taps fall outside the image -- or outside [core_lo, core_hi)^2, the non-zero core of a zero-bordered gradient
    // buffer -- is the same for every row of the tile and those k-tiles are skipped instead of multiplied by zeros
    // (adding exact zeros changes no bit of the result).  ksplit must be 1.
    int pixmajor, core_lo, core_hi;
};
// a split-K launch whose fixed-order reduce (+ scale, shift, ReLU) is left to the consumer of the rows (the heads kernel stages fc2's rows
// straight from the k-slice slabs: one launch less per forward): slab s holds the raw sums of k-slice s at partial[s * slab + row * N + col]
struct OzDeferredReduce { const float* partial = nullptr; long long slab = 0; int ksplit = 0; const float* scale = nullptr; const float* shift = nullptr; };
int oz_gemm_f32_launch(const float* in, const float* Wt, const float* scale, const float* shift, float* out,
                       const int* d_count, int max_count, int Hin, int Hout, int pad, int Cin, int taps, int N, int relu,
                       hipStream_t s, float* partial, long long partial_floats, int sizing_count = 0, int core_lo = 0, int core_hi = -1,
                       int* tile_rows_out = nullptr, int force_std_tile = 0, OzDeferredReduce* defer = nullptr);
// the f16x2 GEMM (k_gemm_h2, oz_net_h2.h) on h2-layout operands, fp32 rows out; zero_line = >= 256 B of zeros, flag = sticky range flag
int oz_gemm_h2_launch(const void* in_h2, const void* Wh, const float* scale, const float* shift, float* out, const int* d_count, int max_count,
                      int Hin, int Hout, int pad, int Cin, int taps, int N, hipStream_t s, float* partial, long long partial_floats,
                      const void* zero_line, int* flag);
