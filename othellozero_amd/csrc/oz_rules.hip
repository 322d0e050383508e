// oz_rules.hip -- library core (errors, device selection), batched rule kernels (K1-K3) and the
// dihedral symmetry expansion of training examples (K8).  HBM-bound integer/byte work: one thread
// per position / per output example, SoA inputs, coalesced loads and stores.
#include <stdarg.h>
#include <string.h>

#include "oz_internal.h"

// ---------------------------------------------------------------- errors / device
static thread_local char g_err[512] = "";
static thread_local int g_device = -1;

void oz_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
OZ_API const char* oz_last_error(void) { return g_err; }
OZ_API int oz_version(void) { return 200; }     // 200 (round 4): edge_cap left oz_mcts_create / oz_arena_create / oz_selfplay_config
OZ_API int oz_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}
OZ_API int oz_set_device(int device) {
    OZ_HIP(hipSetDevice(device));
    g_device = device;
    return OZ_OK;
}
// The device a new object is created on / a batch call runs on: the one oz_set_device chose for this thread, else whatever
// the HIP runtime's current device is RIGHT NOW (e.g. torch.cuda.set_device(local_rank) in a one-process-per-GPU job) --
// never a value cached from an earlier call, and never overriding a device the caller selected.
int oz_current_device() {
    if (g_device >= 0) { hipSetDevice(g_device); return g_device; }
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    return d;
}

// small RAII device buffer for the batch entry points
template <typename T> struct DevBuf {
    T* p = nullptr;
    hipError_t alloc(size_t count) { return hipMalloc((void**)&p, sizeof(T) * (count ? count : 1)); }
    ~DevBuf() { if (p) hipFree(p); }
};

// Persistent staging of the oz_rules_* entry points (one per device, process lifetime): the drop-in OthelloGame asks about
// ONE position per call, so a call must not pay hipMalloc / hipFree (device-wide syncs) nor one blocking copy per array.
// A call packs its inputs into the pinned host image, does ONE upload, the kernel, ONE download on a private stream.
struct RulesStage {
    std::mutex mu;
    hipStream_t stream = nullptr;
    unsigned char *dev = nullptr, *host = nullptr;       // host = pinned (hipHostMalloc)
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (!stream) OZ_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        if (bytes <= cap) return OZ_OK;
        size_t want = cap ? cap : 4096;
        while (want < bytes) want *= 2;
        if (dev) { hipFree(dev); hipHostFree(host); dev = host = nullptr; cap = 0; }
        OZ_HIP(hipMalloc((void**)&dev, want));
        OZ_HIP(hipHostMalloc((void**)&host, want, hipHostMallocDefault));
        cap = want;
        return OZ_OK;
    }
};
static RulesStage g_stage[16];
static inline size_t pad8(size_t x) { return (x + 7) & ~(size_t)7; }

// ---------------------------------------------------------------- rule kernels
__global__ void k_legal(const uint64_t* __restrict__ own, const uint64_t* __restrict__ opp, int count, uint64_t valid,
                        uint64_t* __restrict__ legal) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) legal[i] = oz_legal(own[i], opp[i], valid);
}
__global__ void k_apply(const uint64_t* __restrict__ own, const uint64_t* __restrict__ opp, const uint8_t* __restrict__ sq,
                        int count, uint64_t* __restrict__ own_out, uint64_t* __restrict__ opp_out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint64_t a = own[i], b = opp[i];
    oz_apply(a, b, sq[i]);
    own_out[i] = a; opp_out[i] = b;
}
__global__ void k_status(const uint64_t* __restrict__ c0, const uint64_t* __restrict__ c1, int count, uint64_t valid,
                         uint8_t* __restrict__ finished, int32_t* __restrict__ p0, int32_t* __restrict__ p1,
                         int8_t* __restrict__ winner) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint64_t a = c0[i], b = c1[i];
    finished[i] = (oz_legal(a, b, valid) == 0 && oz_legal(b, a, valid) == 0) ? 1 : 0;
    int x = oz_popc(a), y = oz_popc(b);
    p0[i] = x; p1[i] = y;
    winner[i] = x >= y ? 1 : -1;          // max() over {BLACK, WHITE} keeps the first maximum
}
__global__ void k_play(const uint64_t* __restrict__ black, const uint64_t* __restrict__ white, const int8_t* __restrict__ player,
                       const uint8_t* __restrict__ sq, int count, uint64_t valid, uint64_t* __restrict__ bo,
                       uint64_t* __restrict__ wo, int8_t* __restrict__ po, uint8_t* __restrict__ fo) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint64_t b = black[i], w = white[i];
    int p = player[i], f = 0;
    oz_game_play(b, w, p, f, sq[i], valid);
    bo[i] = b; wo[i] = w; po[i] = (int8_t)p; fo[i] = (uint8_t)f;
}

static inline int grid_for(int count) { return (count + 255) / 256; }
static int check_n(int n) {
    OZ_REQUIRE(n == 4 || n == 6 || n == 8, "board size must be 4, 6 or 8 (got %d)", n);
    return OZ_OK;
}

// run `launch(dev_in, dev_out, stream)` between one upload of in_bytes and one download of out_bytes
template <typename Pack, typename Launch, typename Unpack>
static int rules_call(size_t in_bytes, size_t out_bytes, Pack pack, Launch launch, Unpack unpack) {
    const int dev = oz_current_device();
    OZ_REQUIRE(dev >= 0 && dev < 16, "device index %d out of range", dev);
    RulesStage& st = g_stage[dev];
    std::lock_guard<std::mutex> lk(st.mu);
    const size_t in_p = pad8(in_bytes);
    if (int rc = st.reserve(in_p + pad8(out_bytes))) return rc;
    pack(st.host);
    OZ_HIP(hipMemcpyAsync(st.dev, st.host, in_bytes, hipMemcpyHostToDevice, st.stream));
    launch(st.dev, st.dev + in_p, st.stream);
    OZ_HIP(hipGetLastError());
    OZ_HIP(hipMemcpyAsync(st.host + in_p, st.dev + in_p, out_bytes, hipMemcpyDeviceToHost, st.stream));
    OZ_HIP(hipStreamSynchronize(st.stream));
    unpack(st.host + in_p);
    return OZ_OK;
}

OZ_API int oz_rules_legal_moves(const uint64_t* own, const uint64_t* opp, int n, int count, uint64_t* legal) {
    if (int rc = check_n(n)) return rc;
    if (count <= 0) return OZ_OK;
    const size_t c8 = 8ull * count;
    return rules_call(2 * c8, c8,
        [&](unsigned char* h) { memcpy(h, own, c8); memcpy(h + c8, opp, c8); },
        [&](unsigned char* di, unsigned char* dout, hipStream_t s) {
            hipLaunchKernelGGL(k_legal, dim3(grid_for(count)), dim3(256), 0, s, (const uint64_t*)di, (const uint64_t*)(di + c8), count,
                               oz_valid_mask(n), (uint64_t*)dout);
        },
        [&](const unsigned char* h) { memcpy(legal, h, c8); });
}

OZ_API int oz_rules_apply_moves(const uint64_t* own, const uint64_t* opp, const uint8_t* sq, int n, int count,
                                uint64_t* own_out, uint64_t* opp_out) {
    if (int rc = check_n(n)) return rc;
    if (count <= 0) return OZ_OK;
    for (int i = 0; i < count; ++i) OZ_REQUIRE((sq[i] >> 3) < n && (sq[i] & 7) < n, "square %d outside the %dx%d board", sq[i], n, n);
    const size_t c8 = 8ull * count;
    return rules_call(2 * c8 + count, 2 * c8,
        [&](unsigned char* h) { memcpy(h, own, c8); memcpy(h + c8, opp, c8); memcpy(h + 2 * c8, sq, count); },
        [&](unsigned char* di, unsigned char* dout, hipStream_t s) {
            hipLaunchKernelGGL(k_apply, dim3(grid_for(count)), dim3(256), 0, s, (const uint64_t*)di, (const uint64_t*)(di + c8),
                               (const uint8_t*)(di + 2 * c8), count, (uint64_t*)dout, (uint64_t*)(dout + c8));
        },
        [&](const unsigned char* h) { memcpy(own_out, h, c8); memcpy(opp_out, h + c8, c8); });
}

OZ_API int oz_rules_status(const uint64_t* ch0, const uint64_t* ch1, int n, int count, uint8_t* finished,
                           int32_t* pts0, int32_t* pts1, int8_t* winner) {
    if (int rc = check_n(n)) return rc;
    if (count <= 0) return OZ_OK;
    const size_t c8 = 8ull * count, c4 = 4ull * count, c1 = pad8(count);
    // output image: pts0 | pts1 | finished | winner (each 8-byte aligned)
    return rules_call(2 * c8, 2 * pad8(c4) + 2 * c1,
        [&](unsigned char* h) { memcpy(h, ch0, c8); memcpy(h + c8, ch1, c8); },
        [&](unsigned char* di, unsigned char* dout, hipStream_t s) {
            hipLaunchKernelGGL(k_status, dim3(grid_for(count)), dim3(256), 0, s, (const uint64_t*)di, (const uint64_t*)(di + c8), count,
                               oz_valid_mask(n), (uint8_t*)(dout + 2 * pad8(c4)), (int32_t*)dout, (int32_t*)(dout + pad8(c4)),
                               (int8_t*)(dout + 2 * pad8(c4) + c1));
        },
        [&](const unsigned char* h) {
            memcpy(pts0, h, c4); memcpy(pts1, h + pad8(c4), c4);
            memcpy(finished, h + 2 * pad8(c4), count); memcpy(winner, h + 2 * pad8(c4) + c1, count);
        });
}

OZ_API int oz_rules_play(const uint64_t* black, const uint64_t* white, const int8_t* player, const uint8_t* sq, int n,
                         int count, uint64_t* black_out, uint64_t* white_out, int8_t* player_out, uint8_t* finished_out) {
    if (int rc = check_n(n)) return rc;
    if (count <= 0) return OZ_OK;
    for (int i = 0; i < count; ++i) {
        OZ_REQUIRE((sq[i] >> 3) < n && (sq[i] & 7) < n, "square %d outside the %dx%d board", sq[i], n, n);
        OZ_REQUIRE(player[i] == 1 || player[i] == -1, "player must be +1 or -1");
    }
    const size_t c8 = 8ull * count, c1 = pad8(count);
    return rules_call(2 * c8 + 2 * c1, 2 * c8 + 2 * c1,
        [&](unsigned char* h) { memcpy(h, black, c8); memcpy(h + c8, white, c8); memcpy(h + 2 * c8, player, count); memcpy(h + 2 * c8 + c1, sq, count); },
        [&](unsigned char* di, unsigned char* dout, hipStream_t s) {
            hipLaunchKernelGGL(k_play, dim3(grid_for(count)), dim3(256), 0, s, (const uint64_t*)di, (const uint64_t*)(di + c8),
                               (const int8_t*)(di + 2 * c8), (const uint8_t*)(di + 2 * c8 + c1), count, oz_valid_mask(n), (uint64_t*)dout,
                               (uint64_t*)(dout + c8), (int8_t*)(dout + 2 * c8), (uint8_t*)(dout + 2 * c8 + c1));
        },
        [&](const unsigned char* h) {
            memcpy(black_out, h, c8); memcpy(white_out, h + c8, c8); memcpy(player_out, h + 2 * c8, count); memcpy(finished_out, h + 2 * c8 + c1, count);
        });
}

// ---------------------------------------------------------------- symmetries (K8)
// training_example_symmetries, training.py:13-23: outputs in the order rot90 k=1..4 (CCW), each
// first with fliplr then without.  src(t, r, c) = source cell of output cell (r, c).
OZ_HD int oz_sym_src(int t, int n, int r, int c) {
    const int k = (t >> 1) + 1, flip = !(t & 1);
    int rr = r, cc = flip ? (n - 1 - c) : c;
    for (int q = 0; q < (k & 3); ++q) { int ti = cc, tj = n - 1 - rr; rr = ti; cc = tj; }
    return rr * n + cc;
}

OZ_API int oz_symmetry_table(int n, int32_t* perm) {
    if (int rc = check_n(n)) return rc;
    for (int t = 0; t < 8; ++t)
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) perm[(t * n + r) * n + c] = oz_sym_src(t, n, r, c);
    return OZ_OK;
}

// one thread per output cell pair: boards[(rec*8+t)][r][c][0..1]; the policy index / z by the cell-0 thread
__global__ void k_expand(const oz_record* __restrict__ recs, int64_t count, int n, int alias_final,
                         uint8_t* __restrict__ boards, int32_t* __restrict__ pol, int8_t* __restrict__ z) {
    const int n2 = n * n;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count * 8 * n2) return;
    const int cell = (int)(idx % n2);
    const int64_t ex = idx / n2;
    const int t = (int)(ex & 7);
    const oz_record rec = recs[ex >> 3];
    const int r = cell / n, c = cell % n;
    const int src = oz_sym_src(t, n, r, c), sr = src / n, sc = src % n;
    const uint64_t b = alias_final ? rec.final_black : rec.black, w = alias_final ? rec.final_white : rec.white;
    uchar2 o;
    o.x = (uint8_t)((b >> (sr * 8 + sc)) & 1); o.y = (uint8_t)((w >> (sr * 8 + sc)) & 1);
    reinterpret_cast<uchar2*>(boards)[idx] = o;
    const int a = (rec.action >> 3) * n + (rec.action & 7);
    if (src == a) pol[ex] = cell;             // the one-hot lands where its source cell is the action
    if (cell == 0) z[ex] = rec.z;
}

OZ_API int oz_examples_expand(const oz_record* records, int64_t count, int n, int alias_final, uint8_t* boards,
                              int32_t* policy_index, int8_t* z) {
    if (int rc = check_n(n)) return rc;
    if (count <= 0) return OZ_OK;
    oz_current_device();
    const int64_t nex = count * 8, cells = nex * n * n;
    DevBuf<oz_record> r; DevBuf<uint8_t> b; DevBuf<int32_t> p; DevBuf<int8_t> zz;
    OZ_HIP(r.alloc(count)); OZ_HIP(b.alloc(cells * 2)); OZ_HIP(p.alloc(nex)); OZ_HIP(zz.alloc(nex));
    OZ_HIP(hipMemcpy(r.p, records, sizeof(oz_record) * count, hipMemcpyHostToDevice));
    const int64_t blocks = (cells + 255) / 256;
    OZ_REQUIRE(blocks < (1ll << 31), "too many examples in one call");
    hipLaunchKernelGGL(k_expand, dim3((unsigned)blocks), dim3(256), 0, 0, r.p, count, n, alias_final, b.p, p.p, zz.p);
    OZ_HIP(hipGetLastError());
    OZ_HIP(hipMemcpy(boards, b.p, cells * 2, hipMemcpyDeviceToHost));
    OZ_HIP(hipMemcpy(policy_index, p.p, 4 * nex, hipMemcpyDeviceToHost));
    OZ_HIP(hipMemcpy(z, zz.p, nex, hipMemcpyDeviceToHost));
    return OZ_OK;
}
