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
OZ_API int oz_version(void) { return 100; }
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
int oz_current_device() {
    if (g_device < 0) {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) d = 0;
        g_device = d;
    }
    hipSetDevice(g_device);
    return g_device;
}

// small RAII device buffer for the batch entry points
template <typename T> struct DevBuf {
    T* p = nullptr;
    hipError_t alloc(size_t count) { return hipMalloc((void**)&p, sizeof(T) * (count ? count : 1)); }
    ~DevBuf() { if (p) hipFree(p); }
};

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

OZ_API int oz_rules_legal_moves(const uint64_t* own, const uint64_t* opp, int n, int count, uint64_t* legal) {
    if (int rc = check_n(n)) return rc;
    if (count <= 0) return OZ_OK;
    oz_current_device();
    DevBuf<uint64_t> a, b, l;
    OZ_HIP(a.alloc(count)); OZ_HIP(b.alloc(count)); OZ_HIP(l.alloc(count));
    OZ_HIP(hipMemcpy(a.p, own, 8ull * count, hipMemcpyHostToDevice));
    OZ_HIP(hipMemcpy(b.p, opp, 8ull * count, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_legal, dim3(grid_for(count)), dim3(256), 0, 0, a.p, b.p, count, oz_valid_mask(n), l.p);
    OZ_HIP(hipGetLastError());
    OZ_HIP(hipMemcpy(legal, l.p, 8ull * count, hipMemcpyDeviceToHost));
    return OZ_OK;
}

OZ_API int oz_rules_apply_moves(const uint64_t* own, const uint64_t* opp, const uint8_t* sq, int n, int count,
                                uint64_t* own_out, uint64_t* opp_out) {
    if (int rc = check_n(n)) return rc;
    if (count <= 0) return OZ_OK;
    for (int i = 0; i < count; ++i) OZ_REQUIRE((sq[i] >> 3) < n && (sq[i] & 7) < n, "square %d outside the %dx%d board", sq[i], n, n);
    oz_current_device();
    DevBuf<uint64_t> a, b, ao, bo; DevBuf<uint8_t> s;
    OZ_HIP(a.alloc(count)); OZ_HIP(b.alloc(count)); OZ_HIP(ao.alloc(count)); OZ_HIP(bo.alloc(count)); OZ_HIP(s.alloc(count));
    OZ_HIP(hipMemcpy(a.p, own, 8ull * count, hipMemcpyHostToDevice));
    OZ_HIP(hipMemcpy(b.p, opp, 8ull * count, hipMemcpyHostToDevice));
    OZ_HIP(hipMemcpy(s.p, sq, count, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_apply, dim3(grid_for(count)), dim3(256), 0, 0, a.p, b.p, s.p, count, ao.p, bo.p);
    OZ_HIP(hipGetLastError());
    OZ_HIP(hipMemcpy(own_out, ao.p, 8ull * count, hipMemcpyDeviceToHost));
    OZ_HIP(hipMemcpy(opp_out, bo.p, 8ull * count, hipMemcpyDeviceToHost));
    return OZ_OK;
}

OZ_API int oz_rules_status(const uint64_t* ch0, const uint64_t* ch1, int n, int count, uint8_t* finished,
                           int32_t* pts0, int32_t* pts1, int8_t* winner) {
    if (int rc = check_n(n)) return rc;
    if (count <= 0) return OZ_OK;
    oz_current_device();
    DevBuf<uint64_t> a, b; DevBuf<uint8_t> f; DevBuf<int32_t> p0, p1; DevBuf<int8_t> w;
    OZ_HIP(a.alloc(count)); OZ_HIP(b.alloc(count)); OZ_HIP(f.alloc(count)); OZ_HIP(p0.alloc(count)); OZ_HIP(p1.alloc(count)); OZ_HIP(w.alloc(count));
    OZ_HIP(hipMemcpy(a.p, ch0, 8ull * count, hipMemcpyHostToDevice));
    OZ_HIP(hipMemcpy(b.p, ch1, 8ull * count, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_status, dim3(grid_for(count)), dim3(256), 0, 0, a.p, b.p, count, oz_valid_mask(n), f.p, p0.p, p1.p, w.p);
    OZ_HIP(hipGetLastError());
    OZ_HIP(hipMemcpy(finished, f.p, count, hipMemcpyDeviceToHost));
    OZ_HIP(hipMemcpy(pts0, p0.p, 4ull * count, hipMemcpyDeviceToHost));
    OZ_HIP(hipMemcpy(pts1, p1.p, 4ull * count, hipMemcpyDeviceToHost));
    OZ_HIP(hipMemcpy(winner, w.p, count, hipMemcpyDeviceToHost));
    return OZ_OK;
}

OZ_API int oz_rules_play(const uint64_t* black, const uint64_t* white, const int8_t* player, const uint8_t* sq, int n,
                         int count, uint64_t* black_out, uint64_t* white_out, int8_t* player_out, uint8_t* finished_out) {
    if (int rc = check_n(n)) return rc;
    if (count <= 0) return OZ_OK;
    for (int i = 0; i < count; ++i) {
        OZ_REQUIRE((sq[i] >> 3) < n && (sq[i] & 7) < n, "square %d outside the %dx%d board", sq[i], n, n);
        OZ_REQUIRE(player[i] == 1 || player[i] == -1, "player must be +1 or -1");
    }
    oz_current_device();
    DevBuf<uint64_t> b, w, bo, wo; DevBuf<int8_t> p, po; DevBuf<uint8_t> s, fo;
    OZ_HIP(b.alloc(count)); OZ_HIP(w.alloc(count)); OZ_HIP(bo.alloc(count)); OZ_HIP(wo.alloc(count));
    OZ_HIP(p.alloc(count)); OZ_HIP(po.alloc(count)); OZ_HIP(s.alloc(count)); OZ_HIP(fo.alloc(count));
    OZ_HIP(hipMemcpy(b.p, black, 8ull * count, hipMemcpyHostToDevice));
    OZ_HIP(hipMemcpy(w.p, white, 8ull * count, hipMemcpyHostToDevice));
    OZ_HIP(hipMemcpy(p.p, player, count, hipMemcpyHostToDevice));
    OZ_HIP(hipMemcpy(s.p, sq, count, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_play, dim3(grid_for(count)), dim3(256), 0, 0, b.p, w.p, p.p, s.p, count, oz_valid_mask(n), bo.p, wo.p, po.p, fo.p);
    OZ_HIP(hipGetLastError());
    OZ_HIP(hipMemcpy(black_out, bo.p, 8ull * count, hipMemcpyDeviceToHost));
    OZ_HIP(hipMemcpy(white_out, wo.p, 8ull * count, hipMemcpyDeviceToHost));
    OZ_HIP(hipMemcpy(player_out, po.p, count, hipMemcpyDeviceToHost));
    OZ_HIP(hipMemcpy(finished_out, fo.p, count, hipMemcpyDeviceToHost));
    return OZ_OK;
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
