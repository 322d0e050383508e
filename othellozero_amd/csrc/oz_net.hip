// oz_net.hip -- batched leaf evaluation: OthelloNN inference (K9-K12) for gfx950.
//
// Replaces NNetWrapper.predict (Net/NNet.py:70-87) over the Keras graph of Net/OthelloNN.py:42-56:
//   conv1 3x3 same 2->C, conv2 3x3 same C->C, conv3 3x3 valid, conv4 3x3 valid (each + BN + ReLU),
//   Flatten (h,w,c), Dense 1024 + BN + ReLU, Dense 512 + BN + ReLU, softmax(n*n) head, tanh head.
// BN uses moving statistics with epsilon 1e-3 (Keras default) folded into a per-channel scale/shift;
// Dropout is identity at inference.
//
// Kernels:
//   k_conv1       bitboards -> NHWC activations; the plane unpack (K9) is fused, K = 18, VALU.
//   k_gemm_f32    implicit-GEMM 3x3 convolution / dense layer on the fp32 matrix cores
//                 (v_mfma_f32_32x32x2_f32): 128x128x32 block tile, 4 waves of 64x64 (2x2 MFMA tiles),
//                 A (activation rows, gathered per 3x3 tap with zero fill) and B (weights, stored
//                 [Cout][K] so k is contiguous) staged through padded LDS tiles, register-prefetch
//                 double buffering, fused scale/shift/ReLU epilogue.  MFMA-bound: 64 MFMAs (4096
//                 cycles) per 32 KB of staged operands.
//   k_heads       policy softmax + value tanh, one wavefront per position.
// Every output row is an independent k-ordered fp32 FMA chain, so a position's (pi, v) does not
// depend on where in the batch it sits (the search relies on that for reproducibility).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "oz_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------- conv1 (+ plane unpack)
// out[m][co] = relu(scale[co] * sum_{tap,ch} x[b][y+ky-1][x+kx-1][ch] * W[tap][ch][co] + shift[co])
// one thread per (row m, 4 consecutive output channels)
__global__ __launch_bounds__(256) void k_conv1(const uint64_t* __restrict__ own, const uint64_t* __restrict__ opp,
                                               const int* __restrict__ d_count, int n, int C,
                                               const float* __restrict__ W /*[9][2][C]*/, const float* __restrict__ scale,
                                               const float* __restrict__ shift, float* __restrict__ out /*[B][n*n][C]*/) {
    const int P = n * n, cq = C >> 2;
    const long long M = (long long)(*d_count) * P;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long m = idx / cq;
    if (m >= M) return;
    const int c4 = (int)(idx % cq) * 4;
    const int b = (int)(m / P), pix = (int)(m % P), y = pix / n, x = pix % n;
    const uint64_t o = own[b], p = opp[b];
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = y + ky - 1, ix = x + kx - 1;
            if (iy < 0 || iy >= n || ix < 0 || ix >= n) continue;
            const int sq = iy * 8 + ix;
            const float a0 = (float)((o >> sq) & 1), a1 = (float)((p >> sq) & 1);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(W + (size_t)((ky * 3 + kx) * 2 + 0) * C + c4);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(W + (size_t)((ky * 3 + kx) * 2 + 1) * C + c4);
            acc0 = fmaf(a0, w0[0], acc0); acc1 = fmaf(a0, w0[1], acc1); acc2 = fmaf(a0, w0[2], acc2); acc3 = fmaf(a0, w0[3], acc3);
            acc0 = fmaf(a1, w1[0], acc0); acc1 = fmaf(a1, w1[1], acc1); acc2 = fmaf(a1, w1[2], acc2); acc3 = fmaf(a1, w1[3], acc3);
        }
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c4), sh = *reinterpret_cast<const f32x4*>(shift + c4);
    f32x4 r;
    r[0] = fmaxf(fmaf(acc0, sc[0], sh[0]), 0.f); r[1] = fmaxf(fmaf(acc1, sc[1], sh[1]), 0.f);
    r[2] = fmaxf(fmaf(acc2, sc[2], sh[2]), 0.f); r[3] = fmaxf(fmaf(acc3, sc[3], sh[3]), 0.f);
    *reinterpret_cast<f32x4*>(out + (size_t)m * C + c4) = r;
}

// ---------------------------------------------------------------- implicit GEMM on fp32 MFMA
#define GM_BM 128
#define GM_BN 128
#define GM_BK 32
#define GM_LDS_STRIDE 32   // floats per staged row: 128 B, no padding -- rows arrive by LDS-DMA (8 rows = 1 KB per wave instruction) and the 16-byte
                           // chunks of a row are XOR-swizzled with row & 7 on the SOURCE side and on the read side (conflict-free, see k_gemm_f32)
typedef const __attribute__((address_space(1))) void* gm_gptr;
typedef __attribute__((address_space(3))) void* gm_lptr;
__device__ float g_gm_zero_line[32];          // 128 B of zeros: the source of taps outside the image / rows beyond M


// out[M][N] = act((A[M][K] . Wt[N][K]^T) * scale + shift), M = *d_count * Hout^2, rows (b, oy, ox)
// Tile configurations (round 4): waves in NWM x NWN, every wave TI x TJ MFMA tiles of 32 x 32.
//   GmStd  2 x 2 waves of 64 x 64   -> block 128 x 128, 256 threads, 64 KB of LDS, two blocks per CU: every layer, every mode (pixel-major
//          tiles, split-K, the trainer's gradients)
//   GmBig  2 x 4 waves of 128 x 64  -> block 256 x 256, 512 threads, 128 KB of LDS, one block per CU: the 3x3 convolutions of large batches --
//          half the operand bytes per MFMA (the ablations put 13 % of a launch on the operand loads themselves), same sums in the same order
template <int NWM_, int NWN_, int TI_, int TJ_, int OCC_> struct GmCfg {
    static constexpr int NWM = NWM_, NWN = NWN_, TI = TI_, TJ = TJ_, OCC = OCC_;
    static constexpr int BM = NWM * TI * 32, BN = NWN * TJ * 32, NT = NWM * NWN * 64, RPP = NT / 8;      // RPP: rows staged per pass of the block
    static constexpr int IA = BM / RPP, IB = BN / RPP;
    static constexpr int LDS_BYTES = 2 * (BM + BN) * GM_LDS_STRIDE * 4;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile shape");
};
typedef GmCfg<2, 2, 2, 2, 2> GmStd;
typedef GmCfg<2, 4, 4, 2, 1> GmBig;
static_assert(GmStd::BM == GM_BM && GmStd::BN == GM_BN, "GM_BM / GM_BN describe the standard tile");

template <typename CF>
__global__ __launch_bounds__(CF::NT, CF::OCC) void k_gemm_f32(const float* __restrict__ in, const float* __restrict__ Wt,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              float* __restrict__ out, const int* __restrict__ d_count,
                                                              GemmGeom g, int num_mt, float* __restrict__ partial) {
    constexpr int BM = CF::BM, BN = CF::BN, IA = CF::IA, IB = CF::IB, RPP = CF::RPP, TI = CF::TI, TJ = CF::TJ;
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];                       // [buf][A rows | B rows][32 floats]
    auto tileA = [&](int buf) -> float* { return lds_dyn + (size_t)buf * (BM + BN) * GM_LDS_STRIDE; };
    auto tileB = [&](int buf) -> float* { return lds_dyn + ((size_t)buf * (BM + BN) + BM) * GM_LDS_STRIDE; };
    // XCD-aware tile order: the N/BN column tiles of one row tile run on the same XCD (ids b, b+8 share an L2)
    const int nnt = g.N / BN;
    const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
    const int mt = (j / nnt) * 8 + xcd, nt = j % nnt;
    const int P = g.Hout * g.Hout;
    const int count = *d_count;
    const long long M = (long long)count * P;
    // row r of tile mt -> (board, output pixel): board-major (b, oy, ox) rows, or one pixel of BM consecutive boards
    int tile_pix = 0, tile_b0 = 0;
    if (g.pixmajor) {
        // tile mt = (pixel, board group) with the GROUP minor: workgroup ids go round-robin over the 8 XCDs, so an XCD sees every
        // pixel of its board groups -- border pixels (few taps) and interior pixels (all nine) in the same mix on every XCD
        // (pixel-minor order put one board COLUMN on each XCD: the two border columns idled while six XCDs did the work)
        const int ngrp = (count + BM - 1) / BM;
        if (mt >= ngrp * P) return;
        tile_pix = mt / ngrp; tile_b0 = (mt % ngrp) * BM;
    } else if (mt >= num_mt || (long long)mt * BM >= M) return;
    auto row_bp = [&](int r, int& b, int& pix) -> bool {
        if (g.pixmajor) { b = tile_b0 + r; pix = tile_pix; return b < count; }
        const long long m = (long long)mt * BM + r;
        b = (int)(m / P); pix = (int)(m % P);
        return m < M;
    };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / CF::NWN, wn = wave % CF::NWN;
    const int chunk = tid & 7, srow = tid >> 3;          // staging: row srow + RPP * i, k offset chunk * 4

    // per staged A row: base offset of input pixel (oy-pad, ox-pad) and the 9-bit tap validity mask
    long long abase[IA];
    unsigned amask[IA];
    const int chi = g.core_hi < 0 ? g.Hin : g.core_hi;
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        int b, pix;
        abase[i] = 0; amask[i] = 0;
        if (row_bp(srow + RPP * i, b, pix)) {
            const int oy = pix / g.Hout, ox = pix % g.Hout;
            abase[i] = (((long long)b * g.Hin + (oy - g.pad)) * g.Hin + (ox - g.pad)) * g.Cin;
            unsigned mk = 0;
            for (int t = 0; t < g.taps; ++t) {
                const int iy = oy - g.pad + t / 3, ix = ox - g.pad + t % 3;
                if (iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Hin) mk |= 1u << t;
            }
            amask[i] = mk;
        }
    }
    // taps this tile has to visit: all of them, or (pixel-major) those that read inside the image / the non-zero core
    unsigned tmask = (1u << g.taps) - 1u;
    if (g.pixmajor) {
        const int oy = tile_pix / g.Hout, ox = tile_pix % g.Hout;
        tmask = 0;
        for (int t = 0; t < g.taps; ++t) {
            const int iy = oy - g.pad + t / 3, ix = ox - g.pad + t % 3;
            if (iy >= g.core_lo && iy < chi && ix >= g.core_lo && ix < chi) tmask |= 1u << t;
        }
    }
    const int kpt = g.Cin / GM_BK;                       // k-tiles per tap
    auto tap_of = [&](int kt) -> int {                   // the (kt / kpt)-th visited tap
        if (!g.pixmajor) return kt / kpt;
        unsigned m = tmask;
        for (int q = kt / kpt; q > 0; --q) m &= m - 1;
        return __builtin_ctz(m);
    };
    // Staging (round 4): global -> LDS by 16-byte LDS-DMA, no staging registers, no ds_write pass.  Thread (srow, chunk) owns LDS rows
    // srow + RPP i, physical chunk `chunk`: a wave's instruction i fills the 8 rows 8 w + RPP i .. + 7 = 1 KB contiguous.  Physical chunk p of row r
    // holds LOGICAL chunk p ^ ((r >> 1) & 7).  A ds_read_b128 is served in 16-lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... over
    // 64 banks = 256 B (MI355X_MICROARCH.md): with the 32x32x2 operand map (lane: row r32, k-half) a group holds 8 even and 8 odd rows, rows of one
    // parity share the 128-byte half of the bank window, and (r >> 1) & 7 is a bijection on each parity's eight rows in every group -- conflict-free
    // without padding (SQ_LDS_BANK_CONFLICT 0; the first key tried, r & 7, measured 50 % conflict cycles).  Rounds 1-3 staged through registers
    // into rows padded to 144 B.  What the change bought (tools/f32_forward_probe.py, 3640 positions): fc1 707 -> 573 us, fc2 103 -> 88 us,
    // conv3 5063 -> 4901 us with the conflict-free key: the ablations put 13 % of a convolution launch on the operand loads THEMSELVES (conv3
    // 5063 us whole, 4401 without the loads, 4903 with every load a cache hit, 5082 without the fragment reads, 5242 without the barrier),
    // not on how they are staged -- hence the 256 x 256 tile for the large convolutions (half the operand bytes per MFMA).
    const int swz = chunk ^ ((srow >> 1) & 7);           // the logical chunk this thread fetches (rows srow + RPP i share the key: RPP / 2 % 8 == 0)
    static_assert((RPP / 2) % 8 == 0, "rows of one thread must share the swizzle key");
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const float* brow[IB];
#pragma unroll
    for (int i = 0; i < IB; ++i) brow[i] = Wt + (size_t)(nt * BN + srow + RPP * i) * g.K + swz * 4;
    const float* zsrc = g_gm_zero_line + swz * 4;

    auto stage = [&](int kt, int buf) {
        const int tap = tap_of(kt), ci0 = (kt % kpt) * GM_BK, k0 = tap * g.Cin + ci0;
        const long long toff = ((long long)(tap / 3) * g.Hin + (tap % 3)) * g.Cin + ci0 + swz * 4;
        float* la = tileA(buf);
        float* lb = tileB(buf);
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const float* ga = ((amask[i] >> tap) & 1) ? in + abase[i] + toff : zsrc;
            __builtin_amdgcn_global_load_lds((gm_gptr)ga, (gm_lptr)(la + (wave_u * 8 + RPP * i) * GM_LDS_STRIDE), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < IB; ++i)
            __builtin_amdgcn_global_load_lds((gm_gptr)(brow[i] + k0), (gm_lptr)(lb + (wave_u * 8 + RPP * i) * GM_LDS_STRIDE), 16, 0, 0);
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int jj = 0; jj < TJ; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    // split-K (training step, few row tiles): blockIdx.y owns k-tiles [kt0, nk); raw partial sums go to partial[split]
    const int nk_all = g.pixmajor ? __builtin_popcount(tmask) * kpt : g.K / GM_BK, nk_s = (nk_all + g.ksplit - 1) / g.ksplit;
    const int kt0 = blockIdx.y * nk_s, nk = min(nk_all, kt0 + nk_s);
    const int r32 = lane & 31, half = lane >> 5;
    const int key = (r32 >> 1) & 7;                      // rows r32 + 32 t of a wave tile share it
    if (kt0 < nk) stage(kt0, kt0 & 1);
    // LDS-DMA rows written on behalf of OTHER waves become visible to a reader only after the ISSUING wave's vmcnt wait and a barrier
    // the reader has passed.  The wait is explicit, as in the h2 kernels: __syncthreads()'s workgroup-scope fence does not promise a
    // vmcnt wait on gfx9 and gfx950 has no automatic wait before s_barrier (this LLVM emits one today, a later one may not).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                     // publishes the tile
    for (int kt = kt0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, buf ^ 1);         // the other buffer: every wave finished reading it before the last barrier
        const float* At = tileA(buf) + (wm * TI * 32 + r32) * GM_LDS_STRIDE;
        const float* Bt = tileB(buf) + (wn * TJ * 32 + r32) * GM_LDS_STRIDE;
        // fragments double-buffered in registers: the ds_read_b128 of step s + 1 are issued BEFORE the MFMAs of step s
        f32x4 fa[2][TI], fb[2][TJ];
        auto ldfrag = [&](int s, int w) {
            const int off = ((2 * s + half) ^ key) * 4;  // lane (r, half) holds k = 8s + 4*half + {0..3}: logical chunk 2s + half of its row
#pragma unroll
            for (int i = 0; i < TI; ++i) fa[w][i] = *reinterpret_cast<const f32x4*>(At + i * 32 * GM_LDS_STRIDE + off);
#pragma unroll
            for (int jj = 0; jj < TJ; ++jj) fb[w][jj] = *reinterpret_cast<const f32x4*>(Bt + jj * 32 * GM_LDS_STRIDE + off);
        };
        ldfrag(0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            // MFMA q pairs k = 8s+q (half 0) with 8s+4+q (half 1)
            const int w = s & 1;
            if (s < 3) ldfrag(s + 1, w ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int jj = 0; jj < TJ; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[w][i][q], fb[w][jj][q], acc[i][jj], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next tile's pieces of THIS wave have landed (explicit: see the prologue) ...
        __syncthreads();                                 // ... and the barrier publishes them; everybody is done with this tile
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    // Rows outer, the TJ column tiles inner: a row's output offset is formed once and used at once.  Output row of tile row tr: board-major
    // tiles hold rows mt BM + tr of the [M][N] output as they are (no division: the (b, pix) round trip of row_bp cost the 256 x 256 tile 84
    // spilled registers = 340 B of scratch per thread, 0.18 GB of extra writes per conv3 launch -- VERDICT r4 #7, round 5); pixel-major
    // tiles hold pixel tile_pix of boards tile_b0 + tr.
    int colj[TJ];
    float scj[TJ], shj[TJ];
#pragma unroll
    for (int jj = 0; jj < TJ; ++jj) {
        colj[jj] = nt * BN + wn * TJ * 32 + jj * 32 + r32;
        scj[jj] = scale[colj[jj]]; shj[jj] = shift[colj[jj]];
    }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tr = wm * TI * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const long long m = g.pixmajor ? (long long)(tile_b0 + tr) * P + tile_pix : (long long)mt * BM + tr;
            const bool ok = g.pixmajor ? tile_b0 + tr < count : m < M;
            if (!ok) continue;
            if (g.ksplit > 1) {
                float* prow = partial + (size_t)blockIdx.y * g.slab + (size_t)m * g.N;
#pragma unroll
                for (int jj = 0; jj < TJ; ++jj) prow[colj[jj]] = acc[i][jj][r];
            } else {
                float* orow = out + (size_t)m * g.N;
#pragma unroll
                for (int jj = 0; jj < TJ; ++jj) {
                    float v = fmaf(acc[i][jj][r], scj[jj], shj[jj]);
                    if (g.relu) v = v > 0.f ? v : 0.f;
                    orow[colj[jj]] = v;
                }
            }
        }
}

// ---------------------------------------------------------------- dense layers on few rows (M <= 64): a weight STREAM, not a GEMM
// out[m][n] = sum_k A[m][k] Wt[n][k] with M = the batch (<= 64 boards): 2 M FLOP per weight, so the layer is bound by reading
// Wt once (33.5 MB for fc1) and the 128-row tiles of k_gemm_f32 waste 3/4 of their MFMAs on padding rows while only
// 64-128 blocks pull on HBM.  Here a block owns 128 output columns x `kb` reduction indices: every lane fetches ITS column's
// weights straight into registers (kb / 8 16-byte loads per lane, all in flight at once; a row's consecutive 32-byte pieces
// meet in the L2 line they share), the few A rows come through LDS, one 32 x 32 MFMA tile per wave and row tile, raw partial
// sums per k-slice; k_splitk_reduce_f32 adds the slices in fixed order and applies scale / shift.  The split depends on
// (N, K) only, so a row's result does not depend on the size of the call.
#define SK_COLS 128
#define SK_KB_MAX 256
__global__ __launch_bounds__(256) void k_gemm_f32_skinny(const float* __restrict__ A, const float* __restrict__ Wt, const int* __restrict__ d_count,
                                                         int K, int N, int kb, float* __restrict__ partial, long long slab, GemmGeom g) {
    __shared__ __attribute__((aligned(16))) float As[64 * (SK_KB_MAX + 4)];
    const int P = g.Hout * g.Hout;
    const int M = *d_count * P;                              // rows of this call: boards (dense) or boards x output pixels (3x3 conv), <= 64
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r32 = lane & 31, half = lane >> 5;
    const int n0 = blockIdx.x * SK_COLS + wave * 32, k0 = blockIdx.y * kb, stride = kb + 4;
    // this lane's column: kb / 8 float4 of weights, k = k0 + 8 q + 4 half + {0..3}
    const float* wrow = Wt + (size_t)(n0 + r32) * K + k0 + 4 * half;
    f32x4 b[SK_KB_MAX / 8];
#pragma unroll
    for (int q = 0; q < SK_KB_MAX / 8; ++q)
        if (q * 8 < kb) b[q] = *reinterpret_cast<const f32x4*>(wrow + 8 * q);
    // A rows [0, 32 * tiles) x kb -> LDS (zero rows beyond M)
    const int tiles = M > 32 ? 2 : 1, q4 = kb / 4;
    // (a k-slice lies inside ONE tap of a 3x3 layer: kb divides Cin; row m of a convolution reads input pixel (oy - pad + dy, ox - pad + dx))
    const int tap = k0 / g.Cin, ci0 = k0 - tap * g.Cin, dy = g.taps == 9 ? tap / 3 : 0, dx = g.taps == 9 ? tap % 3 : 0;
    for (int u = tid; u < tiles * 32 * q4; u += 256) {
        const int m = u / q4, c = (u % q4) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < M) {
            const int b = m / P, pix = m - b * P, iy = pix / g.Hout - g.pad + dy, ix = pix % g.Hout - g.pad + dx;
            if (iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Hin)
                v = *reinterpret_cast<const f32x4*>(A + (((size_t)b * g.Hin + iy) * g.Hin + ix) * g.Cin + ci0 + c);
        }
        *reinterpret_cast<f32x4*>(&As[m * stride + c]) = v;
    }
    __syncthreads();
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const float* a0p = &As[r32 * stride + 4 * half];
    const float* a1p = a0p + 32 * stride;
#pragma unroll
    for (int q = 0; q < SK_KB_MAX / 8; ++q) {
        if (q * 8 < kb) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(a0p + 8 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b[q][j], acc0, 0, 0, 0);
            if (tiles == 2) {
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(a1p + 8 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b[q][j], acc1, 0, 0, 0);
            }
        }
    }
    float* outp = partial + (size_t)blockIdx.y * slab;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < M) outp[(size_t)m * N + n0 + r32] = acc0[r];
        if (tiles == 2 && m + 32 < M) outp[(size_t)(m + 32) * N + n0 + r32] = acc1[r];
    }
}

// out = act(sum over splits (fixed order) * scale + shift)
__global__ __launch_bounds__(256) void k_splitk_reduce_f32(const float* __restrict__ partial, long long slab, int ksplit, int N, int P,
                                                           const int* __restrict__ d_count, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int relu, float* __restrict__ out) {
    const long long total = (long long)(*d_count) * P * N;
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= total) return;
    // the slices are ADDED in order s = 0, 1, 2, ... (bit-reproducible) but FETCHED 24 at a time: a thread's loop over `ksplit` slabs was a chain of
    // dependent-looking L2 round trips -- 72 slabs of one position's conv3 took 18.9 us, twice the weight-stream kernel in front of it (round 5,
    // rocprofv3 of predict(): tools/trace_gaps.py); the additions and their order are unchanged
    f32x4 a = *reinterpret_cast<const f32x4*>(partial + i);
    constexpr int RB = 24;                                   // slices in flight per thread (72 slabs of a one-position conv layer = three round trips)
    for (int s0 = 1; s0 < ksplit; s0 += RB) {
        f32x4 b[RB];
#pragma unroll
        for (int q = 0; q < RB; ++q) if (s0 + q < ksplit) b[q] = *reinterpret_cast<const f32x4*>(partial + (size_t)(s0 + q) * slab + i);
#pragma unroll
        for (int q = 0; q < RB; ++q) if (s0 + q < ksplit) { a[0] += b[q][0]; a[1] += b[q][1]; a[2] += b[q][2]; a[3] += b[q][3]; }
    }
    const int col = (int)(i % N);
    f32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v = fmaf(a[q], scale[col + q], shift[col + q]);
        r[q] = relu && v < 0.f ? 0.f : v;
    }
    *reinterpret_cast<f32x4*>(out + i) = r;
}

// shared launcher (inference f32 path and the training step, oz_train.hip).  `partial` (optional, `partial_floats`
// long): when the launch would have fewer blocks than the chip has CUs, the k loop is split over blockIdx.y and reduced
// in a fixed order by k_splitk_reduce_f32 -- same result for every batch position, different rounding than ksplit = 1.
int oz_gemm_f32_launch(const float* in, const float* Wt, const float* scale, const float* shift, float* out,
                       const int* d_count, int max_count, int Hin, int Hout, int pad, int Cin, int taps, int N, int relu,
                       hipStream_t s, float* partial, long long partial_floats, int sizing_count, int core_lo, int core_hi,
                       int* tile_rows_out, int force_std_tile, OzDeferredReduce* defer) {
    // defer (optional): a split launch leaves its reduce to the consumer (defer->ksplit > 1 says so; relu must be 1, the consumer applies it)
    if (defer) *defer = OzDeferredReduce();
    // tile_rows_out (optional): the row-tile height this launch ran on -- 64 = the weight-stream kernel, 128 = GmStd, 256 = GmBig;
    // force_std_tile: never the 256 x 256 tile (OZ_NET_OPT_F32_STD_TILE: the bit-identity screen of the two tiles)
    int tile_rows_dummy = 0;
    int& tile_rows = tile_rows_out ? *tile_rows_out : tile_rows_dummy;
    OZ_REQUIRE(N % GM_BN == 0 && Cin % GM_BK == 0, "gemm_f32: N %% 128 and Cin %% 32 must be 0 (N=%d Cin=%d)", N, Cin);
    GemmGeom g;
    g.Hin = Hin; g.Hout = Hout; g.pad = pad; g.Cin = Cin; g.taps = taps; g.N = N; g.K = taps * Cin; g.relu = relu;
    g.core_lo = core_lo; g.core_hi = core_hi;
    const long long Mmax = (long long)max_count * Hout * Hout;
    // pixel-major tiles skip the k-tiles of taps that only read zeros; they pay when the boards fill the 128-row tiles
    // (keyed on the call's capacity `max_count` / `sizing_count`, like the split-K choice: per-network constants)
    const int cap = sizing_count > 0 ? sizing_count : max_count;
    // ... and only where there IS a tap to skip ('same' padding, or a zero-bordered input): on 'valid' layers over dense inputs the
    // pixel-major order has nothing to skip and costs L2 reuse of the overlapping 3x3 windows (HBM-side reads of conv3 at 4096
    // positions 3.5 -> 4.2 GB per launch, -4 % throughput -- measured, round 2)
    const bool has_zero_taps = pad > 0 || core_lo > 0 || (core_hi >= 0 && core_hi < Hin);
    g.pixmajor = taps == 9 && has_zero_taps && cap >= 2 * GM_BM && ((cap + GM_BM - 1) / GM_BM) * GM_BM <= cap + cap / 8;
    const int num_mt = g.pixmajor ? ((max_count + GM_BM - 1) / GM_BM) * Hout * Hout : (int)((Mmax + GM_BM - 1) / GM_BM);
    const int grid = ((num_mt + 7) / 8) * 8 * (N / GM_BN);
    // dense layers on at most 64 rows: the weight-stream kernel (the capacity decides, a per-network constant)
    const int Pout = Hout * Hout;
    if (partial && (long long)cap * Pout <= 64 && (long long)max_count * Pout <= 64 && Cin % 64 == 0 && N % SK_COLS == 0) {
        int kb = SK_KB_MAX;
        while (kb > 64 && (Cin % kb != 0 || (long long)(N / SK_COLS) * (g.K / kb) < 192)) kb /= 2;
        const int ks = g.K / kb;
        if ((long long)ks * max_count * Pout * N <= partial_floats) {
            const long long slab = (long long)max_count * Pout * N;
            g.ksplit = ks; g.slab = slab; g.pixmajor = 0;
            tile_rows = 64;
            hipLaunchKernelGGL(k_gemm_f32_skinny, dim3(N / SK_COLS, ks), dim3(256), 0, s, in, Wt, d_count, g.K, N, kb, partial, slab, g);
            if (defer && relu) { defer->partial = partial; defer->slab = slab; defer->ksplit = ks; defer->scale = scale; defer->shift = shift; OZ_HIP(hipGetLastError()); return OZ_OK; }
            const long long quads = (slab + 3) / 4;
            // (round 5, measured and removed: one output per thread with all 72 slices in flight -- 6.1 against 6.6 us: a kernel of this kind is
            //  launch + count + one round trip to the slabs + store ~ 5 us whatever the loop looks like; fewer launches is what is left)
            hipLaunchKernelGGL(k_splitk_reduce_f32, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, partial, slab, ks, N, Pout, d_count, scale,
                               shift, relu, out);
            OZ_HIP(hipGetLastError());
            return OZ_OK;
        }
    }
    int ksplit = 1;
    const int nk = g.K / GM_BK;
    if (partial && !g.pixmajor) {
        // the split is chosen from `sizing_count` when given (a per-network constant: results then do not depend on the
        // size of an individual call), else from this launch's own grid
        const long long Ms = (long long)(sizing_count > 0 ? sizing_count : max_count) * Hout * Hout;
        // blocks that have rows to work on (the grid is padded to a multiple of 8 row tiles for the XCD mapping; padding blocks exit at once)
        const int grid_s = (int)((Ms + GM_BM - 1) / GM_BM) * (N / GM_BN);
        const int split_blocks = 256;                       // one block per CU
        while (ksplit < 16 && grid_s * ksplit < split_blocks && nk / (ksplit * 2) >= 8 && (long long)(ksplit * 2) * (Ms > Mmax ? Ms : Mmax) * N <= partial_floats) ksplit *= 2;
    }
    g.ksplit = ksplit; g.slab = Mmax * N;
    {   // the dynamic-LDS limits of the two instantiations, once per device
        static bool attr_done[64] = {};
        int dev_now = 0;
        OZ_HIP(hipGetDevice(&dev_now));
        if (!attr_done[dev_now & 63]) {
            OZ_HIP(hipFuncSetAttribute((const void*)k_gemm_f32<GmStd>, hipFuncAttributeMaxDynamicSharedMemorySize, GmStd::LDS_BYTES));
            OZ_HIP(hipFuncSetAttribute((const void*)k_gemm_f32<GmBig>, hipFuncAttributeMaxDynamicSharedMemorySize, GmBig::LDS_BYTES));
            attr_done[dev_now & 63] = true;
        }
    }
    // 3x3 convolutions that fill the chip with 256 x 256 tiles (no pixel-major skipping, no k split): the big tile -- every output element's
    // products are added in the same order as on the standard tile (bit-identical), keyed on the call's capacity like the other choices
    const long long big_blocks = ((Mmax + GmBig::BM - 1) / GmBig::BM) * (N / GmBig::BN);
    if (taps == 9 && !g.pixmajor && ksplit == 1 && N % GmBig::BN == 0 && big_blocks >= 192 && !force_std_tile) {
        tile_rows = GmBig::BM;
        const int num_mt_big = (int)((Mmax + GmBig::BM - 1) / GmBig::BM);
        const int grid_big = ((num_mt_big + 7) / 8) * 8 * (N / GmBig::BN);
        hipLaunchKernelGGL(k_gemm_f32<GmBig>, dim3(grid_big, 1), dim3(GmBig::NT), GmBig::LDS_BYTES, s, in, Wt, scale, shift, out, d_count, g, num_mt_big, partial);
        OZ_HIP(hipGetLastError());
        return OZ_OK;
    }
    tile_rows = GmStd::BM;
    hipLaunchKernelGGL(k_gemm_f32<GmStd>, dim3(grid, ksplit), dim3(GmStd::NT), GmStd::LDS_BYTES, s, in, Wt, scale, shift, out, d_count, g, num_mt, partial);
    OZ_HIP(hipGetLastError());
    if (ksplit > 1 && defer && relu) { defer->partial = partial; defer->slab = g.slab; defer->ksplit = ksplit; defer->scale = scale; defer->shift = shift; }
    else if (ksplit > 1) {
        const long long quads = (Mmax * N + 3) / 4;
        hipLaunchKernelGGL(k_splitk_reduce_f32, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, partial, g.slab, ksplit, N, Hout * Hout,
                           d_count, scale, shift, relu, out);
        OZ_HIP(hipGetLastError());
    }
    return OZ_OK;
}

#include "oz_net_h2.h"
#include "oz_net_b3.h"
#define B3_MIN_BATCH 128       // precision bf16x3: networks of at least this capacity run conv3 / conv4 / fc1 on k_gemm_b3 (a per-network constant); below it the
                               // layers are weight streams / 16-way split-K launches -- latency, not matrix rate -- and the exact-fp32 kernels serve them
#define CONV3_LOW_COST 1.30     // a round of 128-row conv3 tiles against 128/192 of a 192-row round (forward_h2's tile choice)

// k_gemm_h2 for callers outside the network object (the trainer's f16x2 mode): out[M][N] fp32 rows = (A . Wh^T) * scale + shift, A and Wh
// in the h2 layout.  Tile and k-split are chosen from `max_count` (the caller's capacity -- a constant of the trainer, so a
// row's result does not depend on the size of one call): the 256 x 256 ping-pong tile once it fills the chip, else
// 128 x 128 tiles with the k loop split until every CU has a block (raw slabs + the fixed-order fp32 reduce).
int oz_gemm_h2_launch(const void* in_h2, const void* Wh, const float* scale, const float* shift, float* out, const int* d_count, int max_count,
                      int Hin, int Hout, int pad, int Cin, int taps, int N, hipStream_t s, float* partial, long long partial_floats,
                      const void* zero_line, int* flag) {
    OZ_REQUIRE(N % 256 == 0 && Cin % 32 == 0, "gemm_h2: N %% 256 and Cin %% 32 must be 0 (N=%d Cin=%d)", N, Cin);
    static bool attr_set_dev[64] = {};                       // per device: function attributes belong to the device the caller is on
    int dev_now = 0;
    OZ_HIP(hipGetDevice(&dev_now));
    bool& attr_set = attr_set_dev[dev_now & 63];
    if (!attr_set) {
        OZ_HIP(hipFuncSetAttribute((const void*)k_gemm_h2<H2BigPP>, hipFuncAttributeMaxDynamicSharedMemorySize, H2BigPP::LDS));
        OZ_HIP(hipFuncSetAttribute((const void*)k_gemm_h2<H2MidPP>, hipFuncAttributeMaxDynamicSharedMemorySize, H2MidPP::LDS));
        OZ_HIP(hipFuncSetAttribute((const void*)k_gemm_h2<H2LowPP>, hipFuncAttributeMaxDynamicSharedMemorySize, H2LowPP::LDS));
        OZ_HIP(hipFuncSetAttribute((const void*)k_gemm_h2<H2Small>, hipFuncAttributeMaxDynamicSharedMemorySize, H2Small::LDS));
        OZ_HIP(hipFuncSetAttribute((const void*)k_gemm_h2<H2Small2>, hipFuncAttributeMaxDynamicSharedMemorySize, H2Small2::LDS));
        attr_set = true;
    }
    H2Geom g;
    g.Hin = Hin; g.Hout = Hout; g.pad = pad; g.Cin = Cin; g.taps = taps; g.N = N; g.K = taps * Cin; g.out_h2 = 0; g.relu = 0;
    const long long Mmax = (long long)max_count * Hout * Hout;
    g.slab = Mmax * N;
    const long long big_blocks = ((Mmax + 255) / 256) * (N / 256);
    const bool big = big_blocks >= 192;
    // between the two ping-pong tiles, the one whose grid pays fewer tile-rows (rounds of 256 CUs x tile height), as the inference forward
    // picks conv3's: 1024 boards of conv3 = 36864 rows -> 256-row tiles 288 blocks = 2 rounds x 256, 192-row tiles 384 blocks = 2 x 192.
    // Both add every output's products in the same order.
    auto tile_cost = [&](int bm) { return ((((Mmax + bm - 1) / bm) * (N / 256) + 255) / 256) * bm; };
    const bool mid = big && tile_cost(192) < tile_cost(256);
    // (round 5, measured and removed: a few 256 x 256 tiles with an 8 .. 16-way k split for launches of 8 .. 96 big tiles -- the trainer's GEMMs at the
    //  reference's batch -- 45 us + a larger reduce against 40 us on the 128 x 128 tiles: 0.99 -> 1.04 ms per step)
    // launches too small for those: the 128 x 256 tile of the 2-phase ping-pong loop (8 waves, 24 MFMAs per M section) with the k loop split until
    // every CU has a block -- twice the 128 x 128 tile's work per k-tile in about the same time (1.0 against 1.06 us).  Needs a k-slice of >= 8 tiles.
    const bool low = !big && partial != nullptr;
    const int BM = mid ? 192 : big ? 256 : 128, BN = (big || low) ? 256 : 128;
    const int num_mt = (int)((Mmax + BM - 1) / BM);
    int ksplit = 1;
    if (!big && partial) {
        const long long blocks = (long long)num_mt * (N / BN);
        const int nk = g.K / H2_BK;
        // the largest split <= 16 that keeps the grid within one round of the 256 CUs (any number of slices: the k range is divided proportionally)
        while (ksplit < 16 && blocks * (ksplit + 1) <= 256 && nk / (ksplit + 1) >= 8 && (long long)(ksplit + 1) * Mmax * N <= partial_floats) ++ksplit;
    }
    g.ksplit = ksplit;
    const int per_mt = (N / BN) * ksplit;
    const int grid = num_mt < 8 ? ((per_mt + 7) / 8) * 8 * num_mt : ((num_mt + 7) / 8) * 8 * per_mt;       // (the kernel's two block mappings)
    void* dst = ksplit > 1 ? (void*)partial : (void*)out;
    if (mid)
        hipLaunchKernelGGL(k_gemm_h2<H2MidPP>, dim3(grid), dim3(H2MidPP::NT), H2MidPP::LDS, s, (const uint4*)in_h2, (const uint4*)Wh, scale, shift, dst,
                           d_count, g, num_mt, (const uint4*)zero_line, flag, (const unsigned*)nullptr);
    else if (big)
        hipLaunchKernelGGL(k_gemm_h2<H2BigPP>, dim3(grid), dim3(H2BigPP::NT), H2BigPP::LDS, s, (const uint4*)in_h2, (const uint4*)Wh, scale, shift, dst,
                           d_count, g, num_mt, (const uint4*)zero_line, flag, (const unsigned*)nullptr);
    else if (low)
        // (round 6: the 1-phase / 3-stage loop of this tile, H2LowPP1, measured SLOWER here -- 1.00 against 0.93-0.96 ms per step at the reference's batch: the
        //  trainer's k-slices are 8 .. 36 tiles and the deeper prologue, two tiles staged before the first MFMA, costs more than the halved barriers return)
        hipLaunchKernelGGL(k_gemm_h2<H2LowPP>, dim3(grid), dim3(H2LowPP::NT), H2LowPP::LDS, s, (const uint4*)in_h2, (const uint4*)Wh, scale, shift, dst,
                           d_count, g, num_mt, (const uint4*)zero_line, flag, (const unsigned*)nullptr);
    else
        hipLaunchKernelGGL(k_gemm_h2<H2Small>, dim3(grid), dim3(H2Small::NT), H2Small::LDS, s, (const uint4*)in_h2, (const uint4*)Wh, scale, shift, dst,
                           d_count, g, num_mt, (const uint4*)zero_line, flag, (const unsigned*)nullptr);
    if (ksplit > 1) {
        const long long quads = (Mmax * N + 3) / 4;
        hipLaunchKernelGGL(k_splitk_reduce_f32, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, (const float*)partial, g.slab, ksplit, N,
                           Hout * Hout, d_count, scale, shift, 0, out);
    }
    OZ_HIP(hipGetLastError());
    return OZ_OK;
}

// ---------------------------------------------------------------- heads
// one 256-thread block per LP positions whose f2 rows (2 KB each) are staged in LDS once: wave w accumulates k in [128w, 128w+128) of
// logits[a] = f2 . Wpi[:,a] (one policy column per lane, every coalesced Wpi row load feeds LP positions through LDS broadcasts) and of
// v = f2 . Wv; the four partial sums are combined in a fixed order through LDS, then wave w finishes positions w LP/4 .. (softmax, tanh).
// LP = 16 for batches, 8 for the few positions of the latency path: the same sums in the same order either way.
// The policy weights of a wave's 128 reduction indices are fetched 32 rows at a time: the loop is a chain of L2 round trips (~0.6 us each),
// so the depth of a batch IS the launch time -- with 4-8 rows in flight the launch took 20 us for ONE position, as long as conv3's 9.4 MB
// weight stream, and 21 us at 512 positions (round 5: rocprofv3 of the arena and of predict(); 32 in flight: see DESIGN.md).
template <int LP>
__global__ __launch_bounds__(256) void k_heads_t(const float* __restrict__ f2 /*[B][512]*/, const int* __restrict__ d_count,
                                                 int A, const float* __restrict__ Wpi /*[512][A]*/, const float* __restrict__ bpi,
                                                 const float* __restrict__ Wv /*[512]*/, const float* __restrict__ bv,
                                                 float* __restrict__ pi, float* __restrict__ v, OzDeferredReduce dr) {
    __shared__ __attribute__((aligned(16))) float xs[LP][512];
    __shared__ float part[4][LP][64];
    __shared__ float vpart_s[4][LP];
    const int b0 = blockIdx.x * LP, lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), count = *d_count;
    if (b0 >= count) return;
    const bool act = lane < A;
    // wave-uniform row base (scalar registers) + ONE per-lane offset for all the loads below; lanes beyond the A policy columns read column 0
    // (their sums are never used: no select behind every load, which doubled the registers of a batch)
    // (buffer loads: ONE per-lane offset register + a scalar row offset per load.  With plain pointers the compiler kept a 64-bit address pair per load
    //  in flight -- 256 registers for 128 loads, and spilled.)
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wpi + (size_t)w * 128 * A), 0, 128 * A * 4, 0x00020000);
    const int lo = (act ? lane : 0) * 4;
    // the small operands of the tail (value weights, biases) are requested here, with the first batch: every dependent round trip of this one-
    // block-per-few-positions kernel is ~1.5 us of its launch
    const float wv0 = Wv[w * 128 + lane * 2], wv1 = Wv[w * 128 + lane * 2 + 1];
    const float bl = act ? bpi[lane] : 0.f, bvv = bv[0];
    // this wave's first batch of weight rows is requested before the f2 rows are staged: the two round trips overlap.  WB rows per batch:
    // 64 on the latency path (two round trips), 32 for batches of 16 positions (registers)
    constexpr int WB = LP <= 8 ? 64 : 32;
    float wgt[WB];
#pragma unroll
    for (int j = 0; j < WB; ++j) wgt[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, lo, j * A * 4, 0));
    if (dr.ksplit > 1) {
        // fc2 ran as k-slices and left its reduce here (OzDeferredReduce): the rows are staged as k_splitk_reduce_f32 would have written them --
        // slices added in order s = 0, 1, ..., then scale, shift, ReLU: the same bits -- one launch and one round trip through memory less
#pragma unroll 1
        for (int q = threadIdx.x; q < LP * 128; q += 256) {             // 16-byte pieces of the rows that exist (the others are never stored)
            const int p = q >> 7, c4 = (q & 127) * 4;
            if (b0 + p >= count) break;
            const float* src = dr.partial + (size_t)(b0 + p) * 512 + c4;
            f32x4 a = *reinterpret_cast<const f32x4*>(src);
            constexpr int RB = 15;
            for (int s0 = 1; s0 < dr.ksplit; s0 += RB) {
                f32x4 b[RB];
#pragma unroll
                for (int t = 0; t < RB; ++t) if (s0 + t < dr.ksplit) b[t] = *reinterpret_cast<const f32x4*>(src + (size_t)(s0 + t) * dr.slab);
#pragma unroll
                for (int t = 0; t < RB; ++t) if (s0 + t < dr.ksplit) { a[0] += b[t][0]; a[1] += b[t][1]; a[2] += b[t][2]; a[3] += b[t][3]; }
            }
            const f32x4 sc = *reinterpret_cast<const f32x4*>(dr.scale + c4), sh = *reinterpret_cast<const f32x4*>(dr.shift + c4);
            f32x4 r;
#pragma unroll
            for (int t = 0; t < 4; ++t) { const float y = fmaf(a[t], sc[t], sh[t]); r[t] = y < 0.f ? 0.f : y; }
            *reinterpret_cast<f32x4*>(&xs[p][c4]) = r;
        }
    } else
    for (int q = threadIdx.x; q < LP * 128; q += 256) {                 // 16-byte pieces; rows beyond the batch repeat the last one (never stored)
        const int p = q >> 7, c4 = (q & 127) * 4;
        *reinterpret_cast<f32x4*>(&xs[p][c4]) = *reinterpret_cast<const f32x4*>(f2 + (size_t)(b0 + p < count ? b0 + p : count - 1) * 512 + c4);
    }
    __syncthreads();
    float logit[LP], vp[LP];
#pragma unroll
    for (int p = 0; p < LP; ++p) { logit[p] = 0.f; vp[p] = 0.f; }
#pragma unroll 1
    for (int i0 = 0; i0 < 128; i0 += WB) {                   // (rolled, with scheduling fences per position: fully unrolled the compiler hoisted every
        float nxt[WB];                                       //  LDS read of the loop to the top -- 512 VGPRs and 6 KB of scratch per thread, 64 us per launch)
        const bool more = i0 + WB < 128;
#pragma unroll
        for (int j = 0; j < WB; ++j) nxt[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, lo, ((more ? i0 + WB : 0) + j) * A * 4, 0));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < LP; ++p) {
#pragma unroll
            for (int j4 = 0; j4 < WB; j4 += 4) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(&xs[p][w * 128 + i0 + j4]);
#pragma unroll
                for (int j = 0; j < 4; ++j) logit[p] = fmaf(x[j], wgt[j4 + j], logit[p]);
            }
            if ((p & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < WB; ++j) wgt[j] = nxt[j];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float wv = i == 0 ? wv0 : wv1;
#pragma unroll
        for (int p = 0; p < LP; ++p) vp[p] = fmaf(xs[p][w * 128 + lane * 2 + i], wv, vp[p]);
    }
#pragma unroll
    for (int p = 0; p < LP; ++p) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vp[p] += __shfl_xor(vp[p], off, 64);
        part[w][p][lane] = logit[p];
        if (lane == 0) vpart_s[w][p] = vp[p];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < LP / 4; ++q) {
        const int p = w * (LP / 4) + q;
        const float sum = ((part[0][p][lane] + part[1][p][lane]) + part[2][p][lane]) + part[3][p][lane];
        const float lg = act ? sum + bl : -INFINITY;
        float mx = lg;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        const float e = act ? expf(lg - mx) : 0.f;
        float s = e;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (b0 + p < count) {
            if (act) pi[(size_t)(b0 + p) * A + lane] = e / s;
            if (lane == 0) v[b0 + p] = tanhf(((vpart_s[0][p] + vpart_s[1][p]) + vpart_s[2][p]) + vpart_s[3][p] + bvv);
        }
    }
}
#define HEADS_P 8
#define HEADS_LP 16

// ---------------------------------------------------------------- stub evaluator (test nets)
__global__ __launch_bounds__(64) void k_stub(const uint64_t* __restrict__ own, const uint64_t* __restrict__ opp,
                                             const int* __restrict__ d_count, int n, uint64_t salt, uint64_t keep,
                                             float* __restrict__ pi, float* __restrict__ v) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= *d_count) return;
    const uint64_t o = own[b], p = opp[b];
    const int r = lane >> 3, c = lane & 7;
    const bool inb = r < n && c < n;
    uint32_t w = 0;
    if (inb) {
        const uint64_t u = oz_stub_h(o, p, salt, (uint64_t)lane);
        w = (uint32_t)((u >> 40) & 0xFFFF);
        if (((u >> 8) & keep) != 0) w = 0;
    }
    uint32_t sum = w;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (sum == 0) { if (lane == 0) w = 1; sum = 1; }
    if (inb) pi[(size_t)b * n * n + r * n + c] = (float)w / (float)sum;
    if (lane == 0) {
        const int32_t m = (int32_t)(oz_stub_h(o, p, salt, 64) >> 40);
        v[b] = (float)(m - 8388608) / 8388608.0f;
    }
}

// ================================================================ host objects
struct StubNet : oz_net {
    uint64_t salt = 0, keep = 0;
    int forward_device(const uint64_t* d_own, const uint64_t* d_opp, const int* d_count, int max_count, float* d_pi,
                       float* d_v, hipStream_t s) override {
        hipLaunchKernelGGL(k_stub, dim3(max_count), dim3(64), 0, s, d_own, d_opp, d_count, n, salt, keep, d_pi, d_v);
        OZ_HIP(hipGetLastError());
        return OZ_OK;
    }
};

// Calibration positions of precision f16x2 (oz_net_commit): mover-canonical boards of uniformly random playouts from the opening
// (Othello/__init__.py:177-184), every position before a move, games back to back until `total` are collected -- every ply of the game
// is represented.  A fixed function of (n, total): the same set on every rank, every host, every commit.
static void calib_positions(int n, int total, std::vector<uint64_t>& own, std::vector<uint64_t>& opp) {
    const uint64_t valid = oz_valid_mask(n);
    const int h = n / 2;
    uint64_t r = 0x0CA11B8A7E5EEDull + (uint64_t)n;
    own.clear(); opp.clear();
    while ((int)own.size() < total) {
        uint64_t white = (1ULL << ((h - 1) * 8 + h - 1)) | (1ULL << (h * 8 + h)), black = (1ULL << ((h - 1) * 8 + h)) | (1ULL << (h * 8 + h - 1));
        int player = 1, finished = 0;
        while (!finished && (int)own.size() < total) {
            const uint64_t mine = player == 1 ? black : white, theirs = player == 1 ? white : black;
            own.push_back(mine); opp.push_back(theirs);
            const uint64_t legal = oz_legal(mine, theirs, valid);
            if (!legal) break;                                   // cannot happen before `finished` (oz_game_play passes), defensive
            r = oz_sm64(r);
            oz_game_play(black, white, player, finished, oz_kth_bit(legal, (int)(r % (uint64_t)oz_popc(legal))), valid);
        }
    }
}
// power-of-two exponents that move the maxima mx[] into [2^(top-1), 2^top); entries without a maximum (a channel that is never active on
// the calibration set, an all-zero weight column) take the median exponent of the others
static void pick_exponents(const std::vector<float>& mx, int top, std::vector<int>& e) {
    e.assign(mx.size(), 0);
    std::vector<int> live;
    for (size_t c = 0; c < mx.size(); ++c)
        if (mx[c] > 0.f) {
            int ex = 0;
            frexpf(mx[c], &ex);
            e[c] = std::min(60, std::max(-60, top - ex));
            live.push_back(e[c]);
        }
    int med = 0;
    if (!live.empty()) { std::nth_element(live.begin(), live.begin() + live.size() / 2, live.end()); med = live[live.size() / 2]; }
    for (size_t c = 0; c < mx.size(); ++c) if (!(mx[c] > 0.f)) e[c] = med;
}

struct OnnNet : oz_net {
    int F = 0, A = 0;
    int cin = 2;             // input planes: 2 = OthelloNN (own, opp), 1 = BaseNN (own - opp, Net/BaseNN.py:41-44)
    std::vector<std::vector<float>> w;      // 40 arrays, keras get_weights() order
    bool committed = false;
    // device
    float *d_w1 = nullptr, *d_wt[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // conv2..4, fc1, fc2 as [N][K]
    float *d_scale[6] = {}, *d_shift[6] = {};
    float *d_wpi = nullptr, *d_bpi = nullptr, *d_wv = nullptr, *d_bv = nullptr;
    float *act1 = nullptr, *act2 = nullptr, *act3 = nullptr, *act4 = nullptr, *f1 = nullptr, *f2 = nullptr;
    // precision 1 ("f32 via 2 x fp16 split", oz_net_h2.h): conv2..4, fc1, fc2 weights in the h2 layout, column c scaled by 2^wexp[c] and row k by
    // 2^-aexp[channel of k] (commit_h2); d_scale_h2 / d_shift_h2 carry the inverse powers and the output tensor's exponents
    int precision = 0;
    uint4* d_wh[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};          // conv2..4, fc1, fc2
    float* d_scale_h2[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int* d_flag = nullptr;
    float* d_partial = nullptr;      // split-K slabs [4][max_batch][1024]
    float* d_part32 = nullptr;       // precision f32, max_batch <= 32: split-K slabs of the latency path
    uint4* d_zero = nullptr;
    // conv1 as a lookup (H2BigPPLut): the OZ_LUT_ROWS possible conv1 output rows and the per-pixel pattern ids of a batch
    uint4* d_lut = nullptr;
    unsigned* d_lut_ids = nullptr;    // per-pixel pattern ids of a batch, padded boards [max_batch][(n + 2)^2] (k_lut_ids)
    bool lut_ok = false;
    float* d_t2 = nullptr;           // conv2 as a gather-sum (k_conv2_lut): [9][OZ_LUT_ROWS][C], the last row of every tap = zeros
    uint4* d_wtap = nullptr;         // build_t2 staging: conv2's kernel as nine [C][C] matrices in the h2 layout
    float* d_raw = nullptr;          // oz_net_commit staging: one Keras kernel as stored
    float* d_lut32 = nullptr;        // precision f32: conv1 pattern table [OZ_LUT_PATTERNS][C] fp32
    float* d_wtap32 = nullptr;       // precision f32: conv2's kernel as nine [C(out)][C(in)] matrices
    bool t2f_ok = false;             // precision f32: d_t2 holds the fp32 T2 tables
    float *d_one = nullptr, *d_nul = nullptr;
    int* d_rows = nullptr;
    bool t2_ok = false;
    std::vector<void*> allocs;
    // HIP-event timing on the launch stream (oz_net_profile): mode 1 = the dominant launch only (bench.py's timed region),
    // mode 2 = every kernel of the forward (slot order: OZ_NET_KERNELS in the header)
    int profile = 0;
    OzTimer timer{OZ_NET_KERNELS};
    int tables_mode = -1;            // oz_net_set_tables: -1 = default (2), 0 / 1 / 2 see forward_h2
    int sizing() const { return max_batch; }     // the batch size the k-splits of the medium path are chosen for: the capacity, a per-network constant
    size_t partial_cap = 0;          // floats d_partial holds
    bool latency_splits = false;     // oz_net_set_option(OZ_NET_OPT_LATENCY_SPLITS): see conv_ksplit
    int conv3_tile = 0;              // oz_net_set_option(OZ_NET_OPT_CONV3_TILE): 0 = the forward picks, 128 / 192 / 256 = that tile (bit-identity screen of the three)
    bool f32_std_tile = false;       // oz_net_set_option(OZ_NET_OPT_F32_STD_TILE): precision f32 never takes the 256 x 256 tile (bit-identity screen)
    bool simple_loop = false;        // oz_net_set_option(OZ_NET_OPT_SIMPLE_LOOP): one-barrier-per-k-tile loop for the 3x3 layers (race screen)
    int b3_tile = 0;                 // oz_net_set_option(OZ_NET_OPT_B3_TILE): 0 = the launcher picks (256 x 256 where its grid fills the chip), 128 / 256 = that tile (bit-identity screen)
    int low_loop_phases = 1;         // oz_net_set_option(OZ_NET_OPT_LOW_LOOP_PHASES): main loop of the 128 x 256 tile -- 1 (default) = one phase per k-tile, three LDS stages; 2 = round 5's 2-phase loop
    float* d_t2rows = nullptr;       // commit staging: one tap's T2 rows [OZ_LUT_PATTERNS][C] before the slice-major re-layout
    int last_conv3_rows = 0;         // row-tile height the last forward ran conv3 on (oz_net_get_info)
    int profiled_layer = 2;          // 2 = conv2 GEMM, 3 = conv3 GEMM (when conv2 runs as the table gather-sum)
    // precision f16x2, scaling and guards (oz_net_h2.h, header): tensor t = 0..4 is act1, act2, act3, act4, f1; layer i = 0..4 is conv2..fc2
    std::vector<int> aexp[5];        // per-channel activation exponents of tensor t (exact powers of two, from the calibration maxima)
    std::vector<int> wexp[5];        // per-column weight exponents of layer i
    std::vector<float> bn_sc[6], bn_sh[6];     // folded BN scale / shift of conv1..fc2 (host copies)
    int* d_aexp[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int* d_wexp = nullptr;           // commit staging: the column exponents of the layer being prepared
    unsigned* d_colmax = nullptr;    // commit staging: per-channel maxima as bit patterns
    float *d_scale1_h2 = nullptr, *d_shift1_h2 = nullptr;        // conv1's BN scale / shift times 2^aexp[0]
    float* d_shift_h2[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};       // shift of layer i times its output tensor's exponents
    unsigned* d_lowcnt[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};      // low-side guard: per-row counters of tensor t (H2Low)
    unsigned char* d_lut_low = nullptr;                          // conv1 table rows that are low (k_lut_rowlow)
    unsigned fwd_seq = 0;            // forward number of the guard counters (26 bits, never 0)
    int act_target_log2 = H2_ACT_TOP;   // OZ_NET_OPT_ACT_TARGET_LOG2: calibration maxima land in [2^(target-1), 2^target)
    int w_target_log2 = H2_W_TOP;    // OZ_NET_OPT_W_TARGET_LOG2: every weight column's largest |w| lands in [2^(target-1), 2^target)
    int low_guard_log2 = H2_LOW_GUARD;  // OZ_NET_OPT_LOW_GUARD_LOG2 (<= -100: guard off); committed value below
    float low_thr = 0.f;
    H2Low next_low;                  // guard of the NEXT launch_gemm_h2 (consumed by it)
    int next_relu = 1;               // 0: the NEXT launch_gemm_h2 writes the BN output without the ReLU (calibration passes; consumed by it)
    int self_check = 1;              // OZ_NET_OPT_SELF_CHECK: compare with the exact-fp32 kernels on the calibration positions at commit
    double sc_dpi = -1.0, sc_dv = -1.0;                          // what the last commit's self-check measured (oz_net_self_check)
    int sc_positions = 0;
    int sc_guard = 0;                // H2_FLAG_* bits a guard raised on the calibration positions during the last self-check (OZ_NET_INFO_SELF_CHECK_GUARD)
    float *d_sc_out = nullptr;       // self-check outputs: [2][cal_total][A + 1]
    uint64_t *d_cal_own = nullptr, *d_cal_opp = nullptr;          // calibration positions (calib_positions), resident
    int* d_cal_count = nullptr;
    int cal_total = 0;

    // precision 2 ("f32 via 3 x bf16 split", oz_net_b3.h): conv3, conv4 and fc1 on k_gemm_b3 (weights and activations in the b3 layout, 6 B per
    // element); conv1 + conv2 from the exact-fp32 pattern tables, fc2 and the heads on the exact-fp32 kernels.  Networks below B3_MIN_BATCH
    // positions of capacity run the exact-fp32 forward as it is (their layers are weight streams / split-K launches: latency, not matrix rate) --
    // a per-network constant, so a position's result does not depend on the size of the call it sits in.
    uint4* d_wb[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // conv2 (the all-GEMM form only), conv3, conv4, fc1, fc2
    uint4* b3a1 = nullptr;           // conv1's output in the b3 layout (the all-GEMM form only: allocated at its first use)
    uint4 *b3a2 = nullptr, *b3a3 = nullptr, *b3a4 = nullptr, *b3f1 = nullptr;   // conv2 / conv3 / conv4 / fc1 outputs in the b3 layout
    float* d_part_b3 = nullptr;                              // fc1's k-slices (fixed-order fp32 reduce)
    bool use_b3() const { return precision == 2 && max_batch >= B3_MIN_BATCH; }
    int fc1_b3_ksplit() const {                              // k-slices of fc1 (1024 columns = 4 column tiles): until the grid fills the chip, at most 8
        const long long blocks = (((long long)max_batch + B3_BM - 1) / B3_BM) * 4;
        int k = 1;
        while (k < 8 && blocks * k < 192) k *= 2;
        return k < 2 ? 2 : k;                                // (always split: its fixed-order reduce is what writes fc2's b3 operand)
    }
    // medium networks (128 <= max_batch < ~1000): a 3x3 layer whose grid would leave most CUs idle splits its k loop -- the smallest power of two <= 8 that
    // brings the grid of 128-row tiles to >= 192 blocks, from max_batch (a per-network constant); 1 from 1024 positions of 8x8 on (the bench)
    int conv_b3_ksplit(int pixels) const {
        const long long blocks = (((long long)max_batch * pixels + B3_BM - 1) / B3_BM) * (C / B3_BN);
        int k = 1;
        while (k < 8 && blocks * k < 192) k *= 2;
        return k;
    }
    int fc2_b3_ksplit() const {                              // fc2 (512 columns = 2 column tiles, 32 k-tiles): 4 slices of 8 k-tiles; the heads kernel adds them
        return 4;
    }

    template <typename T> int alloc(T** p, size_t count) {
        OZ_HIP(hipMalloc((void**)p, sizeof(T) * (count ? count : 1)));
        allocs.push_back(*p);
        return OZ_OK;
    }
    std::vector<int64_t> sizes() const {
        std::vector<int64_t> s;
        const int cins[4] = {cin, C, C, C};
        for (int l = 0; l < 4; ++l) { s.push_back(9ll * cins[l] * C); for (int k = 0; k < 5; ++k) s.push_back(C); }
        s.push_back((int64_t)F * 1024); for (int k = 0; k < 5; ++k) s.push_back(1024);
        s.push_back(1024ll * 512); for (int k = 0; k < 5; ++k) s.push_back(512);
        s.push_back(512ll * A); s.push_back(A); s.push_back(512); s.push_back(1);
        return s;
    }
    ~OnnNet() override {
        hipSetDevice(device);
        timer.destroy();
        for (void* p : allocs) hipFree(p);
    }

    int launch_gemm(const float* in, const float* Wt, int layer, float* out, const int* d_count, int max_count, int Hin,
                    int Hout, int pad, int Cin, int taps, int N, hipStream_t s, OzDeferredReduce* defer = nullptr) {
        // small and medium networks (max_batch <= 512) split K over the idle CUs: latency, not throughput
        return oz_gemm_f32_launch(in, Wt, d_scale[layer], d_shift[layer], out, d_count, max_count, Hin, Hout, pad, Cin, taps, N, 1, s,
                                  d_part32, d_part32 ? (long long)part32_floats() : 0, sizing(), 0, -1,
                                  layer == 2 ? &last_conv3_rows : nullptr, f32_std_tile ? 1 : 0, defer);  // layer 2 = conv3: what oz_net_get_info reports
    }

    // precision f32: split-K slabs per position-row budget (small networks 16 slices, medium ones fewer; none for large batches)
    int part32_mult() const { return max_batch <= 32 ? 16 : max_batch <= 128 ? 8 : max_batch <= 512 ? 2 : 0; }
    // ... in floats; layers of at most 64 rows run as weight streams with up to K / 64 = 128 k-slices of 64 rows x 1024 columns
    size_t part32_floats() const {
        const size_t split = (size_t)part32_mult() * max_batch * 64 * 512, stream = max_batch <= 64 ? (size_t)128 * 64 * 1024 : 0;
        return split > stream ? split : stream;
    }
    // k-split of a 3x3 convolution on BM-row tiles (N = C, 256-column tiles): the smallest power of two <= 8 that brings
    // the grid to >= 192 blocks, from max_batch (a per-network constant, so results do not depend on the size of a call)
    // OZ_NET_OPT_LATENCY_SPLITS (round 5, default off): ... or the split a cost model prefers -- rounds of the 256 CUs x k-tiles per block + the
    // fixed-order reduce's slabs.  Written for max_batch = 512 on 8x8 (the arena's networks): conv3 = 192 blocks of 144 k-tiles = 205 us per
    // launch WHATEVER the batch holds; 4 k-slices = 768 blocks of 36 k-tiles.  Measured (bench.py's config5 leg, 512 games x 800 sims): with
    // every expansion evaluated (512-leaf batches) the split LOSES, 407 -> 430 us per step (the slab epilogues and the reduce cost more than
    // the idle quarter of the chip), so it is not the default; with the library's cross-game de-duplication + evaluation cache (batches of
    // ~15 leaves) it is what the step waits for: 31.3 -> 44.8 games/s.  Every call of a network uses the same split either way.
    int conv_ksplit(int pixels, int BM) const {
        const long long blocks = (((long long)sizing() * pixels + BM - 1) / BM) * (C / 256);
        int k = 1;
        while (k < 8 && blocks * k < 192) k *= 2;
        if (latency_splits && sizing() > 32 && blocks * k < 1024) {
            const double nk = 9.0 * C / 32, t_kt = 1.42e-6 * BM / 192.0, t_slab = (double)sizing() * pixels * C * 4.0 / 8.0e12 + 1.0e-6;
            auto cost = [&](int q) { return (double)((blocks * q + 255) / 256) * (nk / q) * t_kt + (q > 1 ? q * t_slab : 0.0); };
            int best = k;
            for (int q = 1; q <= 8; q *= 2)
                if (nk / q >= 8 && cost(q) < 0.95 * cost(best)) best = q;
            k = best;
        }
        return k;
    }
    // conv4 of a medium network (max_batch <= 1024) whose rows fit one grid round of 128 x 256 tiles runs on that tile (the 2-phase ping-pong
    // loop) with half the k-slices the 256-row tile would need to fill the chip: the arena's 512-game networks 2 slices of 72 k-tiles instead of
    // 4 of 36 (83 + 14 us -> see DESIGN.md); a per-network constant like every k-split
    bool conv4_low() const {
        const long long rows = (long long)sizing() * (n - 4) * (n - 4);
        return sizing() > 32 && sizing() <= 1024 && ((rows + 127) / 128) * (C / 256) <= 256;
    }
    size_t partial_floats() const {
        if (max_batch <= 32) return (size_t)16 * max_batch * 64 * 1024;
        size_t need = (size_t)(sizing() >= 1024 ? 4 : 16) * max_batch * 1024;        // fc1 (forward_h2: kfc1)
        const int px[3] = {n * n, (n - 2) * (n - 2), (n - 4) * (n - 4)}, bm[3] = {256, 192, conv4_low() ? 128 : 256};
        for (int i = 0; i < 3; ++i) {
            int k = conv_ksplit(px[i], bm[i]);
            if (i == 1 && conv_ksplit(px[i], 256) > k) k = conv_ksplit(px[i], 256);          // conv3 may run on either tile
            if (k > 1 && (size_t)k * max_batch * px[i] * C > need) need = (size_t)k * max_batch * px[i] * C;
        }
        return need;
    }

    // layer: 1..3 = conv2..4 (3x3, Cin = N = C), 4 = fc1, 5 = fc2 (taps 1)
    template <typename CF, int TAG = 0>
    int launch_gemm_h2(const void* in, int layer, void* out, int out_h2, const int* d_count, int max_count, int Hin,
                       int Hout, int pad, int Cin, int taps, int N, hipStream_t s, int ksplit = 1,
                       const unsigned* lut_ids = nullptr, const uint4* w_alt = nullptr, const float* scale_alt = nullptr,
                       const float* shift_alt = nullptr, int relu = 1) {
        H2Geom g;
        g.Hin = Hin; g.Hout = Hout; g.pad = pad; g.Cin = Cin; g.taps = taps; g.N = N; g.K = taps * Cin; g.out_h2 = out_h2; g.relu = relu;
        g.ksplit = ksplit; g.slab = (long long)max_batch * Hout * Hout * N;
        const H2Low low = next_low;               // the guard of this launch's h2 output, if the caller armed one
        next_low = H2Low();
        if (!next_relu) { relu = 0; g.relu = 0; next_relu = 1; }          // calibration pass: the layer's BN output before the ReLU
        if (out_h2 && ksplit == 1) g.low = low;
        const long long Mmax = (long long)max_count * Hout * Hout;
        const int num_mt = (int)((Mmax + CF::BM - 1) / CF::BM);
        const int per_mt = (N / CF::BN) * ksplit;
        const int grid = num_mt < 8 ? ((per_mt + 7) / 8) * 8 * num_mt : ((num_mt + 7) / 8) * 8 * per_mt;   // (the kernel's two block mappings)
        void* out_final = out;
        if (ksplit > 1) out = d_partial;          // raw k-slice sums; k_splitk_reduce_h2 below writes out_final (h2 layout)
        {   // the dynamic-LDS limit of THIS instantiation, once per device
            static bool attr_done[64] = {};
            if (!attr_done[device & 63]) {
                OZ_HIP(hipFuncSetAttribute((const void*)k_gemm_h2<CF, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, CF::LDS + (CF::LUT ? 9 * CF::BM * 2 : 0)));
                attr_done[device & 63] = true;
            }
        }
        hipLaunchKernelGGL((k_gemm_h2<CF, TAG>), dim3(grid), dim3(CF::NT), CF::LDS + (CF::LUT ? 9 * CF::BM * 2 : 0), s, (const uint4*)in,
                           w_alt ? w_alt : (const uint4*)d_wh[layer - 1], scale_alt ? scale_alt : d_scale_h2[layer - 1],
                           shift_alt ? shift_alt : d_shift_h2[layer - 1], out, d_count, g, num_mt, d_zero, d_flag, lut_ids);
        if (ksplit > 1 && out_h2) {
            const long long threads = (long long)max_count * Hout * Hout * (N / 8);
            hipLaunchKernelGGL(k_splitk_reduce_h2, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, (const float*)d_partial,
                               g.slab, ksplit, N, Hout * Hout, d_count, d_scale_h2[layer - 1], d_shift_h2[layer - 1], 1, (uint4*)out_final, d_flag, low);
        } else if (ksplit > 1 && layer == 5 && relu && !scale_alt) {      // fc2: the heads kernel adds the slices (launch_heads)
            fc2_defer.partial = d_partial; fc2_defer.slab = g.slab; fc2_defer.ksplit = ksplit;
            fc2_defer.scale = d_scale_h2[layer - 1]; fc2_defer.shift = d_shift_h2[layer - 1];
        } else if (ksplit > 1) {                  // fp32 rows out (calibration passes): the fp32 path's fixed-order reduce
            const long long quads = ((long long)max_count * Hout * Hout * N + 3) / 4;
            hipLaunchKernelGGL(k_splitk_reduce_f32, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, (const float*)d_partial, g.slab, ksplit, N,
                               Hout * Hout, d_count, scale_alt ? scale_alt : d_scale_h2[layer - 1], shift_alt ? shift_alt : d_shift_h2[layer - 1], relu,
                               (float*)out_final);
        }
        OZ_HIP(hipGetLastError());
        return OZ_OK;
    }

    // small tiles (dense layers, latency path): three LDS stages for medium and large networks (-4 .. -6 % per forward at 128 .. 512
    // positions, +-0 at 4096), the two-stage loop for the 16-way split-K launches of small networks, where the deeper pipeline
    // measured SLOWER (one position: 0.152 -> 0.186 ms).  Bit-identical either way (tools/pp_race_check.py compares networks of both kinds).
    template <typename CF3, typename CF2, typename... Args> int launch_small(Args... args) {
        return max_batch <= 32 ? launch_gemm_h2<CF2>(args...) : launch_gemm_h2<CF3>(args...);
    }

    // policy / value heads: the largest batches stage the f2 rows of 16 positions per block in LDS, smaller ones 4 (a block's time is its chain of
    // L2 round trips, not its arithmetic: 512 positions 21.6 us on 32 blocks, 11.0 us on 128; 2048: 22 -> 14 us; 3640: 23 us either way); one
    // block of 8 for the few positions of the latency path (same sums in the same order in all three)
    OzDeferredReduce fc2_defer;       // set by the fc2 launch of a forward when its k-slices are reduced by the heads kernel
    void launch_heads(int max_count, const int* d_count, float* d_pi, float* d_v, hipStream_t s) {
        const OzDeferredReduce dr = fc2_defer;
        fc2_defer = OzDeferredReduce();
        if (max_count >= 3072) hipLaunchKernelGGL(k_heads_t<HEADS_LP>, dim3((max_count + HEADS_LP - 1) / HEADS_LP), dim3(256), 0, s, f2, d_count, A, d_wpi, d_bpi, d_wv, d_bv, d_pi, d_v, dr);
        else if (max_count > HEADS_P) hipLaunchKernelGGL(k_heads_t<4>, dim3((max_count + 3) / 4), dim3(256), 0, s, f2, d_count, A, d_wpi, d_bpi, d_wv, d_bv, d_pi, d_v, dr);
        else hipLaunchKernelGGL(k_heads_t<HEADS_P>, dim3((max_count + HEADS_P - 1) / HEADS_P), dim3(256), 0, s, f2, d_count, A, d_wpi, d_bpi, d_wv, d_bv, d_pi, d_v, dr);
    }

    // conv1 + conv2 as the table gather-sum: one table slice per XCD at 512 filters, the thread-per-(pixel, 8 channels) kernel otherwise
    template <int OUT_H2> void launch_conv2_lut(int max_count, const int* d_count, const float* scale, const float* shift, void* out, hipStream_t s,
                                                 H2Low low = H2Low(), float floor = 0.f) {
        const long long pixels = (long long)max_count * n * n;
        if (C == 512) {
            const unsigned blocks = 8u * (unsigned)((pixels + 32 * OZ_C2L_PPT - 1) / (32 * OZ_C2L_PPT));
            if (n == 8) hipLaunchKernelGGL((k_conv2_lut_xcd<8, OUT_H2>), dim3(blocks), dim3(256), 0, s, d_lut_ids, d_count, d_t2, scale, shift, out, d_flag, low, floor);
            else hipLaunchKernelGGL((k_conv2_lut_xcd<6, OUT_H2>), dim3(blocks), dim3(256), 0, s, d_lut_ids, d_count, d_t2, scale, shift, out, d_flag, low, floor);
        } else {
            const long long threads = pixels * (C / 8);
            hipLaunchKernelGGL(k_conv2_lut<OUT_H2>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, d_lut_ids, d_count, n, C, d_t2, scale, shift,
                               out, d_flag, low, floor);
        }
    }
    // the guard of tensor t (0 = act1 .. 4 = f1) for this forward; off in calibration passes and when the option disables it
    H2Low low_of(int t, bool on) const {
        H2Low lo;
        if (on && low_thr > 0.f && d_lowcnt[t]) { lo.cnt = d_lowcnt[t]; lo.seq = fwd_seq; lo.thr = low_thr; lo.flag = d_flag; }
        return lo;
    }

    // T2[t] = table . W_t^T (raw k-sums, column c scaled 2^wexp[0][c] like the convolution's): nine GEMMs M = OZ_LUT_PATTERNS, K = N = C
    int build_t2() {
        // d_wtap (conv2's kernel as nine [C][C] matrices in the h2 layout, same power-of-two scale as d_wh[0]) was written by oz_net_commit
        if (!d_one) {
            if (int rc = alloc(&d_one, (size_t)C)) return rc;
            if (int rc = alloc(&d_nul, (size_t)C)) return rc;
            if (int rc = alloc(&d_rows, (size_t)1)) return rc;
            std::vector<float> one((size_t)C, 1.0f);
            const int rows = OZ_LUT_PATTERNS;
            OZ_HIP(hipMemcpy(d_one, one.data(), sizeof(float) * C, hipMemcpyHostToDevice));
            OZ_HIP(hipMemset(d_nul, 0, sizeof(float) * C));
            OZ_HIP(hipMemcpy(d_rows, &rows, sizeof(int), hipMemcpyHostToDevice));
        }
        if (!d_t2) { if (int rc = alloc(&d_t2, (size_t)9 * OZ_LUT_ROWS * C)) return rc; }
        int rc = OZ_OK;
        if (!d_t2rows) { if (int rc2 = alloc(&d_t2rows, (size_t)OZ_LUT_PATTERNS * C)) return rc2; }
        for (int t = 0; t < 9 && rc == OZ_OK; ++t) {        // one tap at a time: GEMM rows -> staging -> slice-major records (same stream: ordered)
            rc = launch_gemm_h2<H2BigPP>(d_lut, 1, d_t2rows, 0, d_rows, OZ_LUT_PATTERNS, 1, 1, 0, C, 1, C, 0, 1, nullptr,
                                         d_wtap + (size_t)t * C * (C / 4), d_one, d_nul, 0);
            const long long quads = (long long)OZ_LUT_ROWS * C / 4;
            hipLaunchKernelGGL(k_t2_to_slices, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, 0, d_t2rows, C, t, d_t2);
        }
        OZ_HIP(hipDeviceSynchronize());
        if (rc == OZ_OK) t2_ok = true;
        return rc;
    }

    // precision f32: the same tables in exact fp32 arithmetic (rows by k_lut_build_f32, T2 by nine fp32 MFMA GEMMs)
    int build_t2_f32() {
        if (!d_lut32) { if (int rc = alloc(&d_lut32, (size_t)OZ_LUT_PATTERNS * C)) return rc; }
        if (!d_lut_ids) { if (int rc = alloc(&d_lut_ids, (size_t)max_batch * (n + 2) * (n + 2))) return rc; }
        if (!d_wtap32) { if (int rc = alloc(&d_wtap32, (size_t)9 * C * C)) return rc; }
        if (!d_one) {
            if (int rc = alloc(&d_one, (size_t)C)) return rc;
            if (int rc = alloc(&d_nul, (size_t)C)) return rc;
            if (int rc = alloc(&d_rows, (size_t)1)) return rc;
            std::vector<float> one((size_t)C, 1.0f);
            const int rows = OZ_LUT_PATTERNS;
            OZ_HIP(hipMemcpy(d_one, one.data(), sizeof(float) * C, hipMemcpyHostToDevice));
            OZ_HIP(hipMemset(d_nul, 0, sizeof(float) * C));
            OZ_HIP(hipMemcpy(d_rows, &rows, sizeof(int), hipMemcpyHostToDevice));
        }
        if (!d_t2) { if (int rc = alloc(&d_t2, (size_t)9 * OZ_LUT_ROWS * C)) return rc; }
        if (!d_t2rows) { if (int rc2 = alloc(&d_t2rows, (size_t)OZ_LUT_PATTERNS * C)) return rc2; }
        const long long threads = (long long)OZ_LUT_PATTERNS * C;
        hipLaunchKernelGGL(k_lut_build_f32, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, C, d_w1, d_scale[0], d_shift[0], d_lut32);
        OZ_HIP(hipGetLastError());
        for (int t = 0; t < 9; ++t) {
            if (int rc = oz_gemm_f32_launch(d_lut32, d_wtap32 + (size_t)t * C * C, d_one, d_nul, d_t2rows, d_rows,
                                            OZ_LUT_PATTERNS, 1, 1, 0, C, 1, C, 0, 0, nullptr, 0, 0)) return rc;
            const long long quads = (long long)OZ_LUT_ROWS * C / 4;
            hipLaunchKernelGGL(k_t2_to_slices, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, 0, d_t2rows, C, t, d_t2);
        }
        OZ_HIP(hipDeviceSynchronize());
        t2f_ok = true;
        return OZ_OK;
    }

    template <typename T> int up(T** dst, const std::vector<T>& h) {
        if (!*dst) { if (int rc = alloc(dst, h.size())) return rc; }
        OZ_HIP(hipMemcpy(*dst, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
        return OZ_OK;
    }
    int read_colmax(int count, std::vector<float>& mx, const char* what) {
        mx.resize((size_t)count);
        OZ_HIP(hipMemcpy(mx.data(), d_colmax, sizeof(float) * (size_t)count, hipMemcpyDeviceToHost));      // bit patterns of non-negative floats
        for (float x : mx)
            if (!(x <= 3.0e38f)) {
                oz_set_error("oz_net_commit (precision f16x2): %s is not finite on the calibration positions; the weights are unusable", what);
                return OZ_ERR_STATE;
            }
        return OZ_OK;
    }

    // precision f16x2: operands, scaling and guards (header of oz_net_h2.h).  Layer by layer, because a layer's weight image needs the
    // exponents of the tensor it consumes and its output exponents need its own calibration pass:
    //   conv1: per-channel supremum over all 3^9 patterns -> aexp[0], the scaled conv1 table, its low rows;
    //   layer i = conv2 .. fc2: column maxima of w / 2^aexp_in -> wexp[i], the h2 weight image, (conv2: the T2 tables,) then the layer's output on
    //   the calibration positions as fp32 rows -> per-channel maxima -> aexp[i + 1], folded into the layer's scale and shift.
    int commit_h2() {
        const int gl[5] = {6, 12, 18, 24, 30};
        const int Ks[5] = {9 * C, 9 * C, 9 * C, F, 1024}, Ns[5] = {C, C, C, 1024, 512};
        const int Wt[5] = {C, C, C, C, 1024};                                          // channels of tensor t
        const size_t rows_t[5] = {(size_t)max_batch * n * n, (size_t)max_batch * n * n, (size_t)max_batch * (n - 2) * (n - 2),
                                  (size_t)max_batch * (n - 4) * (n - 4), (size_t)max_batch};
        const int wide = std::max(C, 1024);
        if (!d_flag) { if (int rc = alloc(&d_flag, 1)) return rc; }
        OZ_HIP(hipMemset(d_flag, 0, sizeof(int)));
        // split-K slabs: fc1 (4 x max_batch x 1024); small networks also split the convolutions 16 ways (latency path)
        if (!d_partial || partial_floats() > partial_cap) {
            d_partial = nullptr;
            if (int rc = alloc(&d_partial, partial_floats())) return rc;
            partial_cap = partial_floats();
        }
        if (!d_zero) { if (int rc = alloc(&d_zero, 16)) return rc; }
        OZ_HIP(hipMemset(d_zero, 0, 256));
        if (!d_colmax) { if (int rc = alloc(&d_colmax, (size_t)wide)) return rc; }
        if (!d_wexp) { if (int rc = alloc(&d_wexp, (size_t)wide)) return rc; }
        for (int t = 0; t < 5; ++t)
            if (!d_lowcnt[t]) {
                if (int rc = alloc(&d_lowcnt[t], rows_t[t])) return rc;
                OZ_HIP(hipMemset(d_lowcnt[t], 0, sizeof(unsigned) * rows_t[t]));
            }
        low_thr = low_guard_log2 <= -100 ? 0.f : ldexpf(1.0f, low_guard_log2);
        if (!d_cal_own) {
            cal_total = max_batch >= 64 ? 512 : 64;
            std::vector<uint64_t> own, opp;
            calib_positions(n, cal_total, own, opp);
            if (int rc = up(&d_cal_own, own)) return rc;
            if (int rc = up(&d_cal_opp, opp)) return rc;
            if (int rc = alloc(&d_cal_count, 1)) return rc;
        }
        std::vector<float> mx, sc, sh;
        struct ProfileOff {                                  // calibration launches are not part of anybody's timing
            int& p; int keep;
            explicit ProfileOff(int& q) : p(q), keep(q) { p = 0; }
            ~ProfileOff() { p = keep; }
        } profile_off(profile);
        float* const act_of[4] = {act2, act3, act4, f1};
        const int P_of[4] = {n * n, (n - 2) * (n - 2), (n - 4) * (n - 4), 1};
        // Pass 0 calibrates (maxima -> [2^(H2_ACT_TOP - 1), 2^H2_ACT_TOP)) and builds every image as it goes.  Another OZ_NET_OPT_ACT_TARGET_LOG2
        // (a test hook / the window experiments of tools/target_probe.py) is applied AFTERWARDS as an exact bump of every exponent, and pass 1
        // rebuilds the images from the bumped exponents without calibrating: the calibration passes themselves never run outside the fp16
        // range, whatever the target.
        constexpr int TOP = H2_ACT_TOP;
        for (int pass = 0; pass < 2; ++pass) {
            const bool calibrating = pass == 0;
            if (!calibrating) {
                if (act_target_log2 == TOP) break;
                for (int t = 0; t < 5; ++t) {
                    for (int& e : aexp[t]) e += act_target_log2 - TOP;
                    for (int& e : wexp[t]) e += act_target_log2 - TOP;       // layer t consumes tensor t: its columns' maxima shrank by the same power
                    if (int rc = up(&d_aexp[t], aexp[t])) return rc;
                }
            }
            // ---- tensor 0: conv1's output, exact supremum per channel over the 3^9 patterns
            if (calibrating) {
                OZ_HIP(hipMemset(d_colmax, 0, sizeof(unsigned) * wide));
                hipLaunchKernelGGL(k_lut_colmax, dim3((C + 255) / 256, (OZ_LUT_PATTERNS + 63) / 64), dim3(256), 0, 0, C, d_w1, d_scale[0], d_shift[0], d_colmax);
                OZ_HIP(hipGetLastError());
                if (int rc = read_colmax(C, mx, "conv1's output")) return rc;
                pick_exponents(mx, TOP, aexp[0]);
            }
            if (int rc = up(&d_aexp[0], aexp[0])) return rc;
            sc.resize(C); sh.resize(C);
            for (int c = 0; c < C; ++c) { sc[c] = ldexpf(bn_sc[0][c], aexp[0][c]); sh[c] = ldexpf(bn_sh[0][c], aexp[0][c]); }
            if (int rc = up(&d_scale1_h2, sc)) return rc;
            if (int rc = up(&d_shift1_h2, sh)) return rc;
            {
                // conv1 pattern table (k_lut_build): OZ_LUT_ROWS rows of C channels in the h2 layout, scaled like act1; an entry beyond the fp16
                // range (possible only with a raised OZ_NET_OPT_ACT_TARGET_LOG2) disables the tables for this network (the conv1 kernel then
                // raises the flag on real positions)
                const size_t row_q = (size_t)C / 4;                                  // uint4 per row
                if (!d_lut) { if (int rc = alloc(&d_lut, (size_t)OZ_LUT_ROWS * row_q)) return rc; }
                if (!d_lut_ids) { if (int rc = alloc(&d_lut_ids, (size_t)max_batch * (n + 2) * (n + 2))) return rc; }
                if (!d_lut_low) { if (int rc = alloc(&d_lut_low, (size_t)OZ_LUT_ROWS)) return rc; }
                OZ_HIP(hipMemset(d_lut + (size_t)OZ_LUT_PATTERNS * row_q, 0, row_q * sizeof(uint4)));
                OZ_HIP(hipMemset(d_lut_low, 0, OZ_LUT_ROWS));
                const long long threads = (long long)OZ_LUT_PATTERNS * (C / 8);
                hipLaunchKernelGGL(k_lut_build, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, C, d_w1, d_scale1_h2, d_shift1_h2, d_lut, d_flag);
                OZ_HIP(hipGetLastError());
                int over = 0;
                OZ_HIP(hipMemcpy(&over, d_flag, sizeof(int), hipMemcpyDeviceToHost));
                lut_ok = !over;
                OZ_HIP(hipMemset(d_flag, 0, sizeof(int)));
                if (lut_ok && low_thr > 0.f) hipLaunchKernelGGL(k_lut_rowlow, dim3(OZ_LUT_PATTERNS), dim3(64), 0, 0, C, d_lut, low_thr, d_lut_low);
                OZ_HIP(hipGetLastError());
            }
            for (int i = 0; i < 5; ++i) {
                const auto& src = w[gl[i]];
                const int K = Ks[i], N = Ns[i], Cmod = Wt[i], taps = i < 3 ? 9 : 1;
                OZ_REQUIRE(src.size() == (size_t)K * N, "weight %d has %zu values, expected %zu", gl[i], src.size(), (size_t)K * N);
                OZ_HIP(hipMemcpy(d_raw, src.data(), sizeof(float) * src.size(), hipMemcpyHostToDevice));
                if (calibrating) {
                    OZ_HIP(hipMemset(d_colmax, 0, sizeof(unsigned) * wide));
                    hipLaunchKernelGGL(k_w_colmax, dim3((N + 255) / 256, (K + 63) / 64), dim3(256), 0, 0, d_raw, K, N, d_aexp[i], Cmod, d_colmax);
                    OZ_HIP(hipGetLastError());
                    if (int rc = read_colmax(N, mx, "a weight kernel")) return rc;
                    pick_exponents(mx, w_target_log2, wexp[i]);
                }
                OZ_HIP(hipMemcpy(d_wexp, wexp[i].data(), sizeof(int) * (size_t)N, hipMemcpyHostToDevice));
                if (!d_wh[i]) { if (int rc = alloc(&d_wh[i], (size_t)N * K / 4)) return rc; }
                const long long threads = (long long)N * (K / 8);
                hipLaunchKernelGGL(k_w_to_h2, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, d_raw, K, N, taps, d_wexp, d_aexp[i], Cmod, d_wh[i]);
                if (i == 0) {     // conv2 once more as nine [C][C] matrices (taps = 1) for the T2 tables (build_t2)
                    if (!d_wtap) { if (int rc = alloc(&d_wtap, (size_t)9 * C * C / 4)) return rc; }
                    for (int t = 0; t < 9; ++t)
                        hipLaunchKernelGGL(k_w_to_h2, dim3((unsigned)(((long long)C * (C / 8) + 255) / 256)), dim3(256), 0, 0,
                                           d_raw + (size_t)t * C * C, C, C, 1, d_wexp, d_aexp[0], C, d_wtap + (size_t)t * C * (C / 4));
                }
                OZ_HIP(hipGetLastError());
                // the layer's scale / shift: 2^-wexp per column, times the output tensor's exponents once they are known (fc2 writes fp32 rows
                // for the heads: no output exponents)
                const bool out_known = !calibrating && i < 4;
                sc.resize(N); sh.resize(N);
                for (int c = 0; c < N; ++c) {
                    const int ao = out_known ? aexp[i + 1][c] : 0;
                    sc[c] = ldexpf(bn_sc[i + 1][c], ao - wexp[i][c]); sh[c] = ldexpf(bn_sh[i + 1][c], ao);
                }
                if (int rc = up(&d_scale_h2[i], sc)) return rc;
                if (int rc = up(&d_shift_h2[i], sh)) return rc;
                OZ_HIP(hipDeviceSynchronize());                       // d_raw and d_wexp are reused by the next layer
                if (i == 0) { t2_ok = false; if (lut_ok) { if (int rc = build_t2()) return rc; } }
                if (i == 4 || !calibrating) continue;
                OZ_HIP(hipMemset(d_colmax, 0, sizeof(unsigned) * wide));
                for (int c0 = 0; c0 < cal_total; c0 += max_batch) {
                    const int cnt = std::min(max_batch, cal_total - c0);
                    OZ_HIP(hipMemcpy(d_cal_count, &cnt, sizeof(int), hipMemcpyHostToDevice));
                    if (int rc = forward_h2(d_cal_own + c0, d_cal_opp + c0, d_cal_count, cnt, nullptr, nullptr, 0, i + 1)) return rc;
                    const long long rows = (long long)cnt * P_of[i];
                    hipLaunchKernelGGL(k_rows_colmax, dim3((N + 255) / 256, (unsigned)((rows + 63) / 64)), dim3(256), 0, 0, act_of[i], d_cal_count, P_of[i], N, d_colmax);
                    OZ_HIP(hipGetLastError());
                    OZ_HIP(hipDeviceSynchronize());                   // d_cal_count is rewritten for the next chunk
                }
                static const char* const names[4] = {"conv2's output", "conv3's output", "conv4's output", "fc1's output"};
                if (int rc = read_colmax(N, mx, names[i])) return rc;
                pick_exponents(mx, TOP, aexp[i + 1]);
                if (int rc = up(&d_aexp[i + 1], aexp[i + 1])) return rc;
                for (int c = 0; c < N; ++c) { sc[c] = ldexpf(bn_sc[i + 1][c], aexp[i + 1][c] - wexp[i][c]); sh[c] = ldexpf(bn_sh[i + 1][c], aexp[i + 1][c]); }
                if (int rc = up(&d_scale_h2[i], sc)) return rc;
                if (int rc = up(&d_shift_h2[i], sh)) return rc;
            }
        }
        OZ_HIP(hipMemset(d_flag, 0, sizeof(int)));
        sc_dpi = sc_dv = -1.0; sc_positions = 0; sc_guard = 0;
        if (self_check && act_target_log2 == TOP) { if (int rc = run_self_check()) return rc; }
        return OZ_OK;
    }

    // Commit-time self-check of precision f16x2: the calibration positions through the f16x2 kernels AND through the exact-fp32 kernels
    // (fp32 copies of the five GEMM kernels, conv1 kernel + conv2 GEMM: +61 MB and a few ms per commit at 8x8 / 512), and the largest
    // |d pi|, |d v| between the two must stay below H2_SELF_CHECK_LIMIT = 8e-6.  What it catches is not range (the scaling and the
    // guards handle that) but CONDITIONING: the split carries ~22 bits where fp32 carries 24, and a network that amplifies rounding -- a
    // BN variance far below epsilon behind a large constant, say -- turns those two bits into a miss of 1e-5 that no range guard sees.
    // Where the limit comes from (tools/f16x2_error_probe.py, round 4, final scaling defaults; E = error against the float64 oracle on 48 test
    // boards, D = what this check measures): healthy networks -- every parameter kind random, heads x 4 -- E16 <= 1.3e-6, E32 <= 2.9e-6 (the fp32
    // kernels have their own rounding), D <= 3.8e-6; the badly conditioned test network (conv2 kernel x 2^-12, conv3's BN variance x 2^-24):
    // 8x8 / 256 filters E16 6.5e-6, D 1.4e-5; 8x8 / 512 E16 2.5e-6, D 6.2e-6; 6x6 / 512 E16 1.06e-5 (a MISS), D 2.3e-5 -- D runs at 2-3 x E16 once
    // conditioning dominates, so 8e-6 refuses networks from E16 ~ 3-4e-6 on and leaves every healthy one alone.
    // TRAINED networks (round 5, tools/trained_net_self_check.py: 8x8 / 512 filters, three iterations of self-play + fit on the GPU, the trained weights
    // committed in measure-only mode): D = 1.7e-6, 8.9e-7, 4.8e-7 after iterations 1, 2, 3 (E16 <= 8.4e-7) -- further from the limit than the random-init
    // networks above (profiles/r5_trained_network_self_check.json).
    // A refused network fails here, at commit, with OZ_ERR_STATE ("use precision f32"; NNetWrapper.train falls back to precision f32 with a warning).
    // OZ_NET_OPT_SELF_CHECK: 0 off, 2 measure only (the commit succeeds; oz_net_self_check / OZ_NET_INFO_SELF_CHECK_GUARD say what was seen).
    int run_self_check() {
        const int gl[5] = {6, 12, 18, 24, 30};
        const int Ks[5] = {9 * C, 9 * C, 9 * C, F, 1024}, Ns[5] = {C, C, C, 1024, 512};
        for (int i = 0; i < 5; ++i) {
            const auto& src = w[gl[i]];
            OZ_HIP(hipMemcpy(d_raw, src.data(), sizeof(float) * src.size(), hipMemcpyHostToDevice));
            if (!d_wt[i]) { if (int rc = alloc(&d_wt[i], (size_t)Ks[i] * Ns[i])) return rc; }
            hipLaunchKernelGGL(k_w_transpose, dim3((Ks[i] + 31) / 32, (Ns[i] + 31) / 32), dim3(256), 0, 0, d_raw, Ks[i], Ns[i], d_wt[i]);
            OZ_HIP(hipGetLastError());
            OZ_HIP(hipDeviceSynchronize());
        }
        const int per = A + 1;
        if (!d_sc_out) { if (int rc = alloc(&d_sc_out, (size_t)2 * cal_total * per)) return rc; }
        for (int which = 0; which < 2; ++which)
            for (int c0 = 0; c0 < cal_total; c0 += max_batch) {
                const int cnt = std::min(max_batch, cal_total - c0);
                OZ_HIP(hipMemcpy(d_cal_count, &cnt, sizeof(int), hipMemcpyHostToDevice));
                float* pi = d_sc_out + (size_t)which * cal_total * per + (size_t)c0 * A;
                float* v = d_sc_out + (size_t)which * cal_total * per + (size_t)cal_total * A + c0;
                if (int rc = which == 0 ? forward_h2(d_cal_own + c0, d_cal_opp + c0, d_cal_count, cnt, pi, v, 0)
                                        : forward_f32(d_cal_own + c0, d_cal_opp + c0, d_cal_count, cnt, pi, v, 0)) return rc;
                OZ_HIP(hipDeviceSynchronize());
            }
        std::vector<float> out((size_t)2 * cal_total * per);
        OZ_HIP(hipMemcpy(out.data(), d_sc_out, sizeof(float) * out.size(), hipMemcpyDeviceToHost));
        int flag = 0;
        OZ_HIP(hipMemcpy(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost));
        // a guard fired on the calibration positions themselves: mode 1 fails the commit with the guard's own message, mode 2 (measure only)
        // records it (OZ_NET_INFO_SELF_CHECK_GUARD) and goes on; either way the flag does not stay sticky beyond the commit that reported it
        int guard_rc = OZ_OK;
        if (flag) {
            guard_rc = check();
            OZ_HIP(hipMemset(d_flag, 0, sizeof(int)));
        }
        sc_guard = flag;
        double dpi = 0.0, dv = 0.0;
        const float *a = out.data(), *b = out.data() + (size_t)cal_total * per;
        bool finite = true;
        for (size_t i = 0; i < (size_t)cal_total * per; ++i) {
            const double d = fabs((double)a[i] - (double)b[i]);
            if (!(d <= 1e30)) finite = false;
            if (i < (size_t)cal_total * A) dpi = std::max(dpi, d); else dv = std::max(dv, d);
        }
        sc_dpi = dpi; sc_dv = dv; sc_positions = cal_total;
        if (self_check == 1 && flag) return guard_rc;
        if (self_check == 1 && (!finite || dpi > H2_SELF_CHECK_LIMIT || dv > H2_SELF_CHECK_LIMIT)) {
            oz_set_error("oz_net_commit (precision f16x2): self-check failed -- on %d calibration positions the f16x2 kernels and the exact-fp32 kernels differ by "
                         "max |d pi| = %.3g, max |d v| = %.3g (limit %.1g): this network amplifies rounding beyond what the 2 x fp16 split (22 of fp32's 24 "
                         "bits) holds within 1e-5; use precision f32 for it", cal_total, dpi, dv, H2_SELF_CHECK_LIMIT);
            return OZ_ERR_STATE;
        }
        return OZ_OK;
    }

    const int* flag_device() override { return d_flag; }
    int check() override {
        if (!d_flag) return OZ_OK;
        int f = 0;
        OZ_HIP(hipMemcpy(&f, d_flag, sizeof(int), hipMemcpyDeviceToHost));
        if (f & H2_FLAG_OVER) {
            oz_set_error("an activation exceeded the fp16 range (65504) in precision mode f16x2 (more than 2^%d above its channel's calibration "
                         "maximum): results are invalid; use precision f32 for this network", 16 - act_target_log2);
            return OZ_ERR_STATE;
        }
        if (f & H2_FLAG_LOW) {
            oz_set_error("precision mode f16x2: a position's activations fell below the range the commit-time calibration covers (a pixel row whose "
                         "largest activation is non-zero and 2^%d or more below its channels' calibration maxima): results may miss the 1e-5 class; "
                         "use precision f32 for this network", act_target_log2 - low_guard_log2);
            return OZ_ERR_STATE;
        }
        return OZ_OK;
    }

    // calib = 0: the whole forward.  calib = 1 .. 4 (oz_net_commit's calibration passes): stop after conv2 / conv3 / conv4 / fc1 and write THAT
    // layer's BN output BEFORE the ReLU as fp32 rows (no h2 split, no guard) into its activation buffer, for k_rows_colmax.
    int forward_h2(const uint64_t* d_own, const uint64_t* d_opp, const int* d_count, int max_count, float* d_pi, float* d_v,
                   hipStream_t s, int calib = 0) {
        // main loop of the 3x3 convolutions: the ping-pong loops (4 phases per k-tile on 256-row tiles, 2 on 192 / 128), or (oz_net_set_option OZ_NET_OPT_SIMPLE_LOOP: the reference
        // form the race screen compares against, tools/pp_race_check.py) one barrier per k-tile; same accumulation order, bit-identical
        const bool pp = !simple_loop;
        // The input planes are discrete, so conv1 (and conv2 behind it) are functions of small neighbourhood patterns
        // (oz_net_set_tables): 2 (default) = conv1 + conv2 as a gather-sum over the per-tap tables T2 (k_conv2_lut) -- no GEMM for conv2;
        // 1 = conv1 as a table lookup inside conv2's operand gather (H2BigPPLut; bit-identical to the conv1 kernel); 0 = conv1 kernel + conv2 GEMM
        const bool want_lut = tables_mode < 0 || tables_mode >= 1;
        const bool want_t2 = tables_mode < 0 || tables_mode >= 2;
        const bool use_t2 = want_t2 && t2_ok;
        const bool use_lut = !use_t2 && pp && want_lut && lut_ok && max_batch > 32;
        // HIP events around the dominant launch (the conv2 GEMM, or conv3 when conv2 is the gather-sum), or around every kernel
        profiled_layer = use_t2 ? 3 : 2;
        if (profile && timer.backlog() > 4096) timer.drain();          // no host stall inside an enqueue loop: only pairs that have completed
        long long tidx = -1;
        auto mark = [&](int slot, bool begin) {
            if (!(profile == 2 || (profile == 1 && slot == profiled_layer - 1))) return;
            if (begin) tidx = timer.begin(slot, s);
            else { timer.end(tidx, s); tidx = -1; }
        };
        const bool guard = calib == 0;
        if (guard) { fwd_seq = (fwd_seq + 1) & 0x3FFFFFFu; if (!fwd_seq) fwd_seq = 1; }
        mark(0, true);
        if (use_t2 || use_lut) {
            const long long threads = (long long)max_count * (n + 2) * (n + 2);
            hipLaunchKernelGGL(k_lut_ids, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, d_own, d_opp, d_count, n, d_lut_ids,
                               (guard && low_thr > 0.f) ? d_lut_low : (const unsigned char*)nullptr, d_flag);
        } else {
            const long long threads = (long long)max_count * n * (C / 8);       // one thread per (board row, 8 channels)
            hipLaunchKernelGGL(k_conv1_h2, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, d_own, d_opp, d_count, n, C,
                               d_w1, d_scale1_h2, d_shift1_h2, (uint4*)act1, d_flag, low_of(0, guard));
        }
        mark(0, false);
        const bool small = max_batch <= 32;
        // small networks (the drop-in OthelloMCTS / agents path, one position per call): latency, not throughput --
        // 128 x 128 tiles with the k loop split 16 ways over otherwise idle CUs, fixed-order reduce (keyed on max_batch,
        // a per-network constant, so a position's result does not depend on the size of the call)
        // medium networks (arenas, evaluation batches, the loop's 100 episodes): a convolution whose grid would leave most
        // CUs idle splits its k loop (conv_ksplit: from max_batch, a per-network constant; 1 at the bench's 4096 games)
        // conv3 tile: 192 rows on 8x8 (36 output pixels per board: 6.0 grid rounds at 4096 boards instead of 4.5); on 6x6 (16 pixels per board) the
        // 256-row tile = 16 whole boards, 2.0 rounds instead of 2.7
        // ... and on either board the tile whose grid pays fewer tile-rows for the batch this call may hold: rounds of 256 CUs x tile height
        // (4096 boards of 8x8: 192 rows -> 6 rounds x 192; a caller that caps its batches at 3640 gets 256 rows -> 4.0 rounds x 256, -13 % per launch).
        // Both tiles add every output element's products in the same order: bit-identical results.
        // (round 3, measured and removed: a MIXED plan -- whole rounds of 256-row tiles, the rest of the rows on 192-row tiles in a second launch over
        //  a position range -- with the cap that suits it, 3584 leaves: conv4 = 2048 boards on 256-row + 1536 on 192-row tiles = 1.75 rounds instead
        //  of 2.0.  conv4 547 / 529 -> 508 us, not the 483 the round count promises (a round of 192-row tiles costs 0.84 of a 256-row round on conv4,
        //  235 against 280 us, not 0.75), and the layers that do not shrink with the cap eat most of it: 1.696 / 1.680 M -> 1.696 / 1.691 M
        //  expansions/s, within the run-to-run spread.)
        auto tile_cost = [&](int BM) {
            const long long blocks = (((long long)max_count * (n - 2) * (n - 2) + BM - 1) / BM) * (C / 256);
            return ((blocks + 255) / 256) * BM;
        };
        // ... and the 128-row tile when a call's rows fit ONE round on it but leave a third of the chip idle on the taller tiles (the arena's
        // <= 512-leaf batches: 430 leaves = 162 blocks of 192 rows, 242 of 128).  Its round costs CONV3_LOW_COST of the row-proportional figure
        // (24 instead of 36 MFMAs per phase of the 2-phase loop against the same LDS reads and barriers: 151 / 158 us against 172 / 177 us
        // for one round of 192-row tiles).
        int c3rows = n == 6 ? 256 : tile_cost(256) < tile_cost(192) ? 256 : 192;
        if (n != 6 && pp && (double)tile_cost(128) * CONV3_LOW_COST < (double)tile_cost(c3rows)) c3rows = 128;
        if (conv3_tile && pp && !small) c3rows = conv3_tile;
        const bool conv3_big = c3rows == 256, conv3_low = c3rows == 128;
        last_conv3_rows = small ? 128 : pp ? c3rows : 192;
        // (the k-split stays a constant of the network -- max_batch and the board decide it, not the tile this call picked -- so a position's
        //  result does not depend on the size of the call it sits in)
        const int k2 = conv_ksplit(n * n, 256), k3 = conv_ksplit((n - 2) * (n - 2), n == 6 ? 256 : 192),
                  k4 = conv_ksplit((n - 4) * (n - 4), conv4_low() ? 128 : 256);
        // h2 output (and the low-side guard of the tensor) unless this is the layer a calibration pass wants as fp32 rows
        const int h2o1 = calib == 1 ? 0 : 1, h2o2 = calib == 2 ? 0 : 1, h2o3 = calib == 3 ? 0 : 1, h2o4 = calib == 4 ? 0 : 1;
        mark(1, true);
        if (use_t2) {
            if (h2o1) launch_conv2_lut<true>(max_count, d_count, d_scale_h2[0], d_shift_h2[0], act2, s, low_of(1, guard));
            else launch_conv2_lut<false>(max_count, d_count, d_scale_h2[0], d_shift_h2[0], act2, s, H2Low(), -__builtin_inff());
        } else {
            next_low = low_of(1, guard);
            next_relu = h2o1;
            if (int rc = small     ? launch_small<H2Small, H2Small2>(act1, 1, act2, h2o1, d_count, max_count, n, n, 1, C, 9, C, s, 16)
                         : use_lut ? launch_gemm_h2<H2BigPPLut, 2>(d_lut, 1, act2, h2o1, d_count, max_count, n, n, 1, C, 9, C, s, k2, d_lut_ids)
                         : pp      ? launch_gemm_h2<H2BigPP, 2>(act1, 1, act2, h2o1, d_count, max_count, n, n, 1, C, 9, C, s, k2)
                                   : launch_gemm_h2<H2Big>(act1, 1, act2, h2o1, d_count, max_count, n, n, 1, C, 9, C, s, k2)) return rc;
        }
        mark(1, false);
        if (calib == 1) { OZ_HIP(hipGetLastError()); return OZ_OK; }
        mark(2, true);
        // (a 3-phase loop on the 192-row tile -- 24-MFMA clusters -- measured 0 .. +2 % in round 2: the layer is clock / power bound,
        //  not load-section bound; deleted in round 3)
        next_low = low_of(2, guard);
        next_relu = h2o2;
        if (int rc = small ? launch_small<H2Small, H2Small2>(act2, 2, act3, h2o2, d_count, max_count, n, n - 2, 0, C, 9, C, s, 16)
                     : pp && conv3_big ? launch_gemm_h2<H2BigPP, 3>(act2, 2, act3, h2o2, d_count, max_count, n, n - 2, 0, C, 9, C, s, k3)
                     : pp && conv3_low && low_loop_phases == 2 ? launch_gemm_h2<H2LowPP, 3>(act2, 2, act3, h2o2, d_count, max_count, n, n - 2, 0, C, 9, C, s, k3)
                     : pp && conv3_low ? launch_gemm_h2<H2LowPP1, 3>(act2, 2, act3, h2o2, d_count, max_count, n, n - 2, 0, C, 9, C, s, k3)
                     : pp  ? launch_gemm_h2<H2MidPP, 3>(act2, 2, act3, h2o2, d_count, max_count, n, n - 2, 0, C, 9, C, s, k3)
                           : launch_gemm_h2<H2Mid>(act2, 2, act3, h2o2, d_count, max_count, n, n - 2, 0, C, 9, C, s, k3)) return rc;
        mark(2, false);
        if (calib == 2) { OZ_HIP(hipGetLastError()); return OZ_OK; }
        mark(3, true);
        next_low = low_of(3, guard);
        next_relu = h2o3;
        if (int rc = small ? launch_small<H2Small, H2Small2>(act3, 3, act4, h2o3, d_count, max_count, n - 2, n - 4, 0, C, 9, C, s, 16)
                     : pp && conv4_low() && low_loop_phases == 2 ? launch_gemm_h2<H2LowPP, 4>(act3, 3, act4, h2o3, d_count, max_count, n - 2, n - 4, 0, C, 9, C, s, k4)
                     : pp && conv4_low() ? launch_gemm_h2<H2LowPP1, 4>(act3, 3, act4, h2o3, d_count, max_count, n - 2, n - 4, 0, C, 9, C, s, k4)
                     : pp  ? launch_gemm_h2<H2BigPP, 4>(act3, 3, act4, h2o3, d_count, max_count, n - 2, n - 4, 0, C, 9, C, s, k4)
                           : launch_gemm_h2<H2Big>(act3, 3, act4, h2o3, d_count, max_count, n - 2, n - 4, 0, C, 9, C, s, k4)) return rc;
        mark(3, false);
        if (calib == 3) { OZ_HIP(hipGetLastError()); return OZ_OK; }
        mark(4, true);
        // fc1: K = 8192 but only batch x 1024 outputs -> split-K (fixed-order reduce) to fill the chip
        // (large batches: on the 256 x 256 ping-pong tile, 16 x 4 tiles x 4 k-slices = one block per CU; bit-identical to
        //  the 128 x 128 tile because the k-slices and the order inside them are the same -- tools/pp_race_check.py)
        next_low = low_of(4, guard);
        next_relu = h2o4;
        // (medium networks, max_batch < 1024: 16 k-slices on the 128 x 256 tile of the 2-phase ping-pong loop -- fc1 at 512 rows is 16 output tiles,
        //  x 16 slices = one block of 16 k-tiles per CU: 34 us in the arena's 512-leaf batches; 8 slices on 128 x 128 tiles 41 us, 4 slices 58 us
        //  (round 5); the split is keyed on max_batch, a per-network constant)
        const int kfc1 = sizing() >= 1024 ? 4 : 16;
        if (int rc = small ? launch_small<H2Small, H2Small2>(act4, 4, f1, h2o4, d_count, max_count, 1, 1, 0, F, 1, 1024, s, 16)
                     : (pp && sizing() >= 1024 && max_count >= 1024) ? launch_gemm_h2<H2BigPP, 5>(act4, 4, f1, h2o4, d_count, max_count, 1, 1, 0, F, 1, 1024, s, 4)
                     : pp && low_loop_phases == 2 ? launch_gemm_h2<H2LowPP, 5>(act4, 4, f1, h2o4, d_count, max_count, 1, 1, 0, F, 1, 1024, s, kfc1)
                     : pp ? launch_gemm_h2<H2LowPP1, 5>(act4, 4, f1, h2o4, d_count, max_count, 1, 1, 0, F, 1, 1024, s, kfc1)
                          : launch_small<H2Small, H2Small2>(act4, 4, f1, h2o4, d_count, max_count, 1, 1, 0, F, 1, 1024, s, kfc1)) return rc;
        mark(4, false);
        if (calib == 4) { OZ_HIP(hipGetLastError()); return OZ_OK; }
        mark(5, true);
        // fc2: one position has 4 blocks of 32 k-tiles -> small and medium networks split k 8 ways (from max_batch); the heads kernel adds the slices
        // (large networks: one k-slice on the four-wave form of the thin tile, bit-identical to the two-wave one)
        if (int rc = sizing() > 512 ? launch_gemm_h2<H2Thin4w, 6>(f1, 5, f2, 0, d_count, max_count, 1, 1, 0, 1024, 1, 512, s, 1)
                                   : launch_small<H2Thin, H2Thin2>(f1, 5, f2, 0, d_count, max_count, 1, 1, 0, 1024, 1, 512, s, 8)) return rc;
        mark(5, false);
        mark(6, true);
        launch_heads(max_count, d_count, d_pi, d_v, s);
        mark(6, false);
        OZ_HIP(hipGetLastError());
        return OZ_OK;
    }

    // layer: 1 = conv2 (the all-GEMM form), 2 = conv3, 3 = conv4 (3x3, Cin = N = C), 4 = fc1, 5 = fc2 (taps 1); the d_scale / d_shift of precision f32 (no power-of-two bookkeeping)
    template <int TAG>
    int launch_gemm_b3(const uint4* in, int layer, void* out, int out_b3, const int* d_count, int max_count, int Hin, int Hout, int pad,
                       int Cin, int taps, int N, hipStream_t s, int ksplit = 1) {
        B3Geom g;
        g.Hin = Hin; g.Hout = Hout; g.pad = pad; g.Cin = Cin; g.taps = taps; g.N = N; g.K = taps * Cin; g.out_b3 = out_b3; g.relu = 1;
        g.ksplit = ksplit; g.slab = (long long)max_batch * Hout * Hout * N;
        OZ_REQUIRE(N % B3_BN == 0 && Cin % B3_BK == 0, "gemm_b3: N %% 256 and Cin %% 32 must be 0 (N=%d Cin=%d)", N, Cin);
        const long long Mmax = (long long)max_count * Hout * Hout;
        // the 256 x 256 tile (two thirds of the operand bytes per MFMA: the small tile is bound by its LDS-DMA stream out of the L2s) for unsplit 'valid'
        // layers whose grid fills the chip on it -- keyed on the CAPACITY of the network (max_batch), like every tile / split choice, so that a position's
        // bits do not depend on the call (they would not anyway: both tiles add the same products in the same order; OZ_NET_OPT_B3_TILE screens that)
        // Which tile: the one whose grid pays fewer tile-rows at the network's capacity (rounds of the 256 CUs x tile height), the 256-row tile priced at
        // 0.96 of its rows (measured at equal fill: 3.59 us per k-tile against 2 x 1.87).  4096 positions of 8x8: conv3 = 9 rounds of 128 rows against 5 (4.5
        // paid as 5) of 256 -> the small tile; conv4 = 4 rounds of 128 against 2.0 of 256 -> the big one (974 against 1048 us at 3916 leaves).  The tile does not
        // dissolve the kernel's real bound, which turned out to be the clock under load, not the L2 stream (docs/HISTORY.md, round 6).
        const long long rows_cap = (long long)max_batch * Hout * Hout;
        auto paid = [&](int bm) { return (double)(((((rows_cap + bm - 1) / bm) * (N / B3_BN)) + 255) / 256) * bm; };
        const long long big_blocks = ((rows_cap + B3B_BM - 1) / B3B_BM) * (N / B3B_BN);
        const bool big = b3_tile != 128 && pad == 0 && ksplit == 1 && (b3_tile == 256 || (big_blocks >= 192 && 0.96 * paid(B3B_BM) <= paid(B3_BM)));
        if (layer == 2) last_conv3_rows = big ? B3B_BM : B3_BM;
        const int BMt = big ? B3B_BM : B3_BM;
        const int num_mt = (int)((Mmax + BMt - 1) / BMt);
        const int per_mt = (N / B3_BN) * ksplit;
        const int grid = num_mt < 8 ? ((per_mt + 7) / 8) * 8 * num_mt : ((num_mt + 7) / 8) * 8 * per_mt;   // (the kernels' two block mappings)
        {
            static bool attr_done[64] = {};
            if (!attr_done[device & 63]) {
                OZ_HIP(hipFuncSetAttribute((const void*)k_gemm_b3<TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, B3_LDS));
                OZ_HIP(hipFuncSetAttribute((const void*)k_gemm_b3_big<TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, B3B_LDS));
                attr_done[device & 63] = true;
            }
        }
        void* dst = ksplit > 1 ? (void*)d_part_b3 : out;
        if (big)
            hipLaunchKernelGGL((k_gemm_b3_big<TAG>), dim3(grid), dim3(B3_NT), B3B_LDS, s, in, (const uint4*)d_wb[layer - 1], d_scale[layer], d_shift[layer], dst,
                               d_count, g, num_mt, d_zero);
        else
            hipLaunchKernelGGL((k_gemm_b3<TAG>), dim3(grid), dim3(B3_NT), B3_LDS, s, in, (const uint4*)d_wb[layer - 1], d_scale[layer], d_shift[layer], dst,
                               d_count, g, num_mt, d_zero);
        if (ksplit > 1 && layer == 5) {           // fc2: the heads kernel adds the slices in fixed order (launch_heads), BN + ReLU there
            fc2_defer.partial = d_part_b3; fc2_defer.slab = g.slab; fc2_defer.ksplit = ksplit; fc2_defer.scale = d_scale[layer]; fc2_defer.shift = d_shift[layer];
        } else if (ksplit > 1 && out_b3) {        // the consumer reads the b3 layout: fixed-order reduce + BN + ReLU + split
            const long long threads = (long long)max_count * Hout * Hout * (N / 8);
            hipLaunchKernelGGL(k_splitk_reduce_b3, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, (const float*)d_part_b3, g.slab, ksplit, N,
                               Hout * Hout, d_count, d_scale[layer], d_shift[layer], (uint4*)out);
        } else if (ksplit > 1) {                  // fp32 rows out: the fp32 path's fixed-order reduce (BN + ReLU there)
            const long long quads = ((long long)max_count * Hout * Hout * N + 3) / 4;
            hipLaunchKernelGGL(k_splitk_reduce_f32, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, (const float*)d_part_b3, g.slab, ksplit, N,
                               Hout * Hout, d_count, d_scale[layer], d_shift[layer], 1, (float*)out);
        }
        OZ_HIP(hipGetLastError());
        return OZ_OK;
    }

    // precision bf16x3, networks of >= B3_MIN_BATCH positions: gather (exact fp32 tables) -> b3 rows -> conv3 / conv4 / fc1 on k_gemm_b3 -> fc2 and heads in fp32
    int forward_b3(const uint64_t* d_own, const uint64_t* d_opp, const int* d_count, int max_count, float* d_pi, float* d_v, hipStream_t s) {
        const bool use_t2f = (tables_mode < 0 || tables_mode >= 2) && t2f_ok;      // else (oz_net_set_tables 0 / 1): conv1 kernel + conv2 as a b3 GEMM
        last_conv3_rows = B3_BM;                     // (set by conv3's launch below)
        profiled_layer = use_t2f ? 3 : 2;            // the dominant launch: conv3 on k_gemm_b3, or conv2 on it when the tables are off
        if (profile && timer.backlog() > 4096) timer.drain();
        long long tidx = -1;
        auto mark = [&](int slot, bool begin) {
            if (!(profile == 2 || (profile == 1 && slot == profiled_layer - 1))) return;
            if (begin) tidx = timer.begin(slot, s);
            else { timer.end(tidx, s); tidx = -1; }
        };
        const long long cells = (long long)max_count * (n + 2) * (n + 2);
        mark(0, true);
        if (use_t2f)
            hipLaunchKernelGGL(k_lut_ids, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, s, d_own, d_opp, d_count, n, d_lut_ids,
                               (const unsigned char*)nullptr, (int*)nullptr);
        else {
            const long long threads = (long long)max_count * n * n * (C / 4);
            hipLaunchKernelGGL(k_conv1, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, d_own, d_opp, d_count, n, C, d_w1, d_scale[0], d_shift[0], act1);
        }
        mark(0, false);
        mark(1, true);
        if (use_t2f) launch_conv2_lut<2>(max_count, d_count, d_scale[1], d_shift[1], b3a2, s);      // the gather writes the b3 layout itself
        else {      // every layer as a kernel / GEMM: conv1's rows split into the three planes, conv2 on k_gemm_b3 like the rest ('same' padding: the 128-row tile)
            if (!b3a1) { if (int rc = alloc(&b3a1, (size_t)max_batch * n * n * (C / 32 * 12))) return rc; }
            const long long threads = (long long)max_count * n * n * (C / 8);
            hipLaunchKernelGGL(k_f32_to_b3, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, act1, d_count, n * n, C, b3a1);
            if (int rc = launch_gemm_b3<2>(b3a1, 1, b3a2, 1, d_count, max_count, n, n, 1, C, 9, C, s, conv_b3_ksplit(n * n))) return rc;
        }
        mark(1, false);
        mark(2, true);
        if (int rc = launch_gemm_b3<3>(b3a2, 2, b3a3, 1, d_count, max_count, n, n - 2, 0, C, 9, C, s, conv_b3_ksplit((n - 2) * (n - 2)))) return rc;
        mark(2, false);
        mark(3, true);
        if (int rc = launch_gemm_b3<4>(b3a3, 3, b3a4, 1, d_count, max_count, n - 2, n - 4, 0, C, 9, C, s, conv_b3_ksplit((n - 4) * (n - 4)))) return rc;
        mark(3, false);
        mark(4, true);
        if (int rc = launch_gemm_b3<5>(b3a4, 4, b3f1, 1, d_count, max_count, 1, 1, 0, F, 1, 1024, s, fc1_b3_ksplit())) return rc;
        mark(4, false);
        mark(5, true);
        if (int rc = launch_gemm_b3<6>(b3f1, 5, f2, 0, d_count, max_count, 1, 1, 0, 1024, 1, 512, s, fc2_b3_ksplit())) return rc;
        mark(5, false);
        mark(6, true);
        launch_heads(max_count, d_count, d_pi, d_v, s);
        mark(6, false);
        OZ_HIP(hipGetLastError());
        return OZ_OK;
    }

    int forward_device(const uint64_t* d_own, const uint64_t* d_opp, const int* d_count, int max_count, float* d_pi,
                       float* d_v, hipStream_t s) override {
        if (!committed) { oz_set_error("network weights not committed (call oz_net_commit)"); return OZ_ERR_STATE; }
        if (max_count > max_batch) { oz_set_error("batch %d exceeds max_batch %d", max_count, max_batch); return OZ_ERR_ARG; }
        if (precision == 1) return forward_h2(d_own, d_opp, d_count, max_count, d_pi, d_v, s);
        if (use_b3()) return forward_b3(d_own, d_opp, d_count, max_count, d_pi, d_v, s);
        return forward_f32(d_own, d_opp, d_count, max_count, d_pi, d_v, s);
    }
    // the exact-fp32 forward (precision f32; also the reference of precision f16x2's commit-time self-check, which keeps fp32 copies of
    // the kernels for it and no fp32 tables: conv1 kernel + conv2 GEMM then)
    int forward_f32(const uint64_t* d_own, const uint64_t* d_opp, const int* d_count, int max_count, float* d_pi,
                    float* d_v, hipStream_t s) {
        const int P = n * n;
        // precision f32: conv1 + conv2 from the fp32 pattern tables (default), or conv1 kernel + conv2 GEMM (oz_net_set_tables 0 / 1)
        const bool use_t2f = (tables_mode < 0 || tables_mode >= 2) && t2f_ok;
        last_conv3_rows = 0;             // set by conv3's launch below (launch_gemm, layer 2): 64 weight stream / 128 GmStd / 256 GmBig
        profiled_layer = use_t2f ? 3 : 2;
        if (profile && timer.backlog() > 4096) timer.drain();          // no host stall inside an enqueue loop: only pairs that have completed
        long long tidx = -1;
        auto mark = [&](int slot, bool begin) {
            if (!(profile == 2 || (profile == 1 && slot == profiled_layer - 1))) return;
            if (begin) tidx = timer.begin(slot, s);
            else { timer.end(tidx, s); tidx = -1; }
        };
        if (use_t2f && max_batch <= 32 && C == 512) {
            // few positions: pattern ids computed inside the gather (INLINE_IDS) -- one launch less
            mark(1, true);
            const unsigned blocks = 8u * (unsigned)(((long long)max_count * P + 32 * OZ_C2L_PPT - 1) / (32 * OZ_C2L_PPT));
            if (n == 8) hipLaunchKernelGGL((k_conv2_lut_xcd<8, false, true>), dim3(blocks), dim3(256), 0, s, (const unsigned*)nullptr, d_count, d_t2, d_scale[1], d_shift[1], (void*)act2, (int*)nullptr, H2Low(), 0.f, d_own, d_opp);
            else hipLaunchKernelGGL((k_conv2_lut_xcd<6, false, true>), dim3(blocks), dim3(256), 0, s, (const unsigned*)nullptr, d_count, d_t2, d_scale[1], d_shift[1], (void*)act2, (int*)nullptr, H2Low(), 0.f, d_own, d_opp);
            mark(1, false);
        } else if (use_t2f) {
            const long long cells = (long long)max_count * (n + 2) * (n + 2);
            mark(0, true);
            hipLaunchKernelGGL(k_lut_ids, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, s, d_own, d_opp, d_count, n, d_lut_ids,
                               (const unsigned char*)nullptr, (int*)nullptr);
            mark(0, false);
            mark(1, true);
            launch_conv2_lut<false>(max_count, d_count, d_scale[1], d_shift[1], act2, s);
            mark(1, false);
        } else {
            const long long threads = (long long)max_count * P * (C / 4);
            mark(0, true);
            hipLaunchKernelGGL(k_conv1, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, d_own, d_opp, d_count, n, C,
                               d_w1, d_scale[0], d_shift[0], act1);
            mark(0, false);
            mark(1, true);
            if (int rc = launch_gemm(act1, d_wt[0], 1, act2, d_count, max_count, n, n, 1, C, 9, C, s)) return rc;
            mark(1, false);
        }
        mark(2, true);
        if (int rc = launch_gemm(act2, d_wt[1], 2, act3, d_count, max_count, n, n - 2, 0, C, 9, C, s)) return rc;
        mark(2, false);
        mark(3, true);
        if (int rc = launch_gemm(act3, d_wt[2], 3, act4, d_count, max_count, n - 2, n - 4, 0, C, 9, C, s)) return rc;
        mark(3, false);
        mark(4, true);
        if (int rc = launch_gemm(act4, d_wt[3], 4, f1, d_count, max_count, 1, 1, 0, F, 1, 1024, s)) return rc;
        mark(4, false);
        mark(5, true);
        if (int rc = launch_gemm(f1, d_wt[4], 5, f2, d_count, max_count, 1, 1, 0, 1024, 1, 512, s, &fc2_defer)) return rc;
        mark(5, false);
        mark(6, true);
        launch_heads(max_count, d_count, d_pi, d_v, s);
        mark(6, false);
        OZ_HIP(hipGetLastError());
        return OZ_OK;
    }
    int collect_profile() {
        if (timer.collect() != OZ_OK) { oz_set_error("HIP event timing failed"); return OZ_ERR_HIP; }
        return OZ_OK;
    }
};

int oz_net_forward_device(oz_net* net, const uint64_t* d_own, const uint64_t* d_opp, const int* d_count, int max_count,
                          float* d_pi, float* d_v, hipStream_t s) {
    return net->forward_device(d_own, d_opp, d_count, max_count, d_pi, d_v, s);
}

static int net_create(oz_net** out, int n, int channels, int max_batch, int cin);
OZ_API int oz_net_create(oz_net** out, int n, int channels, int max_batch) { return net_create(out, n, channels, max_batch, 2); }
// BaseNN (Net/BaseNN.py:41-56): the same trunk on ONE input plane holding +1 (mover) / -1 (opponent) / 0
OZ_API int oz_net_create_bnn(oz_net** out, int n, int channels, int max_batch) { return net_create(out, n, channels, max_batch, 1); }
static int net_create(oz_net** out, int n, int channels, int max_batch, int cin) {
    OZ_REQUIRE(out, "null out pointer");
    OZ_REQUIRE(n == 6 || n == 8, "OthelloNN needs board size 6 or 8 (two 'valid' 3x3 convolutions); got %d", n);
    OZ_REQUIRE(channels >= 128 && channels % 128 == 0, "channels must be a positive multiple of 128 (got %d)", channels);
    OZ_REQUIRE(max_batch > 0, "max_batch must be positive");
    OnnNet* net = new OnnNet();
    net->kind = 0; net->n = n; net->C = channels; net->max_batch = max_batch; net->device = oz_current_device();
    net->cin = cin;
    net->F = (n - 4) * (n - 4) * channels; net->A = n * n;
    auto sz = net->sizes();
    net->w.resize(sz.size());
    for (size_t i = 0; i < sz.size(); ++i) net->w[i].assign((size_t)sz[i], 0.f);
    // BN defaults: gamma 1, variance 1
    for (int l = 0; l < 6; ++l) {
        std::fill(net->w[6 * l + 2].begin(), net->w[6 * l + 2].end(), 1.f);
        std::fill(net->w[6 * l + 5].begin(), net->w[6 * l + 5].end(), 1.f);
    }
    *out = net;
    return OZ_OK;
}

OZ_API int oz_net_create_stub(oz_net** out, int n, uint64_t salt, uint64_t keep_mask, int max_batch) {
    OZ_REQUIRE(out, "null out pointer");
    OZ_REQUIRE(n == 4 || n == 6 || n == 8, "board size must be 4, 6 or 8");
    StubNet* net = new StubNet();
    net->kind = 1; net->n = n; net->C = 0; net->max_batch = max_batch; net->device = oz_current_device();
    net->salt = salt; net->keep = keep_mask;
    *out = net;
    return OZ_OK;
}

OZ_API int oz_net_destroy(oz_net* net) { delete net; return OZ_OK; }

// ---------------------------------------------------------------- persistent evaluation cache (EvalCacheDev, oz_internal.h)
static int eval_cache_clear(oz_net* net) {
    EvalCacheDev& c = net->ec;
    if (!c.buckets) return OZ_OK;
    const size_t entries = (size_t)c.buckets * OZ_EC_WAYS;
    OZ_HIP(hipMemset(c.keys, 0, entries * 16));
    OZ_HIP(hipMemset(c.stamp, 0, entries * sizeof(unsigned)));
    return OZ_OK;
}
OZ_API int oz_net_set_eval_cache(oz_net* net, int64_t entries) {
    OZ_REQUIRE(net, "null net");
    OZ_REQUIRE(entries >= 0 && entries <= (1ll << 30), "oz_net_set_eval_cache: entries %lld", (long long)entries);
    std::lock_guard<std::mutex> lk(net->mu);
    hipSetDevice(net->device);
    OZ_HIP(hipDeviceSynchronize());                          // no search may be using the old table
    net->free_eval_cache();
    if (entries == 0) return OZ_OK;
    unsigned buckets = 1024;
    while ((long long)buckets * OZ_EC_WAYS < entries) buckets <<= 1;
    const size_t e = (size_t)buckets * OZ_EC_WAYS;
    EvalCacheDev c;
    c.n2 = net->n * net->n;
    hipError_t err = hipMalloc((void**)&c.keys, e * 16);
    if (err == hipSuccess) err = hipMalloc((void**)&c.pi, e * c.n2 * sizeof(float));
    if (err == hipSuccess) err = hipMalloc((void**)&c.v, e * sizeof(float));
    if (err == hipSuccess) err = hipMalloc((void**)&c.stamp, e * sizeof(unsigned));
    if (err == hipSuccess) err = hipMalloc((void**)&c.counters, 4 * sizeof(unsigned long long));
    if (err != hipSuccess) {
        hipFree(c.keys); hipFree(c.pi); hipFree(c.v); hipFree(c.stamp); hipFree(c.counters);
        oz_set_error("oz_net_set_eval_cache: %s", hipGetErrorString(err));
        return OZ_ERR_HIP;
    }
    c.buckets = buckets;
    net->ec = c;
    OZ_HIP(hipMemset(c.counters, 0, 4 * sizeof(unsigned long long)));
    return eval_cache_clear(net);
}
OZ_API int oz_net_eval_cache_stats(oz_net* net, int64_t* entries, int64_t* lookups, int64_t* hits, int64_t* inserts) {
    OZ_REQUIRE(net, "null net");
    std::lock_guard<std::mutex> lk(net->mu);
    hipSetDevice(net->device);
    unsigned long long c[4] = {0, 0, 0, 0};
    if (net->ec.buckets) { OZ_HIP(hipDeviceSynchronize()); OZ_HIP(hipMemcpy(c, net->ec.counters, sizeof c, hipMemcpyDeviceToHost)); }
    if (entries) *entries = (int64_t)net->ec.buckets * OZ_EC_WAYS;
    if (lookups) *lookups = (int64_t)c[0];
    if (hits) *hits = (int64_t)c[1];
    if (inserts) *inserts = (int64_t)c[2];
    return OZ_OK;
}

static OnnNet* as_onn(oz_net* net) { return net && net->kind == 0 ? static_cast<OnnNet*>(net) : nullptr; }

OZ_API int oz_net_num_weights(const oz_net* net) { return net && net->kind == 0 ? 40 : 0; }
OZ_API int oz_net_weight_size(const oz_net* net, int index, int64_t* nelem) {
    const OnnNet* o = as_onn(const_cast<oz_net*>(net));
    OZ_REQUIRE(o && nelem, "not an OthelloNN network");
    OZ_REQUIRE(index >= 0 && index < 40, "weight index out of range");
    *nelem = (int64_t)o->w[index].size();
    return OZ_OK;
}
OZ_API int oz_net_set_weight(oz_net* net, int index, const float* data, int64_t nelem) {
    OnnNet* o = as_onn(net);
    OZ_REQUIRE(o && data, "not an OthelloNN network");
    OZ_REQUIRE(index >= 0 && index < 40, "weight index out of range");
    OZ_REQUIRE(nelem == (int64_t)o->w[index].size(), "weight %d: expected %lld elements, got %lld", index, (long long)o->w[index].size(), (long long)nelem);
    std::lock_guard<std::mutex> lk(o->mu);
    memcpy(o->w[index].data(), data, sizeof(float) * (size_t)nelem);
    o->committed = false;
    return OZ_OK;
}
// Keras' defaults for a fresh OthelloNN (Net/OthelloNN.py:42-56): glorot_uniform kernels -- U(-l, l), l = sqrt(6 / (fan_in + fan_out)), fan = receptive
// field x channels for the 3x3 layers -- zero biases, BatchNormalization gamma 1, beta 0, moving mean 0, moving variance 1.  The stream is
// splitmix64 keyed by (seed, array index, element): the same weights on every rank and every host that asks for the same seed (they are NOT the
// numbers NumPy would draw for that seed; read them back with oz_net_get_weight).  oz_net_commit afterwards, as after oz_net_set_weight.
OZ_API int oz_net_init_random(oz_net* net, uint64_t seed) {
    OnnNet* o = as_onn(net);
    OZ_REQUIRE(o, "not an OthelloNN network");
    std::lock_guard<std::mutex> lk(o->mu);
    const int C_ = o->C, n2 = o->A;
    for (int i = 0; i < 40; ++i) {
        std::vector<float>& a = o->w[i];
        const int layer = i < 36 ? i / 6 : 6 + (i - 36) / 2, part = i < 36 ? i % 6 : (i - 36) % 2;      // part 0 = kernel, 1 = bias, 2..5 = gamma, beta, mean, variance
        if (part == 0) {
            double fan_in, fan_out;
            if (layer < 4) { fan_in = 9.0 * (layer == 0 ? o->cin : C_); fan_out = 9.0 * C_; }
            else if (layer == 4) { fan_in = o->F; fan_out = 1024; }
            else if (layer == 5) { fan_in = 1024; fan_out = 512; }
            else { fan_in = 512; fan_out = layer == 6 ? n2 : 1; }
            const double lim = sqrt(6.0 / (fan_in + fan_out));
            for (size_t k = 0; k < a.size(); ++k) {
                const uint64_t r = oz_sm64(oz_sm64(seed * 0x9E3779B97F4A7C15ull + (uint64_t)i) + k);
                a[k] = (float)((2.0 * ((double)(r >> 11) * (1.0 / 9007199254740992.0)) - 1.0) * lim);
            }
        } else {
            const float fill = (part == 2 || part == 5) ? 1.0f : 0.0f;
            for (auto& x : a) x = fill;
        }
    }
    o->committed = false;
    return OZ_OK;
}
OZ_API int oz_net_get_weight(const oz_net* net, int index, float* data, int64_t nelem) {
    const OnnNet* o = as_onn(const_cast<oz_net*>(net));
    OZ_REQUIRE(o && data, "not an OthelloNN network");
    OZ_REQUIRE(index >= 0 && index < 40, "weight index out of range");
    OZ_REQUIRE(nelem == (int64_t)o->w[index].size(), "weight %d: size mismatch", index);
    memcpy(data, o->w[index].data(), sizeof(float) * (size_t)nelem);
    return OZ_OK;
}

static int upload(OnnNet* o, float** dst, const std::vector<float>& h) {
    if (!*dst) { if (int rc = o->alloc(dst, h.size())) return rc; }
    OZ_HIP(hipMemcpy(*dst, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice));
    return OZ_OK;
}

OZ_API int oz_net_commit(oz_net* net) {
    OnnNet* o = as_onn(net);
    OZ_REQUIRE(o, "not an OthelloNN network");
    std::lock_guard<std::mutex> lk(o->mu);
    hipSetDevice(o->device);
    const int C = o->C, n = o->n;
    // BN fold (epsilon 1e-3): y = (x + bias - mean) * gamma / sqrt(var + eps) + beta
    const int widths[6] = {C, C, C, C, 1024, 512};
    for (int l = 0; l < 6; ++l) {
        std::vector<float> sc(widths[l]), sh(widths[l]);
        const auto &bias = o->w[6 * l + 1], &g = o->w[6 * l + 2], &be = o->w[6 * l + 3], &mu = o->w[6 * l + 4], &var = o->w[6 * l + 5];
        for (int c = 0; c < widths[l]; ++c) {
            const double s = (double)g[c] / sqrt((double)var[c] + 1e-3);
            sc[c] = (float)s;
            sh[c] = (float)(((double)bias[c] - (double)mu[c]) * s + (double)be[c]);
        }
        if (int rc = upload(o, &o->d_scale[l], sc)) return rc;
        if (int rc = upload(o, &o->d_shift[l], sh)) return rc;
        o->bn_sc[l] = sc; o->bn_sh[l] = sh;
    }
    if (o->cin == 2) {
        if (int rc = upload(o, &o->d_w1, o->w[0])) return rc;                   // [9][2][C] as stored
    } else {
        // one plane x in {+1, 0, -1} = own - opp:  x*w == own*w + opp*(-w) exactly, so the two-plane kernels serve BaseNN
        std::vector<float> w2((size_t)9 * 2 * C);
        for (int t = 0; t < 9; ++t)
            for (int c = 0; c < C; ++c) { w2[(size_t)(t * 2) * C + c] = o->w[0][(size_t)t * C + c]; w2[(size_t)(t * 2 + 1) * C + c] = -o->w[0][(size_t)t * C + c]; }
        if (int rc = upload(o, &o->d_w1, w2)) return rc;
    }
    // conv2..4 kernels (3,3,Cin,Cout) and dense kernels (in,out) are uploaded as stored ([K][N], k = tap*Cin + ci) and
    // re-laid out on the device: [N][K] fp32 for precision f32; the h2 layout [N][K/8][h1 x8 | h2 x8] in the GEMM's k order,
    // pre-scaled by an exact power of two, for precision f16x2 (k_w_transpose / k_w_to_h2; the host loops took 0.4 s per commit)
    const int gl[5] = {6, 12, 18, 24, 30};
    const int Ks[5] = {9 * C, 9 * C, 9 * C, o->F, 1024}, Ns[5] = {C, C, C, 1024, 512};
    {
        size_t raw_max = 0;
        for (int i = 0; i < 5; ++i) raw_max = std::max(raw_max, (size_t)Ks[i] * Ns[i]);
        if (!o->d_raw) { if (int rc = o->alloc(&o->d_raw, raw_max)) return rc; }
    }
    if (o->precision != 1 && !o->d_part32 && o->part32_mult() > 0) {
        if (int rc = o->alloc(&o->d_part32, o->part32_floats())) return rc;
    }
    if (!o->act1) {
        const size_t B = (size_t)o->max_batch;
        if (int rc = o->alloc(&o->act1, B * n * n * C)) return rc;
        if (int rc = o->alloc(&o->act2, B * n * n * C)) return rc;
        if (int rc = o->alloc(&o->act3, B * (n - 2) * (n - 2) * C)) return rc;
        if (int rc = o->alloc(&o->act4, B * (size_t)o->F)) return rc;
        if (int rc = o->alloc(&o->f1, B * 1024)) return rc;
        if (int rc = o->alloc(&o->f2, B * 512)) return rc;
    }
    // the heads first: precision f16x2's commit ends with whole forwards (its self-check)
    if (int rc = upload(o, &o->d_wpi, o->w[36])) return rc;
    if (int rc = upload(o, &o->d_bpi, o->w[37])) return rc;
    if (int rc = upload(o, &o->d_wv, o->w[38])) return rc;
    if (int rc = upload(o, &o->d_bv, o->w[39])) return rc;
    o->t2f_ok = false;
    o->t2_ok = false;
    if (o->precision != 1) {
        for (int i = 0; i < 5; ++i) {
            const auto& src = o->w[gl[i]];
            const int K = Ks[i], N = Ns[i];
            OZ_REQUIRE(src.size() == (size_t)K * N, "weight %d has %zu values, expected %zu", gl[i], src.size(), (size_t)K * N);
            OZ_HIP(hipMemcpy(o->d_raw, src.data(), sizeof(float) * src.size(), hipMemcpyHostToDevice));
            if (!o->d_wt[i]) { if (int rc = o->alloc(&o->d_wt[i], (size_t)K * N)) return rc; }
            hipLaunchKernelGGL(k_w_transpose, dim3((K + 31) / 32, (N + 31) / 32), dim3(256), 0, 0, o->d_raw, K, N, o->d_wt[i]);
            if (i == 0) {     // conv2 once more as nine [C(out)][C(in)] matrices for the fp32 T2 tables (build_t2_f32)
                if (!o->d_wtap32) { if (int rc = o->alloc(&o->d_wtap32, (size_t)9 * C * C)) return rc; }
                for (int t = 0; t < 9; ++t)
                    hipLaunchKernelGGL(k_w_transpose, dim3((C + 31) / 32, (C + 31) / 32), dim3(256), 0, 0, o->d_raw + (size_t)t * C * C, C, C,
                                       o->d_wtap32 + (size_t)t * C * C);
            }
            if (o->use_b3()) {                         // conv2 .. fc2 once more in the b3 layout (three bf16 planes, the GEMM's tap-inner k order)
                if (!o->d_wb[i]) { if (int rc = o->alloc(&o->d_wb[i], (size_t)N * (K / 32) * 12)) return rc; }
                const long long threads = (long long)N * (K / 8);
                hipLaunchKernelGGL(k_w_to_b3, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, o->d_raw, K, N, i < 3 ? 9 : 1, o->d_wb[i]);
            }
            OZ_HIP(hipGetLastError());
            OZ_HIP(hipDeviceSynchronize());                       // d_raw is reused by the next layer
        }
        if (o->use_b3() && !o->b3a2) {
            const size_t B = (size_t)o->max_batch, rq = (size_t)C / 32 * 12;        // uint4 per pixel row
            if (int rc = o->alloc(&o->b3a2, B * n * n * rq)) return rc;
            if (int rc = o->alloc(&o->b3a3, B * (n - 2) * (n - 2) * rq)) return rc;
            if (int rc = o->alloc(&o->b3a4, B * (n - 4) * (n - 4) * rq)) return rc;
            if (int rc = o->alloc(&o->b3f1, B * (1024 / 32 * 12))) return rc;
            size_t part = (size_t)std::max(o->fc1_b3_ksplit() * 1024, o->fc2_b3_ksplit() * 512) * B;       // the largest set of k-slices any layer writes
            const int px[3] = {n * n, (n - 2) * (n - 2), (n - 4) * (n - 4)};
            for (int q = 0; q < 3; ++q)
                if (o->conv_b3_ksplit(px[q]) > 1) part = std::max(part, (size_t)o->conv_b3_ksplit(px[q]) * B * px[q] * C);
            if (int rc = o->alloc(&o->d_part_b3, part)) return rc;
            if (!o->d_zero) { if (int rc = o->alloc(&o->d_zero, 16)) return rc; }
            OZ_HIP(hipMemset(o->d_zero, 0, 256));
        }
        if (int rc = o->build_t2_f32()) return rc;
    } else {
        if (int rc = o->commit_h2()) return rc;
    }
    OZ_HIP(hipDeviceSynchronize());
    if (int rc = eval_cache_clear(o)) return rc;             // new weights: every cached (pi, v) is stale
    o->committed = true;
    return OZ_OK;
}

OZ_API int oz_net_predict(oz_net* net, const uint64_t* own, const uint64_t* opp, int count, float* pi, float* v) {
    OZ_REQUIRE(net && own && opp && pi && v, "null argument");
    OZ_REQUIRE(count >= 0 && count <= net->max_batch, "count %d outside [0, max_batch=%d]", count, net->max_batch);
    if (count == 0) return OZ_OK;
    std::lock_guard<std::mutex> lk(net->mu);
    hipSetDevice(net->device);
    const int n2 = net->n * net->n;
    const size_t B = (size_t)net->max_batch;
    if (count <= OZ_PREDICT_DIRECT) {
        // few positions (the drop-in predict: one): no staging copies -- see oz_net::hp_in
        if (!net->hp_in) {
            hipError_t e;
            int consts[OZ_PREDICT_DIRECT + 1];
            for (int i = 0; i <= OZ_PREDICT_DIRECT; ++i) consts[i] = i;
            if ((e = hipHostMalloc((void**)&net->hp_in, 8 * 2 * OZ_PREDICT_DIRECT, hipHostMallocDefault)) != hipSuccess ||
                (e = hipHostMalloc((void**)&net->hp_out, 4 * OZ_PREDICT_DIRECT * 65, hipHostMallocDefault)) != hipSuccess ||
                (e = hipMalloc((void**)&net->d_counts, sizeof consts)) != hipSuccess ||
                (e = hipMemcpy(net->d_counts, consts, sizeof consts, hipMemcpyHostToDevice)) != hipSuccess) {
                if (net->hp_in) hipHostFree(net->hp_in);
                if (net->hp_out) hipHostFree(net->hp_out);
                if (net->d_counts) hipFree(net->d_counts);
                net->hp_in = nullptr; net->hp_out = nullptr; net->d_counts = nullptr;
                oz_set_error("pinned staging allocation failed: %s", hipGetErrorString(e));
                return OZ_ERR_HIP;
            }
        }
        memcpy(net->hp_in, own, 8ull * count);
        memcpy(net->hp_in + OZ_PREDICT_DIRECT, opp, 8ull * count);
        float* h_pi = net->hp_out;
        float* h_v = net->hp_out + (size_t)OZ_PREDICT_DIRECT * 64;
        if (int rc = net->forward_device(net->hp_in, net->hp_in + OZ_PREDICT_DIRECT, net->d_counts + count, count, h_pi, h_v, 0)) return rc;
        int flag = 0;
        const int* fd = net->flag_device();
        if (fd) OZ_HIP(hipMemcpyAsync(&flag, fd, sizeof(int), hipMemcpyDeviceToHost, 0));
        hipError_t e = hipStreamSynchronize(0);
        if (e != hipSuccess) { oz_set_error("forward failed: %s", hipGetErrorString(e)); return OZ_ERR_HIP; }
        memcpy(pi, h_pi, 4ull * count * n2);
        memcpy(v, h_v, 4ull * count);
        return flag ? net->check() : OZ_OK;
    }
    if (!net->p_in) {           // staging buffers live with the network
        hipError_t e;
        if ((e = hipMalloc((void**)&net->p_in, 8 * (2 * B + 1))) != hipSuccess || (e = hipMalloc((void**)&net->p_out, 4 * B * (n2 + 1))) != hipSuccess) {
            hipFree(net->p_in); hipFree(net->p_out);
            net->p_in = nullptr; net->p_out = nullptr;
            oz_set_error("hipMalloc failed: %s", hipGetErrorString(e));
            return OZ_ERR_HIP;
        }
        net->h_in.resize(2 * B + 1);
        net->h_out.resize(B * (n2 + 1) + 1);
    }
    // device image: own[0..count) | opp[count..2 count) | count
    memcpy(net->h_in.data(), own, 8ull * count);
    memcpy(net->h_in.data() + count, opp, 8ull * count);
    net->h_in[2 * count] = (uint64_t)(uint32_t)count;
    OZ_HIP(hipMemcpyAsync(net->p_in, net->h_in.data(), 8ull * (2 * count + 1), hipMemcpyHostToDevice, 0));
    float* d_pi = net->p_out;
    float* d_v = net->p_out + (size_t)count * n2;
    if (int rc = net->forward_device(net->p_in, net->p_in + count, reinterpret_cast<const int*>(net->p_in + 2 * count), count, d_pi, d_v, 0)) return rc;
    OZ_HIP(hipMemcpyAsync(net->h_out.data(), net->p_out, 4ull * count * (n2 + 1), hipMemcpyDeviceToHost, 0));
    int flag = 0;
    const int* fd = net->flag_device();
    if (fd) OZ_HIP(hipMemcpyAsync(&flag, fd, sizeof(int), hipMemcpyDeviceToHost, 0));
    hipError_t e = hipStreamSynchronize(0);
    if (e != hipSuccess) { oz_set_error("forward failed: %s", hipGetErrorString(e)); return OZ_ERR_HIP; }
    memcpy(pi, net->h_out.data(), 4ull * count * n2);
    memcpy(v, net->h_out.data() + (size_t)count * n2, 4ull * count);
    return flag ? net->check() : OZ_OK;
}

// The same call on boards in the reference's own layout: `boards` = count x (n, n, 2) bytes, NHWC, channel 0 = the mover's discs, any
// non-zero byte = a disc -- what Net/NNet.py:80-84 feeds Keras (np.bool (n, n, 2) boards).  Packed to bitboards here, on the host.
OZ_API int oz_net_predict_boards(oz_net* net, const uint8_t* boards, int count, float* pi, float* v) {
    OZ_REQUIRE(net && boards && pi && v, "null argument");
    OZ_REQUIRE(count >= 0 && count <= net->max_batch, "count %d outside [0, max_batch=%d]", count, net->max_batch);
    const int n = net->n;
    std::vector<uint64_t> own((size_t)count), opp((size_t)count);
    for (int b = 0; b < count; ++b) {
        const uint8_t* p = boards + (size_t)b * n * n * 2;
        uint64_t o = 0, q = 0;
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) {
                const uint8_t* cell = p + (size_t)(r * n + c) * 2;
                OZ_REQUIRE(!(cell[0] && cell[1]), "board %d: square (%d, %d) holds a disc in both channels", b, r, c);
                if (cell[0]) o |= 1ull << (r * 8 + c);
                if (cell[1]) q |= 1ull << (r * 8 + c);
            }
        own[b] = o; opp[b] = q;
    }
    return oz_net_predict(net, own.data(), opp.data(), count, pi, v);
}

__global__ void k_fill_boards(uint64_t* own, uint64_t* opp, int count, uint64_t valid) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint64_t a = oz_sm64(0x1234 + i), b = oz_sm64(0x9876 + 7ull * i), e = oz_sm64(0x5555 + 3ull * i);
    own[i] = a & ~b & e & valid;
    opp[i] = b & ~a & e & valid;
}

OZ_API int oz_net_time_forward(oz_net* net, int count, int iters, float* ms_avg) {
    OZ_REQUIRE(net && ms_avg, "null argument");
    OZ_REQUIRE(count > 0 && count <= net->max_batch && iters > 0, "bad count / iters");
    std::lock_guard<std::mutex> lk(net->mu);
    hipSetDevice(net->device);
    const int n2 = net->n * net->n;
    uint64_t *d_own, *d_opp; int* d_count; float *d_pi, *d_v;
    OZ_HIP(hipMalloc((void**)&d_own, 8ull * count)); OZ_HIP(hipMalloc((void**)&d_opp, 8ull * count));
    OZ_HIP(hipMalloc((void**)&d_count, 4)); OZ_HIP(hipMalloc((void**)&d_pi, 4ull * count * n2)); OZ_HIP(hipMalloc((void**)&d_v, 4ull * count));
    hipLaunchKernelGGL(k_fill_boards, dim3((count + 255) / 256), dim3(256), 0, 0, d_own, d_opp, count, oz_valid_mask(net->n));
    hipMemcpy(d_count, &count, 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int rc = net->forward_device(d_own, d_opp, d_count, count, d_pi, d_v, 0);   // warm-up
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters && !rc; ++i) rc = net->forward_device(d_own, d_opp, d_count, count, d_pi, d_v, 0);
    hipEventRecord(e1, 0);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    *ms_avg = ms / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(d_own); hipFree(d_opp); hipFree(d_count); hipFree(d_pi); hipFree(d_v);
    if (rc) return rc;
    OZ_HIP(e);
    return OZ_OK;
}

// the three-plane split of precision bf16x3 on the device, for the test that holds "every normal fp32 value exactly" to its word: planes[3 i ..] =
// (b1, b2, b3) of x[i] widened to fp32, sum[i] = (b1 + b2) + b3 evaluated in fp32 (each partial sum is exactly representable when the split is exact)
__global__ __launch_bounds__(256) void k_selftest_b3_split(const float* __restrict__ x, long long count, float* __restrict__ planes, float* __restrict__ sum) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    __bf16 b1, b2, b3;
    b3_split(x[i], b1, b2, b3);
    const float f1 = (float)b1, f2 = (float)b2, f3 = (float)b3;
    if (planes) { planes[3 * i] = f1; planes[3 * i + 1] = f2; planes[3 * i + 2] = f3; }
    sum[i] = (f1 + f2) + f3;
}
OZ_API int oz_selftest_b3_split(const float* x, int64_t count, float* planes, float* sum) {
    OZ_REQUIRE(x && sum && count > 0 && count <= (1ll << 28), "oz_selftest_b3_split: null argument or count outside (0, 2^28]");
    float *d_x = nullptr, *d_p = nullptr, *d_s = nullptr;
    hipError_t e = hipMalloc((void**)&d_x, sizeof(float) * (size_t)count);
    if (e == hipSuccess) e = hipMalloc((void**)&d_s, sizeof(float) * (size_t)count);
    if (e == hipSuccess && planes) e = hipMalloc((void**)&d_p, sizeof(float) * 3 * (size_t)count);
    if (e == hipSuccess) e = hipMemcpy(d_x, x, sizeof(float) * (size_t)count, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_selftest_b3_split, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, 0, d_x, (long long)count, d_p, d_s);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(sum, d_s, sizeof(float) * (size_t)count, hipMemcpyDeviceToHost);
    if (e == hipSuccess && planes) e = hipMemcpy(planes, d_p, sizeof(float) * 3 * (size_t)count, hipMemcpyDeviceToHost);
    if (d_x) hipFree(d_x);
    if (d_s) hipFree(d_s);
    if (d_p) hipFree(d_p);                         // (every buffer is released on the error paths too)
    OZ_HIP(e);
    return OZ_OK;
}

// precision: 0 = exact fp32 matrix cores (k_gemm_f32), 1 = f32 via 2 x fp16 split on the 16-bit matrix cores (oz_net_h2.h),
// 2 = f32 via 3 x bf16 split (oz_net_b3.h: every fp32 value exactly, six bf16 MFMA products per fp32 product; networks of fewer than
// B3_MIN_BATCH positions run the exact-fp32 kernels).
// Takes effect at the next oz_net_commit.
OZ_API int oz_net_set_precision(oz_net* net, int mode) {
    OnnNet* o = as_onn(net);
    OZ_REQUIRE(o, "not an OthelloNN network");
    OZ_REQUIRE(mode == 0 || mode == 1 || mode == 2, "precision must be 0 (f32), 1 (f16x2) or 2 (bf16x3)");
    OZ_REQUIRE(mode == 0 || o->C % 256 == 0, "precision f16x2 / bf16x3 needs channels %% 256 == 0 (got %d)", o->C);
    // the low-side guard counts a row's low 64-channel slices in 6 bits (h2_low_report): 64 slices or more would switch it off silently
    OZ_REQUIRE(mode != 1 || o->C <= 2048, "precision f16x2 supports at most 2048 channels (got %d): use precision f32", o->C);
    std::lock_guard<std::mutex> lk(o->mu);
    if (o->precision != mode) { o->precision = mode; o->committed = false; }
    return OZ_OK;
}
OZ_API int oz_net_get_precision(const oz_net* net) {
    const OnnNet* o = as_onn(const_cast<oz_net*>(net));
    return o ? o->precision : 0;
}
// sticky range check of the f16x2 mode (OZ_ERR_STATE if an activation left the fp16 range since the last commit)
OZ_API int oz_net_check(oz_net* net) {
    OZ_REQUIRE(net, "null net");
    std::lock_guard<std::mutex> lk(net->mu);
    hipSetDevice(net->device);
    return net->check();
}

// profiling of the dominant launch (conv2 implicit GEMM): HIP events on the launching stream
OZ_API int oz_net_profile(oz_net* net, int enable) {
    OnnNet* o = as_onn(net);
    OZ_REQUIRE(o, "not an OthelloNN network");
    OZ_REQUIRE(enable >= 0 && enable <= 2, "profile mode must be 0 (off), 1 (dominant launch) or 2 (every kernel)");
    std::lock_guard<std::mutex> lk(o->mu);
    o->profile = enable;
    return OZ_OK;
}

OZ_API int oz_net_set_tables(oz_net* net, int mode) {
    OnnNet* o = as_onn(net);
    OZ_REQUIRE(o, "not an OthelloNN network");
    OZ_REQUIRE(mode >= -1 && mode <= 2, "tables mode must be -1 (default), 0 (none), 1 (conv1) or 2 (conv1 + conv2)");
    std::lock_guard<std::mutex> lk(o->mu);
    if (o->tables_mode != mode) {
        // the table form adds conv2's products in another order than the GEMM forms: entries cached under the other mode are not what
        // the next forward would compute, and "a hit changes no bit" must hold
        hipSetDevice(o->device);
        OZ_HIP(hipDeviceSynchronize());
        if (int rc = eval_cache_clear(o)) return rc;
    }
    o->tables_mode = mode;
    return OZ_OK;
}

OZ_API int oz_net_set_option(oz_net* net, int option, int value) {
    OnnNet* o = as_onn(net);
    OZ_REQUIRE(o, "not an OthelloNN network");
    OZ_REQUIRE(option == OZ_NET_OPT_SIMPLE_LOOP || option == OZ_NET_OPT_ACT_TARGET_LOG2 || option == OZ_NET_OPT_LOW_GUARD_LOG2 ||
               option == OZ_NET_OPT_SELF_CHECK || option == OZ_NET_OPT_W_TARGET_LOG2 || option == OZ_NET_OPT_F32_STD_TILE ||
               option == OZ_NET_OPT_LATENCY_SPLITS || option == OZ_NET_OPT_CONV3_TILE || option == OZ_NET_OPT_LOW_LOOP_PHASES || option == OZ_NET_OPT_B3_TILE, "unknown network option %d", option);
    std::lock_guard<std::mutex> lk(o->mu);
    if (option == OZ_NET_OPT_SIMPLE_LOOP) o->simple_loop = value != 0;
    else if (option == OZ_NET_OPT_B3_TILE) {
        OZ_REQUIRE(value == 0 || value == 128 || value == 256, "OZ_NET_OPT_B3_TILE must be 0, 128 or 256 (got %d)", value);
        o->b3_tile = value;
    }
    else if (option == OZ_NET_OPT_LOW_LOOP_PHASES) {
        OZ_REQUIRE(value == 1 || value == 2, "OZ_NET_OPT_LOW_LOOP_PHASES must be 1 or 2 (got %d)", value);
        o->low_loop_phases = value;
    }
    else if (option == OZ_NET_OPT_F32_STD_TILE) o->f32_std_tile = value != 0;
    else if (option == OZ_NET_OPT_CONV3_TILE) {
        OZ_REQUIRE(value == 0 || value == 128 || value == 192 || value == 256, "OZ_NET_OPT_CONV3_TILE must be 0, 128, 192 or 256 (got %d)", value);
        o->conv3_tile = value;
    }
    else if (option == OZ_NET_OPT_LATENCY_SPLITS) { if (o->latency_splits != (value != 0)) { o->latency_splits = value != 0; o->committed = false; } }
    else if (option == OZ_NET_OPT_W_TARGET_LOG2) {
        OZ_REQUIRE(value >= -12 && value <= 15, "OZ_NET_OPT_W_TARGET_LOG2 must be in [-12, 15] (got %d)", value);
        if (o->w_target_log2 != value) { o->w_target_log2 = value; o->committed = false; }
    } else if (option == OZ_NET_OPT_SELF_CHECK) {
        OZ_REQUIRE(value >= 0 && value <= 2, "OZ_NET_OPT_SELF_CHECK must be 0 (off), 1 (enforce) or 2 (measure only); got %d", value);
        if (o->self_check != value) { o->self_check = value; o->committed = false; }
    } else if (option == OZ_NET_OPT_ACT_TARGET_LOG2) {
        OZ_REQUIRE(value >= -12 && value <= 20, "OZ_NET_OPT_ACT_TARGET_LOG2 must be in [-12, 20] (got %d)", value);
        if (o->act_target_log2 != value) { o->act_target_log2 = value; o->committed = false; }
    } else {
        OZ_REQUIRE(value <= 15, "OZ_NET_OPT_LOW_GUARD_LOG2 must be <= 15 (got %d)", value);
        if (o->low_guard_log2 != value) { o->low_guard_log2 = value; o->committed = false; }
    }
    return OZ_OK;
}

// what the self-check of the last commit in precision f16x2 measured (negative: it did not run)
OZ_API int oz_net_self_check(oz_net* net, double* max_dpi, double* max_dv, int* positions) {
    OnnNet* o = as_onn(net);
    OZ_REQUIRE(o, "not an OthelloNN network");
    std::lock_guard<std::mutex> lk(o->mu);
    if (max_dpi) *max_dpi = o->sc_dpi;
    if (max_dv) *max_dv = o->sc_dv;
    if (positions) *positions = o->sc_positions;
    return OZ_OK;
}

// the power-of-two exponents precision f16x2 chose at the last commit: which = 0 .. 4 the per-channel activation exponents of conv1 .. conv4, fc1
// outputs; 5 .. 9 the per-column weight exponents of conv2 .. fc2
OZ_API int oz_net_get_scaling(oz_net* net, int which, int32_t* out, int64_t nelem) {
    OnnNet* o = as_onn(net);
    OZ_REQUIRE(o && out, "not an OthelloNN network / null argument");
    OZ_REQUIRE(which >= 0 && which <= 9, "oz_net_get_scaling: which must be 0 .. 9 (got %d)", which);
    std::lock_guard<std::mutex> lk(o->mu);
    OZ_REQUIRE(o->precision == 1 && o->committed, "oz_net_get_scaling: the network is not committed in precision f16x2");
    const std::vector<int>& e = which < 5 ? o->aexp[which] : o->wexp[which - 5];
    OZ_REQUIRE(nelem == (int64_t)e.size(), "oz_net_get_scaling(%d): %lld elements expected, got %lld", which, (long long)e.size(), (long long)nelem);
    for (size_t i = 0; i < e.size(); ++i) out[i] = e[i];
    return OZ_OK;
}

OZ_API int oz_net_get_info(oz_net* net, int what, int* value) {
    OnnNet* o = as_onn(net);
    OZ_REQUIRE(o && value, "not an OthelloNN network / null argument");
    OZ_REQUIRE(what == OZ_NET_INFO_CONV3_TILE_ROWS || what == OZ_NET_INFO_SELF_CHECK_GUARD || what == OZ_NET_INFO_ARITHMETIC, "unknown network info %d", what);
    std::lock_guard<std::mutex> lk(o->mu);
    *value = what == OZ_NET_INFO_CONV3_TILE_ROWS ? o->last_conv3_rows : what == OZ_NET_INFO_ARITHMETIC ? (o->precision == 2 && !o->use_b3() ? 0 : o->precision) : o->sc_guard;
    return OZ_OK;
}

OZ_API int oz_net_profiled_layer(oz_net* net, int* layer) {
    OnnNet* o = as_onn(net);
    OZ_REQUIRE(o && layer, "not an OthelloNN network / null argument");
    std::lock_guard<std::mutex> lk(o->mu);
    *layer = o->profiled_layer;
    return OZ_OK;
}

OZ_API int oz_net_profile_read(oz_net* net, double* conv2_ms_total, int64_t* conv2_launches) {
    OnnNet* o = as_onn(net);
    OZ_REQUIRE(o, "not an OthelloNN network");
    std::lock_guard<std::mutex> lk(o->mu);
    hipSetDevice(o->device);
    if (int rc = o->collect_profile()) return rc;
    const int slot = o->profiled_layer - 1;
    if (conv2_ms_total) *conv2_ms_total = o->timer.ms[slot];
    if (conv2_launches) *conv2_launches = o->timer.count[slot];
    return OZ_OK;
}

OZ_API int oz_net_profile_kernels(oz_net* net, double* ms_total, int64_t* launches, int reset) {
    OnnNet* o = as_onn(net);
    OZ_REQUIRE(o, "not an OthelloNN network");
    std::lock_guard<std::mutex> lk(o->mu);
    hipSetDevice(o->device);
    if (int rc = o->collect_profile()) return rc;
    for (int i = 0; i < OZ_NET_KERNELS; ++i) {
        if (ms_total) ms_total[i] = o->timer.ms[i];
        if (launches) launches[i] = o->timer.count[i];
    }
    if (reset) o->timer.reset();
    return OZ_OK;
}
