// oz_net_b3.h -- "f32 via 3 x bf16 split" convolution / dense GEMM (precision bf16x3; included by oz_net.hip).
//
// Why: the fp32 matrix cores peak at 157 TFLOP/s, 1/16 of the 16-bit MFMA rate, and k_gemm_f32 sits at 0.85 of that roof.  The
// f16x2 mode (oz_net_h2.h) reaches the 16-bit pipes but carries 22 of fp32's 24 significand bits in fp16's narrow exponent range,
// which is what its commit-time placement, guards and self-check are for.  This mode carries an fp32 value EXACTLY:
//     x = b1 + b2 + b3,   b1 = bf16(x), b2 = bf16(x - b1), b3 = bf16(x - b1 - b2)
// bf16 has fp32's exponent range and 8 significand bits, so three planes hold all 24 bits of every normal fp32 value (round to
// nearest: |x - b1| <= 2^-8 |x|, |x - b1 - b2| <= 2^-17 |x|, the third residual is exact; the only loss is a residual below the smallest
// normal fp32 / bf16's subnormal step, i.e. for |x| < 2^-100: absolute error < 2^-120) -- no scaling, no calibration, no guards, no refusal path.  A product keeps six of
// the nine cross terms,
//     a b ~= a3 b1 + a1 b3 + a2 b2 + a2 b1 + a1 b2 + a1 b1          (dropped: a2 b3, a3 b2 <= 2^-25 |a b| each, a3 b3 <= 2^-34 |a b|)
// each an exact bf16 x bf16 product accumulated in fp32 on v_mfma_f32_16x16x32_bf16, small terms first.  Per product the dropped
// terms stay within fp32's own rounding of that product (half an ulp, 2^-24, in the worst case; a quarter of it typically); what remains is the fp32
// accumulation order, as in k_gemm_f32.
// Cost: 6 MFMAs at 16x the fp32 rate = 2.67x the fp32 matrix roof (cap 2500 / 6 = 417 TFLOP/s fp32-equivalent).
//
// Storage ("b3 layout"): a row (pixel or output channel) is cut into k-tiles of 32 channels; a k-tile is 192 bytes =
// [plane 0: 32 bf16][plane 1: 32 bf16][plane 2: 32 bf16] = 12 chunks of 16 B, chunk (plane p, group kg of 8 channels) at index 4 p + kg.
// A row of K values is K / 32 * 192 = 6 K bytes.  Weight rows [N][K'] use the GEMM's tap-inner k order of oz_net_h2.h:
// k' = (slice * taps + tap) * 32 + c32 for input channel 32 * slice + c32.
//
// Kernel k_gemm_b3: implicit GEMM, block tile 128 x 256 x 32, 8 waves (2 x 4, wave tile 64 x 64 = 4 x 4 MFMA tiles of 16 x 16),
// two LDS stages of 72 KB (144 of the CU's 160 KB: with 6 B per element a 256 x 256 tile no longer fits twice), both operands by
// 16-byte LDS-DMA, and the 2-phase ping-pong main loop of k_gemm_h2's 128 / 192-row tiles -- the two wave rows half a phase apart, one
// in its MFMA section (48 MFMAs = 768 cycles here) while the other reads fragments and issues DMA:
//   phase 1  L: A of tile t (12 ds_read_b128);  DMA [B n1 of t + 1] [B n0 of t + 2];  M: (m, n0) -- 4 x 2 tiles x 6 products
//   phase 2  L: B n1 of t, B n0 of t + 1 (12 reads);  DMA [A of t + 2];                M: (m, n1)
// LDS image: rows of 192 B, a 16-row block = 3 KB = three DMA instructions (a DMA instruction writes 64 lanes x 16 B linearly, so
// its lanes span 5.3 rows: every lane carries its own row's source address, computed once before the loop).  The chunk inside a
// plane is XOR-swizzled with g(row) = (-(row >> 2)) & 3: a ds_read_b128 is served in 16-lane groups {0-3, 12-15, 20-27}, ... (oz_net_h2.h,
// h2_swz); with the 16x16x32 operand map (lane l: row l & 15, k-group l >> 4) such a group holds rows {0-3, 12-15} with k-group a
// and rows {4-11} with k-group a ^ 1; the row stride of 12 chunks puts row r into bank quad (3 r + p) & 3, so the four rows of one
// quad -- r, r + 4, r + 8, r + 12 -- need four different chunk slots: {a ^ g0, a ^ 1 ^ g1, a ^ 1 ^ g2, a ^ g3} = all four for g = (0, 3, 2, 1).
// Every output element receives its products in one fixed order (k-tile by k-tile, the six plane products in the order above), so
// a position's result does not depend on its place in the batch or on the batch size.
#pragma once

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define B3_BK 32
#define B3_BM 128
#define B3_BN 256
#define B3_NT 512
#define B3_ROWB 192                               // bytes of one row's k-tile (3 planes x 32 bf16)
#define B3_BLK (16 * B3_ROWB)                     // a 16-row block of an operand tile: 3 KB = 3 DMA instructions
#define B3_TILEA (B3_BM * B3_ROWB)                // 24 KB
#define B3_TILEB (B3_BN * B3_ROWB)                // 48 KB
#define B3_BUF (B3_TILEA + B3_TILEB)              // 72 KB per stage
#define B3_LDS (2 * B3_BUF)                       // 144 KB
#define B3_SLICE (32 * 384)                       // epilogue: a wave transposes 32 rows x 64 channels x 3 planes per pass (12 KB)

struct B3Geom {
    int Hin, Hout, pad, Cin, taps;                // taps 9 (3x3 conv) or 1 (dense, Hin = Hout = 1)
    int N, K;                                     // output channels, taps * Cin
    int out_b3;                                   // 1: write the b3 layout, 0: write fp32 rows
    int relu;
    int ksplit;                                   // > 1: raw fp32 partial sums to slab[ks] (no scale / shift); a fixed-order reduce finishes the layer
    long long slab;                               // floats between two partial slabs
};

__device__ __forceinline__ void b3_split(float x, __bf16& b1, __bf16& b2, __bf16& b3) {
    b1 = (__bf16)x;
    // the top 0.4 % of fp32's range (|x| > 0x7F7F8000 = bf16's largest value + half an ulp) would ROUND to infinity: take bf16's largest value instead --
    // the residual (< 2^120) still fits the other two planes exactly
    if (__builtin_isinf((float)b1) && !__builtin_isinf(x)) b1 = __builtin_bit_cast(__bf16, (unsigned short)(x < 0.f ? 0xFF7Fu : 0x7F7Fu));
    const float r1 = x - (float)b1;               // exact
    b2 = (__bf16)r1;
    b3 = (__bf16)(r1 - (float)b2);                // exact difference, at most 8 significant bits: the cast is exact
}

// fp32 rows [rows][C] -> the b3 layout; rows = *d_count * P.  One thread per (row, 8 channels): two 16-byte loads, three 16-byte stores.
__global__ __launch_bounds__(256) void k_f32_to_b3(const float* __restrict__ x, const int* __restrict__ d_count, int P, int C, uint4* __restrict__ out) {
    const int cg = C >> 3;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long row = idx / cg;
    if (row >= (long long)(*d_count) * P) return;
    const int c8 = (int)(idx % cg) * 8;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(x + (size_t)row * C + c8), hi = *reinterpret_cast<const f32x4*>(x + (size_t)row * C + c8 + 4);
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 a, b, c;
        b3_split(j < 4 ? lo[j] : hi[j - 4], a, b, c);
        p0[j] = a; p1[j] = b; p2[j] = c;
    }
    uint4* dst = out + (size_t)row * (size_t)(C / 32 * 12) + (c8 >> 5) * 12 + ((c8 >> 3) & 3);
    dst[0] = *reinterpret_cast<uint4*>(&p0);
    dst[4] = *reinterpret_cast<uint4*>(&p1);
    dst[8] = *reinterpret_cast<uint4*>(&p2);
}

// finishes a split-K layer whose consumer reads the b3 layout: out[m][n] = relu((sum_s slab[s][m][n]) * scale[n] + shift[n]), the slices added in a
// FIXED order (s = 0, 1, ...: bit-reproducible), then split into the three planes.  One thread per (row, 8 channels).
__global__ __launch_bounds__(256) void k_splitk_reduce_b3(const float* __restrict__ part, long long slab, int ksplit, int N, int P,
                                                          const int* __restrict__ d_count, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, uint4* __restrict__ out) {
    const int ng = N >> 3;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long m = idx / ng;
    if (m >= (long long)(*d_count) * P) return;
    const int c8 = (int)(idx % ng) * 8;
    const float* p = part + (size_t)m * N + c8;
    f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
    for (int s = 1; s < ksplit; ++s) {
        lo += *reinterpret_cast<const f32x4*>(p + (size_t)s * slab);
        hi += *reinterpret_cast<const f32x4*>(p + (size_t)s * slab + 4);
    }
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = fmaxf(fmaf(j < 4 ? lo[j] : hi[j - 4], scale[c8 + j], shift[c8 + j]), 0.f);
        __bf16 a, b, c;
        b3_split(v, a, b, c);
        p0[j] = a; p1[j] = b; p2[j] = c;
    }
    uint4* dst = out + (size_t)m * (size_t)(N / 32 * 12) + (c8 >> 5) * 12 + ((c8 >> 3) & 3);
    dst[0] = *reinterpret_cast<uint4*>(&p0);
    dst[4] = *reinterpret_cast<uint4*>(&p1);
    dst[8] = *reinterpret_cast<uint4*>(&p2);
}

// weights as stored by Keras, [K][N] with k = tap * Cin + ci  ->  b3 rows [N][K / 32][3 planes][32] in the GEMM's k order
// k' = (slice * taps + tap) * 32 + c32.  One thread per (output channel c, group of 8 k'); adjacent threads = adjacent c.
__global__ __launch_bounds__(256) void k_w_to_b3(const float* __restrict__ src, int K, int N, int taps, uint4* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(idx % N);
    const int grp = (int)(idx / N);
    if (grp >= (K >> 3)) return;
    const int Cin = K / taps;
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int kp = grp * 8 + j, tile = kp >> 5, c32 = kp & 31, slice = tile / taps, tap = tile - slice * taps;
        const int k = tap * Cin + slice * 32 + c32;
        __bf16 a, b, cc;
        b3_split(src[(size_t)k * N + c], a, b, cc);
        p0[j] = a; p1[j] = b; p2[j] = cc;
    }
    uint4* dst = out + (size_t)c * (size_t)(K / 32 * 12) + (grp >> 2) * 12 + (grp & 3);
    dst[0] = *reinterpret_cast<uint4*>(&p0);
    dst[4] = *reinterpret_cast<uint4*>(&p1);
    dst[8] = *reinterpret_cast<uint4*>(&p2);
}

// out[M][N] = act((A[M][K] . W[N][K]^T) * scale + shift); A and W in the b3 layout; M = *d_count * Hout^2.
// zero_line: >= 128 B of zeros in global memory (source of out-of-image taps and of rows beyond M).
// TAG: no effect on the code -- one symbol per OthelloNN layer (3 = conv3, 4 = conv4, 5 = fc1) so that rocprofv3 lists them separately.
template <int TAG = 0>
__global__ __launch_bounds__(B3_NT, 2) void k_gemm_b3(const uint4* __restrict__ in, const uint4* __restrict__ Wb,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      void* __restrict__ out, const int* __restrict__ d_count, B3Geom g,
                                                      int num_mt, const uint4* __restrict__ zero_line) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = B3_BM, BN = B3_BN, RI = 4, RJ = 4;
    // block mapping of k_gemm_h2: the column tiles (and k-slices) of one row tile run on the same XCD; launches with fewer than 8 row tiles
    // spread the weight slices over the XCDs instead
    const int nnt = g.N / BN, per_mt = nnt * g.ksplit;
    const int id = blockIdx.x, xcd = id & 7, jb = id >> 3;
    int mt, nt, ks;
    if (num_mt < 8) {
        mt = jb % num_mt;
        const int q = (jb / num_mt) * 8 + xcd;
        if (q >= per_mt) return;
        nt = q / g.ksplit; ks = q - nt * g.ksplit;
    } else {
        const int rem = jb % per_mt;
        mt = (jb / per_mt) * 8 + xcd; nt = rem / g.ksplit; ks = rem - nt * g.ksplit;
    }
    const int P = g.Hout * g.Hout;
    const long long M = (long long)(*d_count) * P;
    if (mt >= num_mt || (long long)mt * BM >= M) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int rowq = g.Cin / 32 * 12;                        // uint4 units per input pixel row
    const int wrowq = g.K / 32 * 12;                         // uint4 units per weight row

    // ---- staging map.  A: wave w stages the 16-row block w (3 instructions).  B: wave w stages block (w >> 1) * 4 + (w & 1) of the n0
    // halves ("early") and block (w >> 1) * 4 + 2 + (w & 1) of the n1 halves ("late") of the wave columns' 64-column tiles.
    // Lane l of instruction i fills chunk 64 i + l of the block: row (64 i + l) / 12, slot (64 i + l) % 12 = 4 p + (kg ^ g(row)).
    long long aidx[3];
    unsigned amask[3];
    unsigned bidx_e[3], bidx_l[3];
    const int eb = (wave >> 1) * 4 + (wave & 1), lb = eb + 2;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int cidx = 64 * i + lane, r = cidx / 12, pos = cidx - 12 * r, pl = pos >> 2, kgs = (pos & 3) ^ ((4 - (r >> 2)) & 3);
        const int src_chunk = pl * 4 + kgs;
        const long long m = (long long)mt * BM + wave * 16 + r;
        aidx[i] = 0; amask[i] = 0;
        if (m < M) {
            const int b = (int)(m / P), pix = (int)(m % P), oy = pix / g.Hout, ox = pix % g.Hout;
            aidx[i] = (((long long)b * g.Hin + (oy - g.pad)) * g.Hin + (ox - g.pad)) * rowq + src_chunk;
            unsigned mk = 0;
            for (int t = 0; t < g.taps; ++t) {
                const int iy = oy - g.pad + t / 3, ix = ox - g.pad + t % 3;
                if (iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Hin) mk |= 1u << t;
            }
            amask[i] = mk;
        }
        bidx_e[i] = (unsigned)(nt * BN + eb * 16 + r) * (unsigned)wrowq + (unsigned)src_chunk;
        bidx_l[i] = (unsigned)(nt * BN + lb * 16 + r) * (unsigned)wrowq + (unsigned)src_chunk;
    }
    const uint4* zsrc = zero_line + (lane & 7);
    const int nk_all = g.K / B3_BK, kbeg = (int)((long long)nk_all * ks / g.ksplit), nk = (int)((long long)nk_all * (ks + 1) / g.ksplit);

    auto put_a = [&](int slice, int tap, unsigned char* la) {
        const int dy = (tap * 11) >> 5, dx = tap - 3 * dy;                                      // tap / 3, tap % 3 for tap < 9
        const long long toff = ((long long)dy * g.Hin + dx) * rowq + slice * 12;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const uint4* ga = ((amask[i] >> tap) & 1) ? in + (aidx[i] + toff) : zsrc;
            __builtin_amdgcn_global_load_lds((h2_gptr)ga, (h2_lptr)(la + wave * B3_BLK + i * 1024), 16, 0, 0);
        }
    };
    auto put_be = [&](int ktc, unsigned char* lbp) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_global_load_lds((h2_gptr)(Wb + bidx_e[i] + ktc * 12), (h2_lptr)(lbp + eb * B3_BLK + i * 1024), 16, 0, 0);
    };
    auto put_bl = [&](int ktc, unsigned char* lbp) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_global_load_lds((h2_gptr)(Wb + bidx_l[i] + ktc * 12), (h2_lptr)(lbp + lb * B3_BLK + i * 1024), 16, 0, 0);
    };

    f32x4v acc[RI][RJ];
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
        for (int j = 0; j < RJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    const int r16 = lane & 15, kg = lane >> 4;
    const int lofs = r16 * B3_ROWB + ((kg ^ ((4 - (r16 >> 2)) & 3)) * 16);     // this lane's chunk of plane 0 inside a 16-row block

    constexpr int KEEP = 9;                                  // youngest DMA instructions that may still be in flight at the end of an L section
    int k1 = kbeg + 1 < nk ? kbeg + 1 : nk - 1, slice1 = k1 / g.taps, tap1 = k1 - slice1 * g.taps;      // tile kt + 1, kt + 2 (past the end: the last again)
    int k2 = kbeg + 2 < nk ? kbeg + 2 : nk - 1, slice2 = k2 / g.taps, tap2 = k2 - slice2 * g.taps;
    {   // tile kbeg whole, then what phases 1 and 2 of the tile before the first would have issued: [B n0] [A] of tile kbeg + 1
        const int slice0 = kbeg / g.taps, tap0 = kbeg - slice0 * g.taps;
        put_a(slice0, tap0, smem);
        put_be(kbeg, smem + B3_TILEA); put_bl(kbeg, smem + B3_TILEA);
        put_be(k1, smem + B3_BUF + B3_TILEA);
        put_a(slice1, tap1, smem + B3_BUF);
    }
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");         // tile kbeg has landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    bf16x8 fa[3][RI], fb[3][RJ];
    {   // B n0 of the first tile (later tiles get it in phase 2 of the tile before)
        const unsigned char* Bt0 = smem + B3_TILEA + wn * 4 * B3_BLK + lofs;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) fb[p][j] = *reinterpret_cast<const bf16x8*>(Bt0 + j * B3_BLK + p * 64);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                            // every wave's prologue reads are complete before any wave refills B n0 of stage 0
    asm volatile("" ::: "memory");
    if (wm == 1) __builtin_amdgcn_s_barrier();               // stagger: wave row 1 is one barrier behind
    for (int kt = kbeg; kt < nk; ++kt) {
        const int buf = (kt - kbeg) & 1;
        unsigned char* la_cur = smem + (size_t)buf * B3_BUF;             // stage of tile kt (and of tile kt + 2)
        unsigned char* la_oth = smem + (size_t)(buf ^ 1) * B3_BUF;       // stage of tile kt + 1
        const unsigned char* At = la_cur + wm * 4 * B3_BLK + lofs;
        const unsigned char* Bt = la_cur + B3_TILEA + wn * 4 * B3_BLK + lofs;
        const unsigned char* Btn = la_oth + B3_TILEA + wn * 4 * B3_BLK + lofs;
        auto ldb = [&](int half, const unsigned char* base) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) fb[p][half * 2 + j] = *reinterpret_cast<const bf16x8*>(base + (half * 2 + j) * B3_BLK + p * 64);
        };
        auto l_end = [&]() {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KEEP) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        auto mma = [&](int nh) {                             // (m, nh): 6 products x 4 x 2 tiles, product-major, small terms first
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int i = 0; i < RI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x4v& c = acc[i][nh * 2 + j];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[PA[q]][i], fb[PB[q]][nh * 2 + j], c, 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto m_end = [&]() {
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        // phase 1
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) fa[p][i] = *reinterpret_cast<const bf16x8*>(At + i * B3_BLK + p * 64);
        put_bl(k1, la_oth + B3_TILEA); put_be(k2, la_cur + B3_TILEA);
        l_end(); mma(0); m_end();
        // phase 2
        ldb(1, Bt); ldb(0, Btn);
        put_a(slice2, tap2, la_cur);
        l_end(); mma(1); m_end();
        k1 = k2; slice1 = slice2; tap1 = tap2;
        if (k2 + 1 < nk) { ++k2; if (++tap2 == g.taps) { tap2 = 0; ++slice2; } }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();               // re-align the two wave rows
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every piece has landed before the epilogue reuses the LDS
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- epilogue.  C/D layout of 16x16: col = lane & 15, row = (lane >> 4) * 4 + reg
    if (g.ksplit > 1) {      // raw fp32 partial sums of this k-slice into slab ks
        float* o = reinterpret_cast<float*>(out) + (size_t)ks * g.slab;
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            const int col = nt * BN + wn * 64 + j * 16 + r16;
#pragma unroll
            for (int i = 0; i < RI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long m = (long long)mt * BM + wm * 64 + i * 16 + kg * 4 + r;
                    if (m < M) o[(size_t)m * g.N + col] = acc[i][j][r];
                }
        }
        return;
    }
    if (!g.out_b3) {
        float* o = reinterpret_cast<float*>(out);
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            const int col = nt * BN + wn * 64 + j * 16 + r16;
            const float sc = scale[col], sh = shift[col];
#pragma unroll
            for (int i = 0; i < RI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long m = (long long)mt * BM + wm * 64 + i * 16 + kg * 4 + r;
                    float v = fmaf(acc[i][j][r], sc, sh);
                    if (g.relu) v = fmaxf(v, 0.f);
                    if (m < M) o[(size_t)m * g.N + col] = v;
                }
        }
        return;
    }
    // b3 output: each wave transposes its 64 x 64 tile through its own 12 KB LDS slice, 32 rows at a time:
    // slice[row][k-tile (2)][plane (3)][32 bf16] = 384 B per row; then 16-byte chunks go out, 24 consecutive lanes per row.
    __bf16* slice = reinterpret_cast<__bf16*>(smem + wave * B3_SLICE);
    uint4* o = reinterpret_cast<uint4*>(out);
    const int nq = g.N / 32 * 12;                            // uint4 units per output row
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = hh * 2 + ii;
#pragma unroll
            for (int j = 0; j < RJ; ++j) {
                const int lc = j * 16 + r16;                 // column inside the wave tile
                const int col = nt * BN + wn * 64 + lc;
                const float sc = scale[col], sh = shift[col];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lr = ii * 16 + kg * 4 + r;     // row inside the pass
                    float v = fmaf(acc[i][j][r], sc, sh);
                    if (g.relu) v = fmaxf(v, 0.f);
                    __bf16 b1, b2, b3;
                    b3_split(v, b1, b2, b3);
                    __bf16* p = slice + lr * 192 + (lc >> 5) * 96 + (lc & 31);
                    p[0] = b1; p[32] = b2; p[64] = b3;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): this wave's LDS writes have landed
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 12; ++c) {                       // 32 rows x 24 chunks of 16 B
            const int q = c * 64 + lane, lr = q / 24, cq = q - lr * 24;
            const long long m = (long long)mt * BM + wm * 64 + hh * 32 + lr;
            const uint4 val = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(slice) + lr * 384 + cq * 16);
            if (m < M) o[(size_t)m * nq + ((nt * BN + wn * 64) >> 5) * 12 + cq] = val;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------- the 256 x 256 tile (round 6, second half)
// Why: the 128 x 256 tile moves 72 KB of operands per 96 MFMAs of a wave pair; with every CU holding a block that is ~10 TB/s of LDS-DMA out of
// the L2s -- measured: a k-tile takes 1.90 us at 256 resident blocks and 1.55 us at 130 (tools/b3_probe.py), while the MFMAs alone would take
// 1.48 us at the 2.07 GHz a pure bf16 MFMA loop sustains (oz_selftest_mfma_rate kind 2).  The kernel is bound by the operand stream, not by the
// matrix pipe.  A 256 x 256 tile needs 96 KB per 192 MFMAs: two thirds of the bytes per MFMA.
// LDS: two whole stages (192 KB) do not exist.  Five 24 KB regions do (120 KB): A m0, A m1, B n1 and TWO B n0 -- every region is refilled by LDS-DMA
// in the phase right after its last ds_read (each L section closes with lgkmcnt(0) BEFORE its barrier, so every wave's reads of a phase are complete
// before any wave issues the next phase's DMA), and is read three phases (B n0: four) after its issue.
// Registers: 128 accumulators leave room for ONE half of each operand's fragments (48 + 24 VGPRs), so the four quadrants of the 128 x 64 wave tile
// are walked (m0,n0) (m0,n1) (m1,n1) (m1,n0) -- one operand half changes per phase -- and B n0 is read twice per k-tile (phases 1 and 4), which is why
// it is the double-buffered region.  Per k-tile and wave: 42 ds_read_b128, 12 DMA instructions, 192 MFMAs in four sections of 48.
//   phase 1  L: A m0(t), B n0(t);  DMA [B n0(t+1)] -> the other B n0 buffer;   M: (m0, n0)
//   phase 2  L: B n1(t);           DMA [A m0(t+1)];                             M: (m0, n1)
//   phase 3  L: A m1(t);           DMA [B n1(t+1)];                             M: (m1, n1)
//   phase 4  L: B n0(t) again;     DMA [A m1(t+1)];                             M: (m1, n0)
// vmcnt(6) at the end of every L section: all but the two youngest groups of three pieces have landed -- exactly what the next phase reads.
// Every output element receives the same products in the same order as on the 128 x 256 tile: bit-identical (OZ_NET_OPT_B3_TILE screens it).
// Requires pad == 0 (the 'valid' convolutions and the dense layers: a row's taps are all inside the image), rows beyond M read the zero line.
#define B3B_BM 256
#define B3B_BN 256
#define B3B_REG (8 * B3_BLK)                      // 24 KB: one half (8 blocks of 16 rows) of an operand tile
#define B3B_LDS (5 * B3B_REG)                     // 120 KB: A m0 | A m1 | B n0 [2] | B n1
template <int TAG = 0>
__global__ __launch_bounds__(B3_NT, 2) void k_gemm_b3_big(const uint4* __restrict__ in, const uint4* __restrict__ Wb,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          void* __restrict__ out, const int* __restrict__ d_count, B3Geom g,
                                                          int num_mt, const uint4* __restrict__ zero_line) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = B3B_BM, BN = B3B_BN, HA = 4, RJ = 4;
    unsigned char* const rA0 = smem;
    unsigned char* const rA1 = smem + B3B_REG;
    unsigned char* const rB0 = smem + 2 * B3B_REG;           // two buffers: + B3B_REG for odd tiles
    unsigned char* const rB1 = smem + 4 * B3B_REG;
    const int nnt = g.N / BN, per_mt = nnt * g.ksplit;
    const int id = blockIdx.x, xcd = id & 7, jb = id >> 3;
    int mt, nt, ks;
    if (num_mt < 8) {
        mt = jb % num_mt;
        const int q = (jb / num_mt) * 8 + xcd;
        if (q >= per_mt) return;
        nt = q / g.ksplit; ks = q - nt * g.ksplit;
    } else {
        const int rem = jb % per_mt;
        mt = (jb / per_mt) * 8 + xcd; nt = rem / g.ksplit; ks = rem - nt * g.ksplit;
    }
    const int P = g.Hout * g.Hout;
    const long long M = (long long)(*d_count) * P;
    if (mt >= num_mt || (long long)mt * BM >= M) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int rowq = g.Cin / 32 * 12, wrowq = g.K / 32 * 12;

    // staging map: wave w fills block w (of 8) of each region -- A halves: block (w >> 2, w & 3) = rows (w >> 2) * 128 + half * 64 + (w & 3) * 16;
    // B halves: block (w >> 1, w & 1) = columns (w >> 1) * 64 + half * 32 + (w & 1) * 16.  ~0u = a row beyond M: the zero line.
    unsigned aoff[2][3], boff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int cidx = 64 * i + lane, r = cidx / 12, pos = cidx - 12 * r, src_chunk = (pos >> 2) * 4 + ((pos & 3) ^ ((4 - (r >> 2)) & 3));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long m = (long long)mt * BM + (wave >> 2) * 128 + h * 64 + (wave & 3) * 16 + r;
            aoff[h][i] = ~0u;
            if (m < M) {
                const int b = (int)(m / P), pix = (int)(m % P), oy = pix / g.Hout, ox = pix % g.Hout;
                aoff[h][i] = (unsigned)((((long long)b * g.Hin + oy) * g.Hin + ox) * rowq + src_chunk);
            }
        }
        boff[i] = (unsigned)(nt * BN + (wave >> 1) * 64 + (wave & 1) * 16 + r) * (unsigned)wrowq + (unsigned)src_chunk;
    }
    const unsigned bhalf = 32u * (unsigned)wrowq;            // B n1 rows = B n0 rows + 32
    const uint4* zsrc = zero_line + (lane & 7);
    const int nk_all = g.K / B3_BK, kbeg = (int)((long long)nk_all * ks / g.ksplit), nk = (int)((long long)nk_all * (ks + 1) / g.ksplit);

    auto put_a = [&](int h, int slice, int tap, unsigned char* reg) {
        const int dy = (tap * 11) >> 5, dx = tap - 3 * dy;
        const unsigned toff = (unsigned)((dy * g.Hin + dx) * rowq + slice * 12);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const uint4* ga = aoff[h][i] != ~0u ? in + (aoff[h][i] + toff) : zsrc;
            __builtin_amdgcn_global_load_lds((h2_gptr)ga, (h2_lptr)(reg + wave * B3_BLK + i * 1024), 16, 0, 0);
        }
    };
    auto put_b = [&](int h, int ktc, unsigned char* reg) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_global_load_lds((h2_gptr)(Wb + (boff[i] + (h ? bhalf : 0u) + (unsigned)ktc * 12u)), (h2_lptr)(reg + wave * B3_BLK + i * 1024), 16, 0, 0);
    };

    f32x4v acc[2 * HA][RJ];
#pragma unroll
    for (int i = 0; i < 2 * HA; ++i)
#pragma unroll
        for (int j = 0; j < RJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    const int r16 = lane & 15, kg = lane >> 4;
    const int lofs = r16 * B3_ROWB + ((kg ^ ((4 - (r16 >> 2)) & 3)) * 16);
    const unsigned char* const At0 = rA0 + wm * HA * B3_BLK + lofs;
    const unsigned char* const At1 = rA1 + wm * HA * B3_BLK + lofs;
    const unsigned char* const Bt1 = rB1 + wn * 2 * B3_BLK + lofs;

    constexpr int KEEP = 6;
    int ktn = kbeg + 1 < nk ? kbeg + 1 : nk - 1, slice_n = ktn / g.taps, tap_n = ktn - slice_n * g.taps;      // the tile being staged (past the end: the last again)
    {   // tile kbeg in the loop's issue order: [B n0] [A m0] [B n1] [A m1]
        const int slice0 = kbeg / g.taps, tap0 = kbeg - slice0 * g.taps;
        put_b(0, kbeg, rB0); put_a(0, slice0, tap0, rA0); put_b(1, kbeg, rB1); put_a(1, slice0, tap0, rA1);
    }
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KEEP) : "memory");     // B n0 and A m0 of tile kbeg have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wm == 1) __builtin_amdgcn_s_barrier();               // stagger: wave row 1 is one barrier behind
    bf16x8 fa[3][HA], fb[3][2];
    for (int kt = kbeg; kt < nk; ++kt) {
        const int par = (kt - kbeg) & 1;
        unsigned char* const rB0cur = rB0 + par * B3B_REG;
        unsigned char* const rB0nxt = rB0 + (par ^ 1) * B3B_REG;
        const unsigned char* const Bt0 = rB0cur + wn * 2 * B3_BLK + lofs;
        auto lda = [&](const unsigned char* base) {
#pragma unroll
            for (int i = 0; i < HA; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) fa[p][i] = *reinterpret_cast<const bf16x8*>(base + i * B3_BLK + p * 64);
        };
        auto ldb = [&](const unsigned char* base) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) fb[p][j] = *reinterpret_cast<const bf16x8*>(base + j * B3_BLK + p * 64);
        };
        auto l_end = [&]() {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KEEP) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        auto mma = [&](int mh, int nh) {                     // quadrant (mh, nh): 6 products x 4 x 2 tiles, product-major, small terms first
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int i = 0; i < HA; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x4v& c = acc[mh * HA + i][nh * 2 + j];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[PA[q]][i], fb[PB[q]][j], c, 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto m_end = [&]() {
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        lda(At0); ldb(Bt0); put_b(0, ktn, rB0nxt); l_end(); mma(0, 0); m_end();             // phase 1
        ldb(Bt1); put_a(0, slice_n, tap_n, rA0); l_end(); mma(0, 1); m_end();               // phase 2
        lda(At1); put_b(1, ktn, rB1); l_end(); mma(1, 1); m_end();                          // phase 3
        ldb(Bt0); put_a(1, slice_n, tap_n, rA1); l_end(); mma(1, 0); m_end();               // phase 4
        if (ktn + 1 < nk) { ++ktn; if (++tap_n == g.taps) { tap_n = 0; ++slice_n; } }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();               // re-align the two wave rows
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every piece has landed before the epilogue reuses the LDS
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- epilogue (as k_gemm_b3's; the wave tile is 128 x 64).  C/D layout of 16x16: col = lane & 15, row = (lane >> 4) * 4 + reg
    if (g.ksplit > 1) {
        float* o = reinterpret_cast<float*>(out) + (size_t)ks * g.slab;
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            const int col = nt * BN + wn * 64 + j * 16 + r16;
#pragma unroll
            for (int i = 0; i < 2 * HA; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long m = (long long)mt * BM + wm * 128 + i * 16 + kg * 4 + r;
                    if (m < M) o[(size_t)m * g.N + col] = acc[i][j][r];
                }
        }
        return;
    }
    if (!g.out_b3) {
        float* o = reinterpret_cast<float*>(out);
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            const int col = nt * BN + wn * 64 + j * 16 + r16;
            const float sc = scale[col], sh = shift[col];
#pragma unroll
            for (int i = 0; i < 2 * HA; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long m = (long long)mt * BM + wm * 128 + i * 16 + kg * 4 + r;
                    float v = fmaf(acc[i][j][r], sc, sh);
                    if (g.relu) v = fmaxf(v, 0.f);
                    if (m < M) o[(size_t)m * g.N + col] = v;
                }
        }
        return;
    }
    __bf16* slice = reinterpret_cast<__bf16*>(smem + wave * B3_SLICE);
    uint4* o = reinterpret_cast<uint4*>(out);
    const int nq = g.N / 32 * 12;
#pragma unroll
    for (int hh = 0; hh < HA; ++hh) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = hh * 2 + ii;
#pragma unroll
            for (int j = 0; j < RJ; ++j) {
                const int lc = j * 16 + r16;
                const int col = nt * BN + wn * 64 + lc;
                const float sc = scale[col], sh = shift[col];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lr = ii * 16 + kg * 4 + r;
                    float v = fmaf(acc[i][j][r], sc, sh);
                    if (g.relu) v = fmaxf(v, 0.f);
                    __bf16 b1, b2, b3;
                    b3_split(v, b1, b2, b3);
                    __bf16* p = slice + lr * 192 + (lc >> 5) * 96 + (lc & 31);
                    p[0] = b1; p[32] = b2; p[64] = b3;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            const int q = c * 64 + lane, lr = q / 24, cq = q - lr * 24;
            const long long m = (long long)mt * BM + wm * 128 + hh * 32 + lr;
            const uint4 val = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(slice) + lr * 384 + cq * 16);
            if (m < M) o[(size_t)m * nq + ((nt * BN + wn * 64) >> 5) * 12 + cq] = val;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
}
