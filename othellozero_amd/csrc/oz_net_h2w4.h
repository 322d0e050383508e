// oz_net_h2w4.h -- software-pipelined variant of the f16x2 split convolution kernel (see oz_net_h2.h for the
// arithmetic and the data layout).  Included by oz_net.hip after oz_net_h2.h.
//
// Why a second structure: in the 8-wave kernel the two waves of every SIMD belong to the same workgroup, meet at
// the same barrier and expose the same LDS / DMA latencies at the same time (measured 65 % matrix-pipe busy;
// ablation: 2.75 ms full, 2.20 ms with no DMA, 1.67 ms with no MFMA at conv2 size).  Here a workgroup is 4 waves,
// ONE PER SIMD, each with the whole 512-register budget:
//   * wave tile 128 x 128 (4 x 4 MFMA tiles, 256 accumulator registers) -> 16 fragment reads per 48 MFMAs
//     (the 8-wave kernel needs 12 per 24), block tile still 256 x 256 x 32 with two LDS stages (128 KB);
//   * fragments are double-buffered in registers: the 16 ds_read_b128 of k-step s+1 are issued under the 48 MFMAs
//     of k-step s (sched_group_barrier pins 3 MFMA : 1 read);
//   * the barrier that publishes tile kt+1 sits in the MIDDLE of tile kt's last k-step: the 24 MFMAs behind it
//     already have their operands in registers and cover the first fragment reads of tile kt+1 and the issue of the
//     next LDS-DMA instructions; a tile's DMA is in flight for a whole iteration before its vmcnt(0) wait
//     (B half issued in phase 1, A half of the tile after next issued behind the barrier).
#pragma once

struct H2W4 {
    static constexpr int BM = 256, BN = 256, NW = 4, NT = 256, TI = 4, TJ = 4;
    static constexpr int TILE = 256 * 128, BUF = 2 * TILE, LDS = 2 * BUF;      // 131 072 B
    static constexpr int IA = BM / (8 * NW), IB = BN / (8 * NW);               // 8 + 8 DMA instructions per wave per tile
};

struct H2Frag { f16x8 a1[4], a2[4], b1[4], b2[4]; };

__global__ __launch_bounds__(256, 1) void k_gemm_h2_w4(const uint4* __restrict__ in, const uint4* __restrict__ Wh,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       void* __restrict__ out, const int* __restrict__ d_count, H2Geom g,
                                                       int num_mt, const uint4* __restrict__ zero_line, int* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = H2W4::BM, BN = H2W4::BN, IA = H2W4::IA, IB = H2W4::IB, TILE = H2W4::TILE, BUF = H2W4::BUF;
    const int nnt = g.N / BN;
    const int id = blockIdx.x, xcd = id & 7, jb = id >> 3;
    const int mt = (jb / nnt) * 8 + xcd, nt = jb % nnt;
    const int P = g.Hout * g.Hout;
    const long long M = (long long)(*d_count) * P;
    if (mt >= num_mt || (long long)mt * BM >= M) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                 // 2 x 2 waves, wave tile 128 x 128
    const int rowq = g.Cin >> 2, wrowq = g.K >> 2;

    long long aidx[IA];
    unsigned amask[IA];
    unsigned bidx[IB];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int row = (wave * IA + i) * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        const long long m = (long long)mt * BM + row;
        aidx[i] = 0; amask[i] = 0;
        if (m < M) {
            const int b = (int)(m / P), pix = (int)(m % P), oy = pix / g.Hout, ox = pix % g.Hout;
            aidx[i] = (((long long)b * g.Hin + (oy - g.pad)) * g.Hin + (ox - g.pad)) * rowq + lc;
            unsigned mk = 0;
            for (int t = 0; t < g.taps; ++t) {
                const int iy = oy - g.pad + t / 3, ix = ox - g.pad + t % 3;
                if (iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Hin) mk |= 1u << t;
            }
            amask[i] = mk;
        }
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
        const int row = (wave * IB + i) * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        bidx[i] = (unsigned)(nt * BN + row) * (unsigned)wrowq + (unsigned)lc;
    }
    const uint4* zsrc = zero_line + (lane & 7);
    const int nk = g.K / H2_BK;

    auto stageA = [&](int kt_raw, int buf) {
        const int kt = kt_raw < nk ? kt_raw : nk - 1;
        const int k0 = kt * H2_BK, tap = k0 / g.Cin, ci0 = k0 - tap * g.Cin;
        const long long toff = ((long long)(tap / 3) * g.Hin + (tap % 3)) * rowq + (ci0 >> 2);
        unsigned char* la = smem + buf * BUF + wave * IA * 1024;
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const uint4* ga = ((amask[i] >> tap) & 1) ? in + (aidx[i] + toff) : zsrc;
            __builtin_amdgcn_global_load_lds((h2_gptr)ga, (h2_lptr)(la + i * 1024), 16, 0, 0);
        }
    };
    auto stageB = [&](int kt_raw, int buf) {
        const int kt = kt_raw < nk ? kt_raw : nk - 1;
        const int k0 = kt * H2_BK;
        unsigned char* lb = smem + buf * BUF + TILE + wave * IB * 1024;
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            const uint4* gb = Wh + bidx[i] + (k0 >> 2);
            __builtin_amdgcn_global_load_lds((h2_gptr)gb, (h2_lptr)(lb + i * 1024), 16, 0, 0);
        }
    };

    const int r32 = lane & 31, half = lane >> 5;
    const int swz = (r32 >> 1) & 7;
    const int o00 = ((2 * half) ^ swz) * 16, o01 = ((2 * half + 1) ^ swz) * 16;
    const int o10 = ((4 + 2 * half) ^ swz) * 16, o11 = ((4 + 2 * half + 1) ^ swz) * 16;
    const int arow = (wm * 128 + r32) * 128, brow = TILE + (wn * 128 + r32) * 128;

    auto loadF = [&](H2Frag& f, int buf, int s) {
        const unsigned char* At = smem + buf * BUF + arow;
        const unsigned char* Bt = smem + buf * BUF + brow;
        const int oh1 = s ? o10 : o00, oh2 = s ? o11 : o01;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f.a1[i] = *reinterpret_cast<const f16x8*>(At + i * 32 * 128 + oh1);
            f.a2[i] = *reinterpret_cast<const f16x8*>(At + i * 32 * 128 + oh2);
            f.b1[i] = *reinterpret_cast<const f16x8*>(Bt + i * 32 * 128 + oh1);
            f.b2[i] = *reinterpret_cast<const f16x8*>(Bt + i * 32 * 128 + oh2);
        }
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto mma = [&](const H2Frag& f, int i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a2[i], f.b1[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a1[i], f.b2[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a1[i], f.b1[j], acc[i][j], 0, 0, 0);
        }
    };

    H2Frag F0, F1;
    stageA(0, 0); stageB(0, 0);
    __syncthreads();                                         // tile 0 landed (vmcnt(0)) and published
    stageA(1, 1);
    loadF(F0, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        // ---- phase 1: k-step 0 of tile kt on F0; fetch k-step 1 into F1; B half of tile kt+1 starts its flight
        stageB(kt + 1, cur ^ 1);
        loadF(F1, cur, 1);
        mma(F0, 0); mma(F0, 1); mma(F0, 2); mma(F0, 3);
        // ---- phase 2a: first half of k-step 1
        mma(F1, 0); mma(F1, 1);
#pragma unroll
        for (int q = 0; q < 8; ++q) {                        // 24 MFMA with the 8 B-DMAs and the first 8 reads
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {                        // 24 MFMA with the other 8 reads
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);  // phase 2a
        // tile kt+1 (A issued one iteration ago, B in phase 1) has landed; every read of tile kt's buffer is done
        __syncthreads();
        // ---- phase 2b: second half of k-step 1 covers the first fragment reads of tile kt+1 and the next A-DMAs
        stageA(kt + 2, cur);
        loadF(F0, cur ^ 1, 0);
        mma(F1, 2); mma(F1, 3);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 1);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // clamped tail DMAs must land before LDS is reused
    __syncthreads();

    // ---- epilogue: 64 rows x 64 channels at a time through this wave's 16 KB LDS slice (see oz_net_h2.h)
    if (!g.out_h2) {
        float* o = reinterpret_cast<float*>(out);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = nt * BN + wn * 128 + j * 32 + r32;
            const float sc = scale[col], sh = shift[col];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long m = (long long)mt * BM + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    float v = fmaf(acc[i][j][r], sc, sh);
                    if (g.relu) v = fmaxf(v, 0.f);
                    if (m < M) o[(size_t)m * g.N + col] = v;
                }
        }
        return;
    }
    bool over = false;
    _Float16* slice = reinterpret_cast<_Float16*>(smem + wave * 16384);
    uint4* o = reinterpret_cast<uint4*>(out);
    const int nq = g.N >> 2;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = hh * 2 + ii;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = jh * 2 + jj;
                    const int lc = jj * 32 + r32;            // column inside the 64-channel pass
                    const int col = nt * BN + wn * 128 + jh * 64 + lc;
                    const float sc = scale[col], sh = shift[col];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int lr = ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        float v = fmaf(acc[i][j][r], sc, sh);
                        if (g.relu) v = fmaxf(v, 0.f);
                        over |= fabsf(v) > H2_F16_MAX;
                        _Float16 h1, h2;
                        h2_split(v, h1, h2);
                        _Float16* p = slice + lr * 128 + (lc >> 3) * 16 + (lc & 7);
                        p[0] = h1; p[8] = h2;
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int q = c * 64 + lane, lr = q >> 4, cq = q & 15;
                const long long m = (long long)mt * BM + wm * 128 + hh * 64 + lr;
                const uint4 val = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(slice) + lr * 256 + cq * 16);
                if (m < M) o[(size_t)m * nq + ((nt * BN + wn * 128 + jh * 64) >> 2) + cq] = val;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
        }
    if (over) atomicOr(flag, 1);
}

// ------------------------------------------------------------------------------------------------------------
// k_gemm_h2_s16: the 8-wave two-stage kernel of oz_net_h2.h on v_mfma_f32_16x16x32_f16 instead of 32x32x16.
// Same FLOP per cycle, same LDS bytes; under load the chip holds a higher clock on this shape
// (MI355X_MICROARCH.md, DVFS give-back item 7), and all three kernel structures above are clock/power bound
// (measured 1.77-1.89 GHz, 66-70 % matrix-pipe busy).  One MFMA consumes the whole 32-deep k-tile:
// lane l holds row/col l&15 and k-group l>>4 (8 consecutive k), i.e. chunks 2*(l>>4) + plane of the 128-byte row.
template <typename CF>
__global__ __launch_bounds__(CF::NT, 2) void k_gemm_h2_s16(const uint4* __restrict__ in, const uint4* __restrict__ Wh,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           void* __restrict__ out, const int* __restrict__ d_count, H2Geom g,
                                                           int num_mt, const uint4* __restrict__ zero_line, int* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = CF::BM, BN = CF::BN, IA = CF::IA, IB = CF::IB;
    constexpr int RI = CF::TI * 2, RJ = CF::TJ * 2;           // 16-row / 16-column blocks per wave
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const int nnt = g.N / BN;
    const int id = blockIdx.x, xcd = id & 7, jb = id >> 3;
    const int mt = (jb / nnt) * 8 + xcd, nt = jb % nnt;
    const int P = g.Hout * g.Hout;
    const long long M = (long long)(*d_count) * P;
    if (mt >= num_mt || (long long)mt * BM >= M) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / CF::WN, wn = wave % CF::WN;
    const int rowq = g.Cin >> 2, wrowq = g.K >> 2;

    long long aidx[IA];
    unsigned amask[IA];
    unsigned bidx[IB];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int row = (wave * IA + i) * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        const long long m = (long long)mt * BM + row;
        aidx[i] = 0; amask[i] = 0;
        if (m < M) {
            const int b = (int)(m / P), pix = (int)(m % P), oy = pix / g.Hout, ox = pix % g.Hout;
            aidx[i] = (((long long)b * g.Hin + (oy - g.pad)) * g.Hin + (ox - g.pad)) * rowq + lc;
            unsigned mk = 0;
            for (int t = 0; t < g.taps; ++t) {
                const int iy = oy - g.pad + t / 3, ix = ox - g.pad + t % 3;
                if (iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Hin) mk |= 1u << t;
            }
            amask[i] = mk;
        }
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
        const int row = (wave * IB + i) * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        bidx[i] = (unsigned)(nt * BN + row) * (unsigned)wrowq + (unsigned)lc;
    }
    const uint4* zsrc = zero_line + (lane & 7);
    const int nk = g.K / H2_BK;

    auto stage = [&](int kt_raw, int buf) {
        const int kt = kt_raw < nk ? kt_raw : nk - 1;
        const int k0 = kt * H2_BK, tap = k0 / g.Cin, ci0 = k0 - tap * g.Cin;
        const long long toff = ((long long)(tap / 3) * g.Hin + (tap % 3)) * rowq + (ci0 >> 2);
        unsigned char* la = smem + (size_t)buf * CF::BUF + wave * IA * 1024;
        unsigned char* lb = smem + (size_t)buf * CF::BUF + CF::TILEA + wave * IB * 1024;
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const uint4* ga = ((amask[i] >> tap) & 1) ? in + (aidx[i] + toff) : zsrc;
            __builtin_amdgcn_global_load_lds((h2_gptr)ga, (h2_lptr)(la + i * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            const uint4* gb = Wh + bidx[i] + (k0 >> 2);
            __builtin_amdgcn_global_load_lds((h2_gptr)gb, (h2_lptr)(lb + i * 1024), 16, 0, 0);
        }
    };

    f32x4v acc[RI][RJ];
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
        for (int j = 0; j < RJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    const int r16 = lane & 15, kg = lane >> 4;
    const int swz = (r16 >> 1) & 7;
    const int oh1 = ((2 * kg) ^ swz) * 16, oh2 = ((2 * kg + 1) ^ swz) * 16;

    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        stage(kt + 1, buf ^ 1);
        const unsigned char* At = smem + (size_t)buf * CF::BUF + (wm * RI * 16 + r16) * 128;
        const unsigned char* Bt = smem + (size_t)buf * CF::BUF + CF::TILEA + (wn * RJ * 16 + r16) * 128;
        f16x8 b1[RJ], b2[RJ];
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            b1[j] = *reinterpret_cast<const f16x8*>(Bt + j * 16 * 128 + oh1);
            b2[j] = *reinterpret_cast<const f16x8*>(Bt + j * 16 * 128 + oh2);
        }
#pragma unroll
        for (int i = 0; i < RI; ++i) {
            const f16x8 a1 = *reinterpret_cast<const f16x8*>(At + i * 16 * 128 + oh1);
            const f16x8 a2 = *reinterpret_cast<const f16x8*>(At + i * 16 * 128 + oh2);
#pragma unroll
            for (int j = 0; j < RJ; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1[j], acc[i][j], 0, 0, 0);
            }
        }
        {
            constexpr int NDMA = IA + IB, NMF = RI * RJ * 3, PER = NMF / (2 * NDMA);
#pragma unroll
            for (int q = 0; q < NDMA; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue.  C/D layout of 16x16: col = lane&15, row = (lane>>4)*4 + reg
    if (!g.out_h2) {
        float* o = reinterpret_cast<float*>(out);
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            const int col = nt * BN + wn * RJ * 16 + j * 16 + r16;
            const float sc = scale[col], sh = shift[col];
#pragma unroll
            for (int i = 0; i < RI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long m = (long long)mt * BM + wm * RI * 16 + i * 16 + kg * 4 + r;
                    float v = fmaf(acc[i][j][r], sc, sh);
                    if (g.relu) v = fmaxf(v, 0.f);
                    if (m < M) o[(size_t)m * g.N + col] = v;
                }
        }
        return;
    }
    static_assert(RJ == 4, "the h2 epilogue assumes a 64-channel wave tile");
    bool over = false;
    _Float16* slice = reinterpret_cast<_Float16*>(smem + wave * 16384);
    uint4* o = reinterpret_cast<uint4*>(out);
    const int nq = g.N >> 2;
#pragma unroll
    for (int hh = 0; hh < RI / 4; ++hh) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = hh * 4 + ii;
#pragma unroll
            for (int j = 0; j < RJ; ++j) {
                const int lc = j * 16 + r16;
                const int col = nt * BN + wn * RJ * 16 + lc;
                const float sc = scale[col], sh = shift[col];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lr = ii * 16 + kg * 4 + r;
                    float v = fmaf(acc[i][j][r], sc, sh);
                    if (g.relu) v = fmaxf(v, 0.f);
                    over |= fabsf(v) > H2_F16_MAX;
                    _Float16 h1, h2;
                    h2_split(v, h1, h2);
                    _Float16* p = slice + lr * 128 + (lc >> 3) * 16 + (lc & 7);
                    p[0] = h1; p[8] = h2;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int q = c * 64 + lane, lr = q >> 4, cq = q & 15;
            const long long m = (long long)mt * BM + wm * RI * 16 + hh * 64 + lr;
            const uint4 val = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(slice) + lr * 256 + cq * 16);
            if (m < M) o[(size_t)m * nq + ((nt * BN + wn * RJ * 16) >> 2) + cq] = val;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
    if (over) atomicOr(flag, 1);
}
