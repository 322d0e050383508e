// oz_train_fused.h -- small-batch BatchNormalization kernels of the training step (included by oz_train.hip).
//
// At the reference's batch size (32 boards: 1152-2048 rows per conv layer, 32 per dense layer) the five launches of the
// general BN forward path (sum, mean, centred sum of squares, rstd + moving statistics, apply) and the five of the
// backward path cost ~5 us each for a few microseconds of work.  Here one 1024-thread block owns 16 channels for ALL rows
// and does the whole layer in one launch: 64 row lanes x 16 columns (C / 16 blocks), every thread walks its rows in order (loads unrolled for
// memory-level parallelism), the 64 partial sums are combined in a fixed order through LDS (bit-reproducible), the activation stays L2-resident between passes.
// Used when rows <= OZ_BN_FUSED_MAX_ROWS; larger batches keep the multi-block reductions (bandwidth-bound there).
#pragma once

#define OZ_BN_FUSED_MAX_ROWS 4096        // (64 boards of 8x8; above that the streaming multi-block reductions win: 32 blocks cannot feed HBM)

#define OZ_BN_RL 64                        // row lanes per block
#define OZ_BN_COLS 16                      // channels per block
__device__ __forceinline__ float t_block_sum4(float v, float (*sh)[OZ_BN_COLS], int lane4, int c64) {
    __syncthreads();                       // sh may still be read from the previous reduction
    sh[lane4][c64] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < OZ_BN_RL; ++i) s += sh[i][c64];          // fixed order
    return s;
}

// a = relu((z - mean) * rstd * gamma + beta) [* keep / (1 - rate)]; also mean, rstd and the staged moving statistics
template <int RMAX>                        // rows per thread the launch is built for: 32 (up to 2048 rows: no spill at 128 registers) or 64
__global__ __launch_bounds__(1024) void k_t_bn_fwd_fused(const float* __restrict__ z, float* __restrict__ a, const int* __restrict__ d_count,
                                                        int P, int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                        const float* __restrict__ mm, const float* __restrict__ mv,
                                                        float* __restrict__ mm_new, float* __restrict__ mv_new, float mom, int fused_var,
                                                        float rate, uint64_t seed, uint64_t step, int dlayer) {
    __shared__ float sh[OZ_BN_RL][OZ_BN_COLS];
    const int c64 = threadIdx.x & (OZ_BN_COLS - 1), lane4 = threadIdx.x / OZ_BN_COLS, c = blockIdx.x * OZ_BN_COLS + c64;
    const long long M = (long long)(*d_count) * P;
    // a thread's rows (at most OZ_BN_FUSED_MAX_ROWS / OZ_BN_RL = 64) are fetched ONCE, all loads in flight, and stay in registers for the three
    // passes (sum, centred squares, apply): one round trip to z instead of 3 x rows / 4 of them -- 38 us for the 2048 rows of conv1 at the
    // reference's batch, round 5 (rocprofv3 timeline of a step, tools/trace_timeline.py).  The sums are formed in the same row order as before.
    static_assert(RMAX * OZ_BN_RL <= OZ_BN_FUSED_MAX_ROWS, "rows per thread");
    float zr[RMAX];
    const float g = gamma[c], b = beta[c];
    // (buffer loads: one per-thread offset register + a scalar row-block offset per load; rows beyond M fall outside the descriptor's range and read 0)
    const auto zrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(z), 0, (int)(M * C * 4), 0x00020000);
    const int zoff = (lane4 * C + c) * 4;
#pragma unroll
    for (int k = 0; k < RMAX; ++k) zr[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(zrs, zoff, k * OZ_BN_RL * C * 4, 0));
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < RMAX; ++k) if (lane4 + (long long)k * OZ_BN_RL < M) s += zr[k];
    const float mean = t_block_sum4(s, sh, lane4, c64) / (float)M;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < RMAX; ++k) if (lane4 + (long long)k * OZ_BN_RL < M) { const float d = zr[k] - mean; q = fmaf(d, d, q); }
    const float var = t_block_sum4(q, sh, lane4, c64) / (float)M;
    const float rstd = 1.0f / sqrtf(var + 1e-3f);
    if (lane4 == 0) {
        mean_out[c] = mean; rstd_out[c] = rstd;
        const float uv = fused_var ? var * ((float)M / (float)(M > 1 ? M - 1 : 1)) : var;
        mm_new[c] = mm[c] * mom + mean * (1.0f - mom);
        mv_new[c] = mv[c] * mom + uv * (1.0f - mom);
    }
#pragma unroll
    for (int k = 0; k < RMAX; ++k) {
        const long long m = lane4 + (long long)k * OZ_BN_RL;
        if (m < M) {
            const size_t i = (size_t)m * C + c;
            float y = (zr[k] - mean) * rstd * g + b;
            y = y > 0.f ? y : 0.f;
            if (rate > 0.f) y = oz_dropout_keep(seed, step, (uint64_t)dlayer, (uint64_t)i, rate) ? y / (1.0f - rate) : 0.f;
            a[i] = y;
        }
    }
}

// dy = (a > 0 ? dA * post_scale : 0); dgamma = sum dy * xhat, dbeta = sum dy;
// dz = gamma * rstd * (dy - dbeta / M - xhat * dgamma / M), written at (b, oy + zoff, ox + zoff) of an Hz x Hz buffer;
// dbias = sum dz (rounding noise behind a training-mode BN, computed like autograd would)
__global__ __launch_bounds__(1024) void k_t_bn_bwd_fused(const float* __restrict__ dA, const float* __restrict__ a, const float* __restrict__ z,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, float post_scale, const int* __restrict__ d_count,
                                                        int Hout, int C, int Hz, int zoff, float* __restrict__ dz,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias,
                                                        unsigned* __restrict__ dzmax /* nullable: atomic max of |dz| (bit pattern), f16x2 data gradient */) {
    __shared__ float sh[OZ_BN_RL][OZ_BN_COLS];
    const int c64 = threadIdx.x & (OZ_BN_COLS - 1), lane4 = threadIdx.x / OZ_BN_COLS, c = blockIdx.x * OZ_BN_COLS + c64;
    const int P = Hout * Hout;
    const long long M = (long long)(*d_count) * P;
    const float mu = mean[c], rs = rstd[c];
    float s0 = 0.f, s1 = 0.f;
    _Pragma("unroll 4") for (long long m = lane4; m < M; m += OZ_BN_RL) {
        const size_t i = (size_t)m * C + c;
        const float dy = a[i] > 0.f ? dA[i] * post_scale : 0.f;
        s0 += dy;
        s1 = fmaf(dy, (z[i] - mu) * rs, s1);
    }
    const float S0 = t_block_sum4(s0, sh, lane4, c64);
    const float S1 = t_block_sum4(s1, sh, lane4, c64);
    const float inv = 1.0f / (float)M, gr = gamma[c] * rs;
    float sb = 0.f, amax = 0.f;
    _Pragma("unroll 4") for (long long m = lane4; m < M; m += OZ_BN_RL) {
        const size_t i = (size_t)m * C + c;
        const float dy = a[i] > 0.f ? dA[i] * post_scale : 0.f;
        const float xh = (z[i] - mu) * rs;
        const float g = gr * (dy - S0 * inv - xh * S1 * inv);
        const int b = (int)(m / P), pix = (int)(m % P);
        dz[(((size_t)b * Hz + pix / Hout + zoff) * Hz + pix % Hout + zoff) * C + c] = g;
        sb += g;
        amax = fmaxf(amax, fabsf(g));
    }
    if (dzmax) {                                             // one atomic per block (see k_t_bnb_apply)
        __shared__ float wm_s[16];
        for (int o = 32; o; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        if ((threadIdx.x & 63) == 0) wm_s[threadIdx.x >> 6] = amax;
        __syncthreads();
        if (threadIdx.x == 0) {
            float mx = wm_s[0];
            for (int k = 1; k < 16; ++k) mx = fmaxf(mx, wm_s[k]);
            atomicMax(dzmax, __float_as_uint(mx));
        }
    }
    const float SB = t_block_sum4(sb, sh, lane4, c64);
    if (lane4 == 0) { dgamma[c] = S1; dbeta[c] = S0; dbias[c] = SB; }
}

// ---- more than OZ_BNB_MIN_ROWS rows (from the reference's 32 boards on the 8x8 conv layers, 1152-2048 rows, upwards): one block per 16 channels leaves
// 224 CUs idle and every thread walks ~100 dependent rows (35-64 us per layer, measured); here the rows are split over
// the grid instead.  Launch 1 = k_t_colreduce<2> over RS row splits (partial sums of dy and dy * xhat), launch 2 = this
// kernel: every block finishes the two sums from the RS partials (fixed order; 2 x RS independent 16-byte loads per thread),
// writes dz for its rows and leaves the column sums of its dz rows in part2[split]; the bias gradient (sum of part2 over the
// splits, k_t_sum_partials) is not on the data-gradient chain and is finished on the second stream.
#define OZ_BNB_MIN_ROWS 256                // at most this many rows: the one-launch kernel above (a handful of rows per thread there)
#define OZ_BNB_MAX_ROWS (1LL << 40)      // (no upper bound: at large batch the same two launches replace five and one pass over dz)
#define OZ_BNB_MAX_RB 256
__global__ __launch_bounds__(256) void k_t_bnb_apply(const float* __restrict__ dA, const float* __restrict__ a, const float* __restrict__ z,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     float post_scale, const int* __restrict__ d_count, int Hout, int C, int Hz, int zoff,
                                                     const float* __restrict__ partial /*[RS][2][C]*/, int RS, float* __restrict__ dz,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ part2 /*[RS][C]*/,
                                                     unsigned* __restrict__ dzmax /* nullable: atomic max of |dz| (bit pattern) for the f16x2 data gradient */) {
    const int Q = C / 4, lpr = Q < 64 ? Q : 64, rpp = 256 / lpr;
    const int q = blockIdx.x * 64 + (int)(threadIdx.x % lpr), rsub = threadIdx.x / lpr, sp = blockIdx.y;
    const bool valid = q < Q;
    const int c = valid ? q * 4 : 0;
    const int P = Hout * Hout;
    const long long M = valid ? (long long)(*d_count) * P : 0;
    // the two sums: row lane r adds the partials of splits r, r + rpp, ... in increasing order, the rpp lane sums are combined in lane order
    // (fixed order; RS / rpp x 2 independent 16-byte loads per thread instead of RS x 2)
    __shared__ f32x4 sh[2][256];
    f32x4 S0 = {0.f, 0.f, 0.f, 0.f}, S1 = S0;
    _Pragma("unroll 8") for (int s = rsub; s < RS; s += rpp) {
        S0 += *reinterpret_cast<const f32x4*>(partial + ((size_t)s * 2 + 0) * C + c);
        S1 += *reinterpret_cast<const f32x4*>(partial + ((size_t)s * 2 + 1) * C + c);
    }
    sh[0][threadIdx.x] = S0; sh[1][threadIdx.x] = S1;
    __syncthreads();
    {
        const int base = threadIdx.x % lpr;
        S0 = sh[0][base]; S1 = sh[1][base];
        for (int k = 1; k < rpp; ++k) { S0 += sh[0][base + k * lpr]; S1 += sh[1][base + k * lpr]; }
    }
    __syncthreads();
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c),
                ga = *reinterpret_cast<const f32x4*>(gamma + c);
    const float inv = 1.0f / (float)((long long)(*d_count) * P);
    f32x4 sb = {0.f, 0.f, 0.f, 0.f};
    float amax = 0.f;
    _Pragma("unroll 4") for (long long m = (long long)sp * rpp + rsub; m < M; m += (long long)RS * rpp) {          // 12 independent 16-byte loads in flight per thread; the adds stay in row order
        const size_t i = (size_t)m * C + c;
        const f32x4 av = *reinterpret_cast<const f32x4*>(a + i), xv = *reinterpret_cast<const f32x4*>(dA + i), zv = *reinterpret_cast<const f32x4*>(z + i);
        f32x4 g;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dy = av[k] > 0.f ? xv[k] * post_scale : 0.f;
            const float xh = (zv[k] - mu[k]) * rs[k];
            g[k] = ga[k] * rs[k] * (dy - S0[k] * inv - xh * S1[k] * inv);
            sb[k] += g[k];
            amax = fmaxf(amax, fabsf(g[k]));
        }
        const int b = (int)(m / P), pix = (int)(m % P);
        *reinterpret_cast<f32x4*>(dz + (((size_t)b * Hz + pix / Hout + zoff) * Hz + pix % Hout + zoff) * C + c) = g;
    }
    // the block's largest |dz|: ONE atomic per block -- an atomic on one word costs ~12 ns each, and with one per wave the 512 .. 2048 of them
    // were most of this launch at the reference's batch (round 5)
    __shared__ float wm_s[4];
    if (dzmax) {
        for (int o = 32; o; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        if ((threadIdx.x & 63) == 0) wm_s[threadIdx.x >> 6] = amax;
    }
    sh[0][threadIdx.x] = sb;
    __syncthreads();
    if (dzmax && threadIdx.x == 0) atomicMax(dzmax, __float_as_uint(fmaxf(fmaxf(wm_s[0], wm_s[1]), fmaxf(wm_s[2], wm_s[3]))));
    if (rsub == 0 && valid) {
        f32x4 t = sh[0][threadIdx.x];
        for (int k = 1; k < rpp; ++k) t += sh[0][threadIdx.x + k * lpr];
        *reinterpret_cast<f32x4*>(part2 + (size_t)sp * C + c) = t;
        if (sp == 0) {
            *reinterpret_cast<f32x4*>(dgamma + c) = S1;
            *reinterpret_cast<f32x4*>(dbeta + c) = S0;
        }
    }
}
