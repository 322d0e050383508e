// oz_train.hip -- one optimiser step of OthelloNN / BaseNN on gfx950.
//
// Replaces NNetWrapper.train (Net/NNet.py:53-68) = keras Model.fit on the graph of Net/OthelloNN.py:42-56 compiled
// with loss=['categorical_crossentropy','mean_squared_error'], Adam(lr, clipvalue=0.5) (BaseNN.py:57: no clipvalue).
// Semantics (restated in oracle/train_ref.py, which is what the parity tests compare with):
//   * BatchNormalization in training mode: batch mean / BIASED batch variance, eps 1e-3, momentum 0.99; the 4-D conv BNs
//     run Keras' fused kernel whose moving-variance update uses the UNBIASED variance, the dense BNs the biased one.
//   * the policy loss is Keras' probability-path categorical cross-entropy applied to the RESHAPED (B, n, n) output:
//     every board row is renormalised, clipped to [1e-7, 1-1e-7], -sum(t log q) per row, mean over batch x rows.
//   * value loss: mean squared error; total = pi + v.   * Dropout: inverted, counter-based keep mask.
//   * Adam as tf.keras: lr_t = lr sqrt(1-b2^t)/(1-b1^t); var -= lr_t m / (sqrt(v) + 1e-7); clipvalue clips g first.
//
// All contractions run on the fp32 matrix cores: forward and data-gradient passes reuse k_gemm_f32 (oz_net.hip; the
// data gradient of a 3x3 convolution is a 3x3 convolution with the taps reversed and the channel roles swapped, 'valid'
// layers through a zero-bordered buffer), the weight gradient is k_wgrad_f32 below (a "TN" implicit GEMM whose k
// index is the batch x pixel row).  Column reductions (BN statistics, BN backward sums, bias gradients) use a
// fixed-order two-stage reduction, so a step is deterministic.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "oz_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BN_EPS 1e-3f
#define RED_S 256         // row splits of the column reductions (2048 waves of 1 KB loads at 512 channels: the passes are HBM streams)

// ---------------------------------------------------------------- dropout hash (same formula in oracle/train_ref.py)
OZ_HD bool oz_dropout_keep(uint64_t seed, uint64_t step, uint64_t layer, uint64_t idx, float rate) {
    const uint64_t a = oz_sm64(seed + 0x632BE59BD9B4E019ULL * step);
    const uint64_t h = oz_sm64(a ^ ((layer + 1) * 0xD1B54A32D192ED03ULL) ^ (idx * 0x9E3779B97F4A7C15ULL));
    const float u = (float)(h >> 40) * (1.0f / 16777216.0f);
    return u >= rate;
}

#include "oz_train_fused.h"

// ---------------------------------------------------------------- conv1 forward (raw, + bias) and its weight gradient
// x[b][iy][ix][ch]: cin = 2 -> (own bit, opp bit); cin = 1 -> own bit - opp bit (Net/BaseNN.py:41-44)
__device__ __forceinline__ float t_plane(uint64_t o, uint64_t p, int sq, int ch, int cin) {
    const float a = (float)((o >> sq) & 1), b = (float)((p >> sq) & 1);
    return cin == 2 ? (ch == 0 ? a : b) : a - b;
}

// one thread = one pixel row m x 4 consecutive output channels (16-byte weight loads and stores)
__global__ __launch_bounds__(256) void k_t_conv1_fwd(const uint64_t* __restrict__ own, const uint64_t* __restrict__ opp,
                                                     const int* __restrict__ d_count, int n, int C, int cin,
                                                     const float* __restrict__ W /*[9][cin][C]*/, const float* __restrict__ bias,
                                                     float* __restrict__ z /*[B][n*n][C]*/) {
    const int P = n * n, Q = C / 4;
    const long long M = (long long)(*d_count) * P;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long m = idx / Q;
    if (m >= M) return;
    const int co = (int)(idx % Q) * 4;
    const int b = (int)(m / P), pix = (int)(m % P), y = pix / n, x = pix % n;
    const uint64_t o = own[b], p = opp[b];
    // all 9 * cin weight vectors of this thread's four channels in flight at once (the tap loop with its border test was a chain of 18 dependent
    // L2 round trips: 33 us per launch at the reference's batch, round 5); the products are added in the same tap / plane order, a tap outside
    // the board is skipped as before
    f32x4 w[18];                                            // [tap][plane]: static register indices whatever cin is
#pragma unroll
    for (int i = 0; i < 18; ++i) if ((i & 1) < cin) w[i] = *reinterpret_cast<const f32x4*>(W + (size_t)((i >> 1) * cin + (i & 1)) * C + co);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + co);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
        if (iy < 0 || iy >= n || ix < 0 || ix >= n) continue;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            if (ch >= cin) break;
            const float xv = t_plane(o, p, iy * 8 + ix, ch, cin);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = fmaf(xv, w[2 * t + ch][k], acc[k]);
        }
    }
    *reinterpret_cast<f32x4*>(z + (size_t)m * C + co) = acc + bv;
}

// dW1[t][ch][co] = sum_m x[m shifted by t][ch] * dz[m][co]: ONE pass over dz -- a thread owns an output channel and keeps the
// 9 * cin sums of its row split in registers (the input planes are bits of the two bitboards, wave-uniform per row)
template <int NB>                          // board size as a constant: the row -> (board, pixel) split is a shift / a multiply, not a 64-bit division per row
__global__ __launch_bounds__(256) void k_t_conv1_wgrad(const uint64_t* __restrict__ own, const uint64_t* __restrict__ opp,
                                                       const int* __restrict__ d_count, int C, int cin,
                                                       const float* __restrict__ dz, float* __restrict__ partial /*[S][9*cin][C]*/, int S) {
    constexpr int n = NB, P = NB * NB;
    const int co = blockIdx.x * 256 + threadIdx.x, sp = blockIdx.y;
    if (co >= C) return;
    const int M = (*d_count) * P;
    float acc[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) acc[i] = 0.f;
    // four rows of the split per trip: their loads (a dz element and two bitboards each) are issued together, the sums stay in row order
    for (int m0 = sp; m0 < M; m0 += 4 * S) {
        float d[4];
        uint64_t o[4], p[4];
        int pixs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = m0 + u * S;
            const bool in = m < M;
            const int mm = in ? m : m0;
            const int b = mm / P;
            pixs[u] = in ? mm % P : -1;
            o[u] = own[b]; p[u] = opp[b];
            d[u] = dz[(size_t)mm * C + co];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (pixs[u] < 0) continue;
            const int y0 = pixs[u] / n, x0 = pixs[u] % n;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int y = y0 + t / 3 - 1, x = x0 + t % 3 - 1;
                if (y < 0 || y >= n || x < 0 || x >= n) continue;
                const int sq = y * 8 + x;
                const float a = (float)((o[u] >> sq) & 1), bb = (float)((p[u] >> sq) & 1);
                if (cin == 2) { acc[2 * t] = fmaf(a, d[u], acc[2 * t]); acc[2 * t + 1] = fmaf(bb, d[u], acc[2 * t + 1]); }
                else acc[t] = fmaf(a - bb, d[u], acc[t]);
            }
        }
    }
    for (int tc = 0; tc < 9 * cin; ++tc) partial[((size_t)sp * 9 * cin + tc) * C + co] = acc[tc];
}

// out[i] = sum_s partial[s][i] in fixed order
__global__ void k_t_sum_partials(const float* __restrict__ partial, int S, long long count, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float a = 0.f;
    _Pragma("unroll 8") for (int s = 0; s < S; ++s) a += partial[(size_t)s * count + i];     // the loads are independent: 8 in flight, the adds stay in order
    out[i] = a;
}

// ---------------------------------------------------------------- column reductions over the rows of [M][C] matrices
// MODE 0: sum x            MODE 1: sum (x - mean)^2
// MODE 2: BN backward sums: dy = (a > 0 ? dA * post_scale : 0); s0 = sum dy, s1 = sum dy * xhat, xhat = (z - mean) * rstd
// MODE 3: sum x with x stored in a zero-bordered buffer (rows (b, oy, ox) of an Hout^2 block at offset zoff in Hz^2)
struct RedArgs {
    const float *x, *a, *z, *mean, *rstd;
    float post_scale;
    int P, C, Hout, Hz, zoff;
};
// Streaming form: a thread owns 4 consecutive channels (16-byte loads), the lanes of a row cover up to 256 channels
// (1 KB per wave instruction), the 256 threads of a block cover 4 rows (8 at 128 channels) per pass, rows strided RED_S
// passes apart; the row groups of a block are combined through LDS in fixed order.
template <int MODE>
__global__ __launch_bounds__(256) void k_t_colreduce(RedArgs r, const int* __restrict__ d_count, float* __restrict__ partial /*[RED_S][2][C]*/) {
    const int Q = r.C / 4;                                   // float4 columns
    const int lpr = Q < 64 ? Q : 64, rpp = 256 / lpr;        // lanes per row, rows per pass
    const int q = blockIdx.x * 64 + (int)(threadIdx.x % lpr), rsub = threadIdx.x / lpr, sp = blockIdx.y;
    const bool valid = q < Q;                                // (channel counts that are not a multiple of 256: the last block is partly idle)
    const int c = valid ? q * 4 : 0;
    const long long M = valid ? (long long)(*d_count) * r.P : 0;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    f32x4 mu = s0, rs = s0;
    if (MODE == 1 || MODE == 2) mu = *reinterpret_cast<const f32x4*>(r.mean + c);
    if (MODE == 2) rs = *reinterpret_cast<const f32x4*>(r.rstd + c);
    for (long long m = (long long)sp * rpp + rsub; m < M; m += (long long)gridDim.y * rpp) {      // gridDim.y row splits (RED_S for the streaming passes)
        const size_t o = (size_t)m * r.C + c;
        if (MODE == 0) s0 += *reinterpret_cast<const f32x4*>(r.x + o);
        if (MODE == 1) {
            const f32x4 d = *reinterpret_cast<const f32x4*>(r.x + o) - mu;
#pragma unroll
            for (int k = 0; k < 4; ++k) s0[k] = fmaf(d[k], d[k], s0[k]);
        }
        if (MODE == 2) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(r.a + o), xv = *reinterpret_cast<const f32x4*>(r.x + o),
                        zv = *reinterpret_cast<const f32x4*>(r.z + o);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dy = av[k] > 0.f ? xv[k] * r.post_scale : 0.f;
                s0[k] += dy;
                s1[k] = fmaf(dy, (zv[k] - mu[k]) * rs[k], s1[k]);
            }
        }
        if (MODE == 3) {
            const int HH = r.Hout * r.Hout, b = (int)(m / HH), pix = (int)(m % HH);
            s0 += *reinterpret_cast<const f32x4*>(r.x + (((size_t)b * r.Hz + pix / r.Hout + r.zoff) * r.Hz + pix % r.Hout + r.zoff) * r.C + c);
        }
    }
    __shared__ f32x4 sh[2][256];
    sh[0][threadIdx.x] = s0;
    sh[1][threadIdx.x] = s1;
    __syncthreads();
    if (rsub == 0 && valid) {
        f32x4 a0 = sh[0][threadIdx.x], a1 = sh[1][threadIdx.x];
        for (int k = 1; k < rpp; ++k) { a0 += sh[0][threadIdx.x + k * lpr]; a1 += sh[1][threadIdx.x + k * lpr]; }
        *reinterpret_cast<f32x4*>(partial + ((size_t)sp * 2 + 0) * r.C + c) = a0;
        *reinterpret_cast<f32x4*>(partial + ((size_t)sp * 2 + 1) * r.C + c) = a1;
    }
}
__device__ __forceinline__ float t_sum_s(const float* partial, int which, int C, int c) {
    float a = 0.f;
    for (int s = 0; s < RED_S; ++s) a += partial[((size_t)s * 2 + which) * C + c];
    return a;
}
// mean[c] = sum / M
__global__ void k_t_fin_mean(const float* __restrict__ partial, const int* __restrict__ d_count, int P, int C, float* __restrict__ mean) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) mean[c] = t_sum_s(partial, 0, C, c) / (float)((long long)(*d_count) * P);
}
// rstd[c] = 1/sqrt(var + eps); staged moving statistics: new = old * mom + batch * (1 - mom) (var unbiased when fused)
__global__ void k_t_fin_var(const float* __restrict__ partial, const int* __restrict__ d_count, int P, int C, const float* __restrict__ mean,
                            float* __restrict__ rstd, const float* __restrict__ mm, const float* __restrict__ mv,
                            float* __restrict__ mm_new, float* __restrict__ mv_new, float mom, int fused) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const long long M = (long long)(*d_count) * P;
    const float var = t_sum_s(partial, 0, C, c) / (float)M;
    rstd[c] = 1.0f / sqrtf(var + BN_EPS);
    const float uv = fused ? var * ((float)M / (float)(M > 1 ? M - 1 : 1)) : var;
    mm_new[c] = mm[c] * mom + mean[c] * (1.0f - mom);
    mv_new[c] = mv[c] * mom + uv * (1.0f - mom);
}
// out[c] = sum over the rows (fixed order over the RED_S partials)
__global__ void k_t_fin_colsum(const float* __restrict__ partial, int C, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) out[c] = t_sum_s(partial, 0, C, c);
}
// a = relu((z - mean) * rstd * gamma + beta) [* keep / (1 - rate)]; 4 consecutive channels per thread
__global__ __launch_bounds__(256) void k_t_bn_fwd(const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ a,
                                                  const int* __restrict__ d_count, int P, int C, float rate, uint64_t seed, uint64_t step, int dlayer) {
    const long long total = (long long)(*d_count) * P * C;
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= total) return;
    const int c = (int)(i % C);
    const f32x4 zv = *reinterpret_cast<const f32x4*>(z + i), mu = *reinterpret_cast<const f32x4*>(mean + c),
                rs = *reinterpret_cast<const f32x4*>(rstd + c), ga = *reinterpret_cast<const f32x4*>(gamma + c),
                be = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 out;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float y = (zv[k] - mu[k]) * rs[k] * ga[k] + be[k];
        y = y > 0.f ? y : 0.f;
        if (rate > 0.f) y = oz_dropout_keep(seed, step, (uint64_t)dlayer, (uint64_t)(i + k), rate) ? y / (1.0f - rate) : 0.f;
        out[k] = y;
    }
    *reinterpret_cast<f32x4*>(a + i) = out;
}
// dgamma = s1, dbeta = s0
__global__ void k_t_fin_bnbwd(const float* __restrict__ partial, int C, float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ sums /*[2][C]*/) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s0 = t_sum_s(partial, 0, C, c), s1 = t_sum_s(partial, 1, C, c);
    dbeta[c] = s0; dgamma[c] = s1; sums[c] = s0; sums[C + c] = s1;
}
// dz = gamma * rstd * (dy - s0/M - xhat * s1/M), written at (b, oy+zoff, ox+zoff) of an Hz x Hz buffer; 4 channels per thread
__global__ __launch_bounds__(256) void k_t_bn_bwd(const float* __restrict__ dA, const float* __restrict__ a, const float* __restrict__ z,
                                                  const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                  const float* __restrict__ sums, float post_scale, const int* __restrict__ d_count,
                                                  int Hout, int C, int Hz, int zoff, float* __restrict__ dz) {
    const int P = Hout * Hout;
    const long long M = (long long)(*d_count) * P, total = M * C;
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= total) return;
    const int c = (int)(i % C);
    const long long m = i / C;
    const f32x4 av = *reinterpret_cast<const f32x4*>(a + i), dv = *reinterpret_cast<const f32x4*>(dA + i), zv = *reinterpret_cast<const f32x4*>(z + i),
                mu = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c),
                ga = *reinterpret_cast<const f32x4*>(gamma + c), s0 = *reinterpret_cast<const f32x4*>(sums + c),
                s1 = *reinterpret_cast<const f32x4*>(sums + C + c);
    const float inv = 1.0f / (float)M;
    f32x4 g;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float dy = av[k] > 0.f ? dv[k] * post_scale : 0.f;
        const float xh = (zv[k] - mu[k]) * rs[k];
        g[k] = ga[k] * rs[k] * (dy - s0[k] * inv - xh * s1[k] * inv);
    }
    const int b = (int)(m / P), pix = (int)(m % P);
    *reinterpret_cast<f32x4*>(dz + (((size_t)b * Hz + pix / Hout + zoff) * Hz + pix % Hout + zoff) * C + c) = g;
}

// ---------------------------------------------------------------- heads: forward, losses, gradient wrt f2
// one 64-thread block per sample (A = n*n <= 64 policy outputs, lane = action)
__global__ __launch_bounds__(64) void k_t_heads(const float* __restrict__ f2 /*[B][512]*/, const int* __restrict__ d_count, int n,
                                                const float* __restrict__ Wpi /*[512][A]*/, const float* __restrict__ bpi,
                                                const float* __restrict__ Wv /*[512]*/, const float* __restrict__ bv,
                                                const float* __restrict__ pit /*[B][A]*/, const float* __restrict__ zt /*[B]*/,
                                                float* __restrict__ p_out, float* __restrict__ v_out,
                                                float* __restrict__ dlogit /*[B][A]*/, float* __restrict__ dvpre /*[B]*/,
                                                float* __restrict__ loss /*[B][2]*/) {
    const int b = blockIdx.x, lane = threadIdx.x, A = n * n, B = *d_count;
    if (b >= B) return;
    const float* x = f2 + (size_t)b * 512;
    float logit = -INFINITY, vacc = 0.f;
    {
        // 64 weight rows in flight per batch (buffer loads: one per-lane offset register + a scalar row offset each), the fmaf chain stays in k order:
        // the loop is a chain of L2 round trips, 32 of them with 16 rows in flight = 19 us per launch at the reference's batch (round 5)
        const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wpi), 0, 512 * A * 4, 0x00020000);
        const int lo = (lane < A ? lane : 0) * 4;
        float acc = 0.f;
#pragma unroll 1
        for (int k0 = 0; k0 < 512; k0 += 64) {
            float wv[64];
#pragma unroll
            for (int j = 0; j < 64; ++j) wv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, lo, (k0 + j) * A * 4, 0));
#pragma unroll
            for (int j = 0; j < 64; ++j) acc = fmaf(x[k0 + j], wv[j], acc);
        }
        if (lane < A) logit = acc + bpi[lane];
    }
    for (int k = lane; k < 512; k += 64) vacc = fmaf(x[k], Wv[k], vacc);
    for (int o = 32; o; o >>= 1) vacc += __shfl_xor(vacc, o);
    float mx = logit;
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float e = lane < A ? expf(logit - mx) : 0.f;
    float se = e;
    for (int o = 32; o; o >>= 1) se += __shfl_xor(se, o);
    const float p = e / se;
    const float v = tanhf(vacc + bv[0]);
    // row-normalised, clipped cross entropy on the (n, n) view; the lanes of one board row are n consecutive lanes
    // (every lane runs the shuffles; lanes >= A read a clamped source and their results are unused)
    const int row = lane / n;
    const float t = lane < A ? pit[(size_t)b * A + lane] : 0.f;
    float S = 0.f;
    for (int c = 0; c < n; ++c) { const int src = row * n + c; S += __shfl(p, src < 63 ? src : 63); }
    const float q = lane < A ? p / S : 0.5f;
    const bool inside = q >= 1e-7f && q <= 1.0f - 1e-7f;
    const float qc = fminf(fmaxf(q, 1e-7f), 1.0f - 1e-7f);
    float lpi = lane < A ? -t * logf(qc) : 0.f;
    for (int o = 32; o; o >>= 1) lpi += __shfl_xor(lpi, o);
    // dL/dq = -t/q inside the clip range (0 outside); q = p / S  =>  dL/dp_j = g_j / S - (sum_{c in row} g_c q_c) / S
    const float gq = (lane < A && inside) ? -t / q : 0.f;
    const float gqq = gq * q;
    float rowdot = 0.f;
    for (int c = 0; c < n; ++c) { const int src = row * n + c; rowdot += __shfl(gqq, src < 63 ? src : 63); }
    const float normp = 1.0f / ((float)B * (float)n);
    const float gp = lane < A ? (gq - rowdot) / S * normp : 0.f;
    // softmax backward: dlogit_j = p_j (gp_j - sum_k gp_k p_k)
    float dot = gp * p;
    for (int o = 32; o; o >>= 1) dot += __shfl_xor(dot, o);
    if (lane < A) {
        dlogit[(size_t)b * A + lane] = p * (gp - dot);
        p_out[(size_t)b * A + lane] = p;
    }
    if (lane == 0) {
        const float d = v - zt[b];
        v_out[b] = v;
        dvpre[b] = 2.0f * d / (float)B * (1.0f - v * v);
        loss[2 * b] = lpi / (float)n;          // per-sample mean over rows
        loss[2 * b + 1] = d * d;
    }
}
// dWpi[k][a] = sum_b f2[b][k] dlogit[b][a];  dWv[k] = sum_b f2[b][k] dvpre[b]   (block = k, thread = a; a == A handles v)
__global__ __launch_bounds__(128) void k_t_heads_wgrad(const float* __restrict__ f2, const float* __restrict__ dlogit, const float* __restrict__ dvpre,
                                                       const int* __restrict__ d_count, int A, float* __restrict__ dWpi, float* __restrict__ dWv) {
    const int k = blockIdx.x, a = threadIdx.x, B = *d_count;
    if (a > A) return;
    float acc = 0.f;
    _Pragma("unroll 8") for (int b = 0; b < B; ++b) acc = fmaf(f2[(size_t)b * 512 + k], a < A ? dlogit[(size_t)b * A + a] : dvpre[b], acc);    // 8 loads in flight, sums in batch order
    if (a < A) dWpi[(size_t)k * A + a] = acc; else dWv[k] = acc;
}
// dbpi[a] = sum_b dlogit[b][a]; dbv = sum_b dvpre[b]; losses[0..2] = total, pi, v (batch means)
__global__ __launch_bounds__(128) void k_t_heads_bias(const float* __restrict__ dlogit, const float* __restrict__ dvpre, const float* __restrict__ loss,
                                                      const int* __restrict__ d_count, int A, float* __restrict__ dbpi, float* __restrict__ dbv,
                                                      float* __restrict__ losses, unsigned* __restrict__ zero6 /* nullable: six words cleared here */) {
    const int a = threadIdx.x, B = *d_count;
    if (zero6 && a >= 120 && a < 126) zero6[a - 120] = 0u;   // the per-layer |dz| maxima of the f16x2 mode (a 24-byte memset was two fill launches on the chain)
    if (a < A) { float s = 0.f; _Pragma("unroll 8") for (int b = 0; b < B; ++b) s += dlogit[(size_t)b * A + a]; dbpi[a] = s; }
    if (a == A) { float s = 0.f; _Pragma("unroll 8") for (int b = 0; b < B; ++b) s += dvpre[b]; dbv[0] = s; }
    if (a == A + 1) {
        float lp = 0.f, lv = 0.f;
        _Pragma("unroll 8") for (int b = 0; b < B; ++b) { lp += loss[2 * b]; lv += loss[2 * b + 1]; }
        lp /= (float)B; lv /= (float)B;
        losses[0] = lp + lv; losses[1] = lp; losses[2] = lv;
    }
}
// df2[b][k] = sum_a dlogit[b][a] Wpi[k][a] + dvpre[b] Wv[k]
__global__ __launch_bounds__(256) void k_t_heads_dgrad(const float* __restrict__ dlogit, const float* __restrict__ dvpre, const float* __restrict__ Wpi,
                                                       const float* __restrict__ Wv, const int* __restrict__ d_count, int A, float* __restrict__ df2) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)(*d_count) * 512) return;
    const int b = (int)(i / 512), k = (int)(i % 512);
    float acc = dvpre[b] * Wv[k];
    for (int a = 0; a < A; ++a) acc = fmaf(dlogit[(size_t)b * A + a], Wpi[(size_t)k * A + a], acc);
    df2[i] = acc;
}

// ---------------------------------------------------------------- weight gradient: TN implicit GEMM on fp32 MFMA
// dW[t][ci][co] = sum_m Xs_t[m][ci] * dZ[m][co], m = (b, oy, ox).  Block tile 128 ci x 128 co for one tap, 4 waves of
// 64x64 (2x2 v_mfma_f32_32x32x2_f32); the k index is the row m: both operand tiles are staged [32 rows][128 channels]
// straight from their row-major tensors (coalesced, no transpose) and the MFMA operands are read k-major from LDS
// (lane (i, kk) reads element [k0 + kk][i]: 32 consecutive floats per half wave, row stride padded by 32 banks).
#define WG_STRIDE 160
struct WgradGeom { int Hin, Hout, pad, Cin, Cout, taps, Hz, zoff; };
__global__ __launch_bounds__(256) void k_wgrad_f32(const float* __restrict__ X, const float* __restrict__ dZ, const int* __restrict__ d_count,
                                                   WgradGeom g, float* __restrict__ dW, int msplit, float* __restrict__ partial, long long slab) {
    __shared__ __attribute__((aligned(16))) float lds[2][32 * WG_STRIDE];
    const int nci = g.Cin / 128, nco = g.Cout / 128;
    const int tap = blockIdx.x / (nci * nco), rem = blockIdx.x % (nci * nco), ci0 = (rem / nco) * 128, co0 = (rem % nco) * 128;
    const int P = g.Hout * g.Hout;
    const long long M = (long long)(*d_count) * P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int srow = tid >> 5, c4 = (tid & 31) * 4;              // staging: rows srow + 8 i, 4 consecutive channels
    const int dy = g.taps == 9 ? tap / 3 : 0, dx = g.taps == 9 ? tap % 3 : 0;

    f32x4 ra[4], rb[4];
    auto gload = [&](long long m0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long m = m0 + srow + 8 * i;
            f32x4 za = {0.f, 0.f, 0.f, 0.f}, zb = za;
            if (m < M) {
                const int b = (int)(m / P), pix = (int)(m % P), oy = pix / g.Hout, ox = pix % g.Hout;
                const int iy = oy - g.pad + dy, ix = ox - g.pad + dx;
                if (iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Hin)
                    za = *reinterpret_cast<const f32x4*>(X + (((size_t)b * g.Hin + iy) * g.Hin + ix) * g.Cin + ci0 + c4);
                zb = *reinterpret_cast<const f32x4*>(dZ + (((size_t)b * g.Hz + oy + g.zoff) * g.Hz + ox + g.zoff) * g.Cout + co0 + c4);
            }
            ra[i] = za; rb[i] = zb;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(&lds[0][(srow + 8 * i) * WG_STRIDE + c4]) = ra[i];
            *reinterpret_cast<f32x4*>(&lds[1][(srow + 8 * i) * WG_STRIDE + c4]) = rb[i];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int r32 = lane & 31, kk = lane >> 5;
    // rows (the k index) split over blockIdx.y when the launch is small: partial sums to partial[split], reduced in fixed order
    const long long nk_all = (M + 31) / 32, nk_s = (nk_all + msplit - 1) / msplit, kt0 = blockIdx.y * nk_s;
    const long long nk = nk_all < kt0 + nk_s ? nk_all : kt0 + nk_s;
    float* __restrict__ outp = msplit > 1 ? partial + (size_t)blockIdx.y * slab : dW;
    if (kt0 < nk) { gload(kt0 * 32); lstore(); }
    __syncthreads();
    for (long long kt = kt0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload((kt + 1) * 32);
        const float* At = &lds[0][kk * WG_STRIDE + wm * 64 + r32];
        const float* Bt = &lds[1][kk * WG_STRIDE + wn * 64 + r32];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a0 = At[2 * s * WG_STRIDE], a1 = At[2 * s * WG_STRIDE + 32];
            const float b0 = Bt[2 * s * WG_STRIDE], b1 = Bt[2 * s * WG_STRIDE + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nk) lstore();
        __syncthreads();
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31 (co), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (ci)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                const int co = co0 + wn * 64 + j * 32 + r32;
                outp[((size_t)tap * g.Cin + ci) * g.Cout + co] = acc[i][j][r];
            }
}

// ---------------------------------------------------------------- weight gradient of a 3x3 convolution, board-resident
// dW[tap][ci][co] = sum over boards b and output pixels p of X[b][p + tap][ci] * dZ[b][p][co].
// k_wgrad_f32 above (kept for the dense layers) gives every (tap, 128 ci, 128 co) tile its own block, so the two operand
// matrices stream through 9 x 4 x 4 blocks: 32 FLOP per byte staged, the launch is bound by the memory system (measured
// 30-49 TFLOP/s at batch 1024).  Here a block owns a 64 (ci) x 128 (co) tile for ALL NINE taps: one board's X tile (with its
// zero border for 'same' layers) and dZ tile are staged in LDS once and every tap reads the SAME staged pixels at a shifted
// address -- 9 accumulators per wave (one 32 x 32 MFMA tile per tap = 144 registers), 190 FLOP per byte staged, one barrier
// per board, next board prefetched into registers during the 288 MFMAs of the current one.  The k index of
// v_mfma_f32_32x32x2_f32 is the output pixel: k-steps of two pixels (Hout^2 is even for every layer of either board size).
#define WC_CI 64
#define WC_CO 128
struct WconvGeom { int Hin, Hout, pad, Cin, Cout, Hz, zoff; };
__global__ __launch_bounds__(512) void k_wgrad_conv(const float* __restrict__ X, const float* __restrict__ dZ, const int* __restrict__ d_count,
                                                    WconvGeom g, float* __restrict__ dW, int msplit, float* __restrict__ partial, long long slab) {
    extern __shared__ __attribute__((aligned(16))) float wc_lds[];
    const int XW = g.Hin + 2 * g.pad, XP = XW * XW, P = g.Hout * g.Hout;
    const int stage_floats = XP * WC_CI + P * WC_CO;
    const int nco = g.Cout / WC_CO;
    const int ci0 = (blockIdx.x / nco) * WC_CI, co0 = (blockIdx.x % nco) * WC_CO;
    const int B = *d_count;
    const int per = (B + msplit - 1) / msplit, b0 = blockIdx.y * per, b1 = b0 + per < B ? b0 + per : B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3, r32 = lane & 31, kk = lane >> 5;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // staging: float4 units; X tile = XP pixels x 16 units (border pixels are zeros), dZ tile = P pixels x 32 units
    const int xu = XP * (WC_CI / 4), zu = P * (WC_CO / 4);
    f32x4 rx[4], rz[4];
    auto gload = [&](int b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = tid + 512 * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (u < xu) {
                const int px = u >> 4, c4 = (u & 15) * 4, iy = px / XW - g.pad, ix = px % XW - g.pad;
                if (iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Hin)
                    v = *reinterpret_cast<const f32x4*>(X + (((size_t)b * g.Hin + iy) * g.Hin + ix) * g.Cin + ci0 + c4);
            }
            rx[i] = v;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (u < zu) {
                const int px = u >> 5, c4 = (u & 31) * 4, oy = px / g.Hout, ox = px % g.Hout;
                z = *reinterpret_cast<const f32x4*>(dZ + (((size_t)b * g.Hz + oy + g.zoff) * g.Hz + ox + g.zoff) * g.Cout + co0 + c4);
            }
            rz[i] = z;
        }
    };
    auto lstore = [&](int st) {
        float* Xs = wc_lds + st * stage_floats;
        float* Zs = Xs + XP * WC_CI;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = tid + 512 * i;
            if (u < xu) *reinterpret_cast<f32x4*>(Xs + u * 4) = rx[i];
            if (u < zu) *reinterpret_cast<f32x4*>(Zs + u * 4) = rz[i];
        }
    };
    if (b0 < b1) { gload(b0); lstore(0); }
    __syncthreads();
    for (int b = b0; b < b1; ++b) {
        const int st = (b - b0) & 1;
        if (b + 1 < b1) gload(b + 1);
        const float* Xs = wc_lds + st * stage_floats + wm * 32 + r32;
        const float* Zs = wc_lds + st * stage_floats + XP * WC_CI + wn * 32 + r32;
        for (int s2 = 0; s2 < P / 2; ++s2) {
            const int p = 2 * s2 + kk, oy = p / g.Hout, ox = p - oy * g.Hout;
            const float zb = Zs[p * WC_CO];
            const float* xa = Xs + (oy * XW + ox) * WC_CI;
#pragma unroll
            for (int t = 0; t < 9; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[((t / 3) * XW + (t % 3)) * WC_CI], zb, acc[t], 0, 0, 0);
        }
        if (b + 1 < b1) lstore(st ^ 1);             // the other stage was last read before the previous barrier
        __syncthreads();
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31 (co), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (ci)
    float* __restrict__ outp = msplit > 1 ? partial + (size_t)blockIdx.y * slab : dW;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            const int co = co0 + wn * 32 + r32;
            outp[((size_t)t * g.Cin + ci) * g.Cout + co] = acc[t][r];
        }
}

// ---------------------------------------------------------------- operand layouts derived from the Keras-layout masters
// out[c][r] = in[r][c]   (forward operand Wt[N][K] of k_gemm_f32 from the Keras kernel [K][N])
__global__ __launch_bounds__(256) void k_t_transpose(const float* __restrict__ in, int R, int Cc, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) if (by + i < R && bx + tx < Cc) tile[i][tx] = in[(size_t)(by + i) * Cc + bx + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8) if (bx + i < Cc && by + tx < R) out[(size_t)(bx + i) * R + by + tx] = tile[tx][i];
}
// data-gradient operand of a 3x3 convolution: Wd[ci][(8 - t) * Cout + co] = W[t][ci][co]
__global__ __launch_bounds__(256) void k_t_dgrad_operand(const float* __restrict__ W, int Cin, int Cout, float* __restrict__ Wd) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 9LL * Cin * Cout) return;
    const int co = (int)(i % Cout), ci = (int)((i / Cout) % Cin), t = (int)(i / ((long long)Cout * Cin));
    Wd[(size_t)ci * 9 * Cout + (size_t)(8 - t) * Cout + co] = W[i];
}

// ---------------------------------------------------------------- f16x2 mode of the 3x3 layers (forward and data gradient on k_gemm_h2)
// An fp32 value travels as two fp16 planes (oz_net_h2.h: 22 significand bits, 3 fp16 MFMA products per fp32 product at 16x the fp32
// matrix rate).  fp16 has 5 exponent bits, so every tensor is moved into range by an exact power of two taken from its own
// maximum, on the device and per step: weights to max |w| ~ 2^9 (as at oz_net_commit), data gradients to max |dz| ~ 2^13; the
// inverse powers ride in the GEMM's per-column scale, so scaling itself changes no bit.  Small elements of a gradient tensor
// (below 2^-11 of its maximum) lose relative precision in the second plane -- an absolute error of 2^-37 of the tensor's
// maximum, far below the fp32 rounding of the sums they enter.  Activations are post-ReLU values < 65504 (sticky flag otherwise).
// Where in the fp16 window the per-step scales put a tensor's maximum: 2^9 .. 2^10 for the weights, 2^13 for the data gradients.  The inference path
// moved its maxima to 2^-2 in round 4 (fewer residual bits for small elements = a higher sustained clock, tools/target_probe.py); that is NOT copied here:
// these scales are per TENSOR, and a gradient tensor's elements lie many binary orders below its maximum -- at a maximum of 2^-2 an element 2^-10 of it
// would keep 13 bits, at 2^13 it keeps all 22.
#define T_W_TARGET 1000.0f
#define T_DZ_TARGET 8192.0f
typedef _Float16 t_f16x8 __attribute__((ext_vector_type(8)));
struct AbsMaxArgs { const float* p[3]; long long n[3]; };
// out[l] = bits of max |p[l][i]| (non-negative floats order like their bit patterns); out is zeroed by the caller
__global__ __launch_bounds__(256) void k_t_absmax(AbsMaxArgs a, unsigned* __restrict__ out) {
    const int l = blockIdx.y;
    const float* p = a.p[l];
    float m = 0.f;
    // twelve 16-byte loads in flight per trip (the one-load-per-trip loop took 40 us for 28 MB at the reference's batch, round 5)
    const long long stride = (long long)gridDim.x * 1024;
    for (long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i0 < a.n[l]; i0 += 12 * stride) {
        f32x4 v[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) { const long long i = i0 + q * stride; v[q] = i < a.n[l] ? *reinterpret_cast<const f32x4*>(p + i) : f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            m = fmaxf(fmaxf(fabsf(v[q][0]), fabsf(v[q][1])), fmaxf(m, fmaxf(fabsf(v[q][2]), fabsf(v[q][3]))));
            if (v[q][0] != v[q][0] || v[q][1] != v[q][1] || v[q][2] != v[q][2] || v[q][3] != v[q][3]) m = INFINITY;   // fmaxf drops NaNs: a NaN weight must not pass as finite
        }
    }
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    // ONE atomic per block: an atomic on one address costs ~12 ns whoever issues it, and the 3072 per-wave atomics of this launch were its 40 us
    // (round 5: the loads in flight changed nothing)
    __shared__ float wmaxs[4];
    if ((threadIdx.x & 63) == 0) wmaxs[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out + l, __float_as_uint(fmaxf(fmaxf(wmaxs[0], wmaxs[1]), fmaxf(wmaxs[2], wmaxs[3]))));
}
// power-of-two exponent that brings a tensor whose |maximum| has bit pattern max_bits to ~target; a non-finite maximum (a diverged
// step) gives exponent 0 and t_bad_max() -- the converting kernels raise the sticky flag (bit 2) so that the step fails loudly
__device__ __forceinline__ bool t_bad_max(unsigned max_bits) { return !(__uint_as_float(max_bits) <= 3.402823466e38f); }      // Inf or NaN
__device__ __forceinline__ int t_exp_for(unsigned max_bits, float target) {
    const float mx = __uint_as_float(max_bits);
    if (!(mx > 0.f) || t_bad_max(max_bits)) return 0;
    const int e = (int)floorf(log2f(target / mx));
    return e < -120 ? -120 : e > 120 ? 120 : e;
}
__device__ __forceinline__ void t_store_h2(uint4* dst, const float* v) {
    t_f16x8 h1, h2;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const _Float16 a = (_Float16)v[j]; h1[j] = a; h2[j] = (_Float16)(v[j] - (float)a); }
    dst[0] = *reinterpret_cast<uint4*>(&h1);
    dst[1] = *reinterpret_cast<uint4*>(&h2);
}
// Keras kernel W[9][Cin][Cout] fp32 -> a k_gemm_h2 weight operand in the h2 layout, k order k' = (slice * 9 + tap) * 32 + c32,
// scaled by 2^kexp (kexp from the tensor's maximum), straight from the master weights (no fp32 intermediate):
//   DGRAD = 0  forward operand:       row n = co, channel of k' = ci:  W[tap][ci][co]        (threads adjacent in co: coalesced reads)
//   DGRAD = 1  data-gradient operand: row n = ci, channel of k' = co:  W[8 - tap][ci][co]    (the reversed, channel-swapped taps;
//              a thread's 8 values are 8 consecutive co = one 32-byte read)
// scale_out[n] = 2^-kexp (forward only).  One thread per (row, group of 8 k').
template <int DGRAD>
__global__ __launch_bounds__(256) void k_t_w_to_h2(const float* __restrict__ W, int Cin, int Cout, const unsigned* __restrict__ wmax,
                                                   uint4* __restrict__ out, float* __restrict__ scale_out, int* __restrict__ flag) {
    const int N = DGRAD ? Cin : Cout, Cch = DGRAD ? Cout : Cin, ng = 9 * Cch / 8;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int nrow, grp;
    if (DGRAD) { grp = (int)(idx % ng); nrow = (int)(idx / ng); }
    else { nrow = (int)(idx % N); grp = (int)(idx / N); }
    if (idx >= (long long)N * ng) return;
    const int kexp = t_exp_for(*wmax, T_W_TARGET);
    if (idx == 0 && t_bad_max(*wmax)) atomicOr(flag, 2);
    const int kp = grp * 8, tile = kp >> 5, c32 = kp & 31, slice = tile / 9, tap = tile - slice * 9, ch = slice * 32 + c32;
    float v[8];
    if (DGRAD) {
        const float* q = W + ((size_t)(8 - tap) * Cin + nrow) * Cout + ch;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ldexpf(q[j], kexp);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ldexpf(W[((size_t)tap * Cin + ch + j) * Cout + nrow], kexp);
    }
    t_store_h2(out + ((size_t)nrow * ng + grp) * 2, v);
    if (!DGRAD && grp == 0) scale_out[nrow] = ldexpf(1.0f, -kexp);
}
// activation rows [M][C] fp32 -> h2 layout (the next layer's A operand)
__global__ __launch_bounds__(256) void k_t_act_to_h2(const float* __restrict__ a, const int* __restrict__ d_count, int P, int C,
                                                     uint4* __restrict__ out, int* __restrict__ flag) {
    const int cg = C >> 3;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x, m = idx / cg;
    if (m >= (long long)(*d_count) * P) return;
    const int g8 = (int)(idx % cg);
    const float* q = a + (size_t)m * C + g8 * 8;
    float v[8];
    bool over = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = q[j]; over |= fabsf(v[j]) > 65504.0f; }
    t_store_h2(out + ((size_t)m * cg + g8) * 2, v);
    if (over) atomicOr(flag, 1);
}
// dz (zero-bordered [B][Hz][Hz][C] fp32, interior Hout^2 at offset zoff) -> the same geometry in the h2 layout, scaled by 2^ez with ez
// from the tensor's maximum (dzmax, left by the BN backward); dscale[c] = 2^-(ez + kexp of the data-gradient weights)
__global__ __launch_bounds__(256) void k_t_dz_to_h2(const float* __restrict__ dz, const int* __restrict__ d_count, int Hout, int Hz, int zoff, int C,
                                                    const unsigned* __restrict__ dzmax, const unsigned* __restrict__ wmax,
                                                    uint4* __restrict__ out, float* __restrict__ dscale, int ncols, int* __restrict__ flag) {
    const int cg = C >> 3, P = Hout * Hout;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x, m = idx / cg;
    const int ez = t_exp_for(*dzmax, T_DZ_TARGET);
    if (idx == 0 && (t_bad_max(*dzmax) || t_bad_max(*wmax))) atomicOr(flag, 2);
    if (idx < ncols) dscale[idx] = ldexpf(1.0f, -(ez + t_exp_for(*wmax, T_W_TARGET)));
    if (m >= (long long)(*d_count) * P) return;
    const int g8 = (int)(idx % cg), b = (int)(m / P), pix = (int)(m % P);
    const size_t row = ((size_t)b * Hz + pix / Hout + zoff) * Hz + pix % Hout + zoff;
    const float* q = dz + row * C + g8 * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = ldexpf(q[j], ez);
    t_store_h2(out + (row * cg + g8) * 2, v);
}

// ---------------------------------------------------------------- f16x2 weight gradient of the 3x3 layers (round 3)
// dW[tap][ci][co] = sum over boards b and output pixels p of X[b][p + tap][ci] * dZ[b][p][co]: the contraction runs over (board, pixel),
// the WRONG major for the fp16 operand map (v_mfma_f32_16x16x32_f16 wants, per lane, 8 consecutive k of one row / column; the tensors are
// stored pixel-major with the channels contiguous).  The k-groups of 8 are therefore 8 BOARDS at one pixel: a tap shift changes the pixel,
// never the alignment of a k-group, and the four k-groups of one MFMA are four neighbouring output pixels.  Both tensors are first written
// as "octet images" -- [octet of 8 boards][row][pixel slot][plane h1 | h2][channel][8 boards] fp16, 16 bytes per (pixel, plane, channel) --
// by two bandwidth-bound kernels (X with its zero border for 'same' layers and zero columns up to WH_XW; dZ scaled into the fp16 range by the
// power of two the data gradient uses, zero columns up to WH_ZW: a row of either is two whole k-quads for every layer), so the GEMM kernel
// stages with LDS-DMA only: no transposes, no staging registers.
// k_wgrad_h2: a block owns a 64 (ci) x 128 (co) tile for ALL NINE taps (9 x 4 accumulator tiles of 16 x 16 per wave, 8 waves = 4 x 16 ci by
// 2 x 64 co).  Per octet it walks the output rows: three X rows (a ring of four 20 KB row slots: the row of the next step streams in during
// the MFMAs of this one) and one dZ row (two 32 KB buffers) are resident; every tap reads the SAME staged rows at a shifted pixel slot, and
// the dZ fragments of a pixel quad are read once for all nine taps: 26 ds_read_b128 per 108 MFMAs per wave.  fp32 accumulation, three
// fp16 products per fp32 product (x2*z1 + x1*z2 + x1*z1), 2^-ez folded into the epilogue.  Octet ranges split over blockIdx.y until every CU
// has a block (raw slabs + k_t_sum_partials, fixed order).  Rows of fewer than 8 real pixels (conv3: 6, conv4: 4) pay for the zero columns
// (+24 % MFMA work over the three layers) -- the price of one uniform step shape.
#define WH_MIN_BATCH 32     // batches from here on take this kernel (round 6: was 128; at the reference's batch of 32 the fp32 weight gradients on the second stream -- 52 + 82 +
                            // 151 us -- WERE the critical path of the backward pass)
#define WH_CI 64
#define WH_CO 128
#define WH_XW 10           // pixel slots of an X row (columns >= Hin + 2 pad hold zeros)
#define WH_ZW 8            // pixel slots of a dZ row (columns >= Hout hold zeros)
#define WH_XROW (2 * WH_XW * WH_CI * 16)        // bytes of one staged X row: [plane][slot][ci] x 16 B = 20 KB
#define WH_ZROW (2 * WH_ZW * WH_CO * 16)        // bytes of one staged dZ row: [plane][slot][co] x 16 B = 32 KB
#define WH_LDS (4 * WH_XROW + 2 * WH_ZROW)      // 144 KB
// X octet image of a[l - 1] ([B][Hin][Hin][C] fp32): out[octet][row < Hin + 2 pad][slot < WH_XW][plane][C] x 16 B.
// One thread per (cell, 4 channels): eight 16-byte loads (one per board), eight 16-byte stores (4 channels x 2 planes).
__device__ __forceinline__ void t_store_octet4(uint4* __restrict__ dst, int C, const f32x4* v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        t_f16x8 h1, h2;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const _Float16 x = (_Float16)v[b][q];
            h1[b] = x; h2[b] = (_Float16)(v[b][q] - (float)x);
        }
        dst[q] = *reinterpret_cast<uint4*>(&h1);
        dst[C + q] = *reinterpret_cast<uint4*>(&h2);
    }
}
__global__ __launch_bounds__(256) void k_t_x_octets(const float* __restrict__ a, const int* __restrict__ d_count, int Hin, int pad, int C, uint4* __restrict__ out) {
    const int B = *d_count, XR = Hin + 2 * pad, noct = (B + 7) >> 3, C4 = C >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)noct * XR * WH_XW * C4) return;
    const int ch = (int)(idx % C4) * 4;
    const long long cell = idx / C4;
    const int c = (int)(cell % WH_XW), r = (int)((cell / WH_XW) % XR), oct = (int)(cell / ((long long)WH_XW * XR));
    const int iy = r - pad, ix = c - pad;
    const bool inside = iy >= 0 && iy < Hin && ix >= 0 && ix < Hin;
    f32x4 v[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const int bb = oct * 8 + b;
        v[b] = (inside && bb < B) ? *reinterpret_cast<const f32x4*>(a + (((size_t)bb * Hin + iy) * Hin + ix) * C + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    t_store_octet4(out + (size_t)cell * 2 * C + ch, C, v);
}
// dZ octet image of dz[l] (zero-bordered [B][Hz][Hz][C] fp32, interior Hout^2 at offset zoff), scaled by 2^ez (ez from the tensor's maximum):
// out[octet][row < Hout][slot < WH_ZW][plane][C] x 16 B
__global__ __launch_bounds__(256) void k_t_z_octets(const float* __restrict__ dz, const int* __restrict__ d_count, int Hout, int Hz, int zoff, int C,
                                                    const unsigned* __restrict__ dzmax, uint4* __restrict__ out, int* __restrict__ flag) {
    const int B = *d_count, noct = (B + 7) >> 3, C4 = C >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0 && t_bad_max(*dzmax)) atomicOr(flag, 2);
    if (idx >= (long long)noct * Hout * WH_ZW * C4) return;
    const int ez = t_exp_for(*dzmax, T_DZ_TARGET);
    const int ch = (int)(idx % C4) * 4;
    const long long cell = idx / C4;
    const int c = (int)(cell % WH_ZW), r = (int)((cell / WH_ZW) % Hout), oct = (int)(cell / ((long long)WH_ZW * Hout));
    f32x4 v[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const int bb = oct * 8 + b;
        v[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < Hout && bb < B) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(dz + (((size_t)bb * Hz + r + zoff) * Hz + c + zoff) * C + ch);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[b][q] = ldexpf(x[q], ez);
        }
    }
    t_store_octet4(out + (size_t)cell * 2 * C + ch, C, v);
}
typedef float t_f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* t_gptr;
typedef __attribute__((address_space(3))) void* t_lptr;
struct WhGeom { int XR, Hout, Cin, Cout; };
__global__ __launch_bounds__(512, 2) void k_wgrad_h2(const uint4* __restrict__ Xt, const uint4* __restrict__ Zt, const int* __restrict__ d_count, WhGeom g,
                                                     const unsigned* __restrict__ dzmax, float* __restrict__ dW, int msplit, float* __restrict__ partial,
                                                     long long slab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wh_lds[];
    unsigned char* const Xring = wh_lds;
    unsigned char* const Zbuf = wh_lds + 4 * WH_XROW;
    const int nco = g.Cout / WH_CO;
    const int ci0 = (blockIdx.x / nco) * WH_CI, co0 = (blockIdx.x % nco) * WH_CO;
    const int noct = (*d_count + 7) >> 3;
    const int per = (noct + msplit - 1) / msplit, o0 = blockIdx.y * per, o1 = o0 + per < noct ? o0 + per : noct;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wm = wave >> 1, wn = wave & 1, r16 = lane & 15, g4 = lane >> 4;

    // LDS-DMA: one instruction = 64 consecutive 16-byte entries.  X row = 20 instructions q = (plane, slot): lanes = the 64 ci of the tile;
    // dZ row = 32 instructions q = (plane, slot, half): lanes = 64 of the tile's 128 co.  Wave w issues q = w, w + 8, ...
    auto dma_x = [&](int oct, int row, int slot) {
        const uint4* src = Xt + (((size_t)oct * g.XR + row) * WH_XW) * 2 * g.Cin + ci0 + lane;
        for (int q = wave; q < 2 * WH_XW; q += 8) {
            const int p = q / WH_XW, c = q - p * WH_XW;
            __builtin_amdgcn_global_load_lds((t_gptr)(src + ((size_t)c * 2 + p) * g.Cin), (t_lptr)(Xring + slot * WH_XROW + q * 1024), 16, 0, 0);
        }
    };
    auto dma_z = [&](int oct, int row, int buf) {
        const uint4* src = Zt + (((size_t)oct * g.Hout + row) * WH_ZW) * 2 * g.Cout + co0 + lane;
        for (int q = wave; q < 4 * WH_ZW; q += 8) {
            const int h = q & 1, c = (q >> 1) % WH_ZW, p = q / (2 * WH_ZW);
            __builtin_amdgcn_global_load_lds((t_gptr)(src + ((size_t)c * 2 + p) * g.Cout + h * 64), (t_lptr)(Zbuf + buf * WH_ZROW + ((p * WH_ZW + c) * 2 + h) * 1024), 16, 0, 0);
        }
    };

    t_f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = t_f32x4{0.f, 0.f, 0.f, 0.f};

    for (int oct = o0; oct < o1; ++oct) {
        // the three X rows and the dZ row of output row 0 (the previous octet's last step ended with a barrier: every slot is free)
        dma_x(oct, 0, 0); dma_x(oct, 1, 1); dma_x(oct, 2, 2); dma_z(oct, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int oy = 0; oy < g.Hout; ++oy) {
            if (oy + 1 < g.Hout) { dma_x(oct, oy + 3, (oy + 3) & 3); dma_z(oct, oy + 1, (oy + 1) & 1); }     // slots last read in step oy - 1
            const unsigned char* Zr = Zbuf + (oy & 1) * WH_ZROW + (wn * 64 + r16) * 16;
#pragma unroll
            for (int quad = 0; quad < 2; ++quad) {
                const int px = quad * 4 + g4;
                t_f16x8 z1[4], z2[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    z1[j] = *reinterpret_cast<const t_f16x8*>(Zr + ((0 * WH_ZW + px) * WH_CO + j * 16) * 16);
                    z2[j] = *reinterpret_cast<const t_f16x8*>(Zr + ((1 * WH_ZW + px) * WH_CO + j * 16) * 16);
                }
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int dy = t / 3, dx = t - 3 * dy;
                    const unsigned char* Xr = Xring + ((oy + dy) & 3) * WH_XROW + (wm * 16 + r16) * 16;
                    const t_f16x8 x1 = *reinterpret_cast<const t_f16x8*>(Xr + ((0 * WH_XW + px + dx) * WH_CI) * 16);
                    const t_f16x8 x2 = *reinterpret_cast<const t_f16x8*>(Xr + ((1 * WH_XW + px + dx) * WH_CI) * 16);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x2, z1[j], acc[t][j], 0, 0, 0);
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x1, z2[j], acc[t][j], 0, 0, 0);
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x1, z1[j], acc[t][j], 0, 0, 0);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next step's rows have landed (this wave's pieces) ...
            __syncthreads();                                       // ... everybody's, and everybody is done reading this step's
        }
    }
    // C/D layout of the 16 x 16 MFMA: col = lane & 15 (co), row = (lane >> 4) * 4 + reg (ci)
    const float sc = ldexpf(1.0f, -t_exp_for(*dzmax, T_DZ_TARGET));
    float* __restrict__ outp = msplit > 1 ? partial + (size_t)blockIdx.y * slab : dW;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci0 + wm * 16 + g4 * 4 + r, co = co0 + wn * 64 + j * 16 + r16;
                outp[((size_t)t * g.Cin + ci) * g.Cout + co] = acc[t][j][r] * sc;
            }
}

// ---------------------------------------------------------------- Adam (tf.keras formulation) with clipvalue
__global__ __launch_bounds__(256) void k_t_adam(float* __restrict__ P, const float* __restrict__ G, float* __restrict__ M1, float* __restrict__ V2,
                                                long long count, float lr_t, float clip) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float g = G[i];
    if (clip > 0.f) g = fminf(fmaxf(g, -clip), clip);
    const float m = 0.9f * M1[i] + 0.1f * g;
    const float v = 0.999f * V2[i] + 0.001f * g * g;
    M1[i] = m; V2[i] = v;
    P[i] -= lr_t * m / (sqrtf(v) + 1e-7f);
}

// ---------------------------------------------------------------- host object
struct oz_trainer {
    int n = 8, C = 512, cin = 2, Bmax = 32, device = 0;
    float lr = 1e-3f, clip = 0.5f, rate = 0.3f, mom = 0.99f;
    uint64_t seed = 0;
    int64_t step = 0;
    hipStream_t s = nullptr;
    int64_t size[40], toff[40];          // element counts; offset in the trainable arena (-1: moving statistic)
    int64_t total = 0;
    float *P = nullptr, *G = nullptr, *M1 = nullptr, *V2 = nullptr;
    bool own_grads = true;
    float *stats[40] = {}, *stats_new[40] = {};
    float *Wt[6] = {}, *Wd[6] = {};
    uint64_t *d_own = nullptr, *d_opp = nullptr;
    float *d_pit = nullptr, *d_zt = nullptr;
    int* d_count = nullptr;
    unsigned char *d_in = nullptr, *h_in = nullptr;      // the device block the five pointers above point into, and its pinned host mirror
    size_t in_bytes = 0;
    float *z[6] = {}, *a[6] = {}, *mean[6] = {}, *rstd[6] = {}, *dz[6] = {};
    float *dA[2] = {}, *sums = nullptr, *partial = nullptr, *ones = nullptr, *zeros = nullptr;
    float *p = nullptr, *v = nullptr, *dlogit = nullptr, *dvpre = nullptr, *loss = nullptr, *losses = nullptr;
    float* gpartial = nullptr;           // split-K scratch of the small-batch GEMMs
    float* wpartial = nullptr;           // row-split scratch of the weight-gradient launches (they run on s2, beside the data-gradient chain)
    hipStream_t s2 = nullptr;
    hipEvent_t ev_dz[6] = {}, ev_w = nullptr;
    hipEvent_t ev_pre = nullptr, ev_wt = nullptr, ev_wd = nullptr;      // derived weight operands are rebuilt on s2 beside the first forward kernels
    hipEvent_t ev_wl[4] = {nullptr, nullptr, nullptr, nullptr};         // f16x2: the h2 forward operand of layer l = 1 .. 3 is ready (the main stream waits layer by layer)
    bool wait_wt = false, wait_wd = false, wait_wl[4] = {false, false, false, false};
    // f16x2 mode (oz_trainer_set_precision 1): conv2..4 forward and data gradient on k_gemm_h2
    int h2 = 0;
    uint4 *Wh[4] = {}, *Whd[4] = {}, *a_h2[3] = {}, *dz_h2[4] = {};
    float *wscale[4] = {}, *dscale[4] = {};
    unsigned *wmax = nullptr, *dzmax = nullptr;          // [3] max |w| of conv2..4, [6] max |dz| per layer (bit patterns)
    int* h2flag = nullptr;
    float* bnb2[6] = {};                 // per-layer row-block partials of the bias gradient (mid-size BN backward)
    bool overlap = true;                 // weight gradients on the second stream beside the data-gradient chain (false: all on the main stream)
    long long gpartial_floats = 40LL << 20;      // 160 MB each: 16 row-split slabs of a 3x3 x 512 x 512 weight gradient
    bool wconv_attr = false, wh_attr = false;
    uint4 *xt_oct[4] = {}, *zt_oct[4] = {};     // f16x2 weight gradient: octet images of a[l - 1] / dz[l] (k_t_x_octets / k_t_z_octets)
    // HBM-resident data set of a fit (oz_trainer_set_dataset / oz_trainer_fit_epoch)
    uint64_t *ds_own = nullptr, *ds_opp = nullptr;
    float *ds_pi = nullptr, *ds_z = nullptr;
    int* ds_order = nullptr;
    double* ds_acc = nullptr;
    int64_t ds_n = 0, ds_cap = 0;
    int split_mask = 7;                  // 1 forward, 2 dense dgrad, 4 conv dgrad GEMMs may split K
    int P_[6], Co[6], Hout[6], Hz[6], zoff[6];
    std::vector<void*> allocs;
    bool dirty = true;                   // derived operands need a refresh
    std::mutex mu;                       // one caller at a time (ThreadWorker-style Python threads)

    ~oz_trainer() {
        hipSetDevice(device);
        if (s) hipStreamSynchronize(s);
        for (void* q : allocs) hipFree(q);
        for (void* q : {(void*)ds_own, (void*)ds_opp, (void*)ds_pi, (void*)ds_z, (void*)ds_order, (void*)ds_acc}) if (q) hipFree(q);
        if (h_in) hipHostFree(h_in);
        for (hipEvent_t e : ev_dz) if (e) hipEventDestroy(e);
        if (ev_w) hipEventDestroy(ev_w);
        for (hipEvent_t e : {ev_pre, ev_wt, ev_wd, ev_wl[1], ev_wl[2], ev_wl[3]}) if (e) hipEventDestroy(e);
        if (s2) hipStreamDestroy(s2);
        if (s) hipStreamDestroy(s);
    }
    float* param(int idx) { return toff[idx] >= 0 ? P + toff[idx] : stats[idx]; }
    float* grad(int idx) { return G + toff[idx]; }
};

// element counts of the 40 get_weights() arrays and their offsets in the trainable arena (-1: moving statistic);
// every array starts 16-byte aligned.  Returns the arena size in floats.
static int64_t t_layout(int n, int C, int cin, int64_t* size, int64_t* toff) {
    const int64_t A = n * n, F = (int64_t)(n - 4) * (n - 4) * C;
    const int cins[4] = {cin, C, C, C};
    int idx = 0;
    for (int l = 0; l < 4; ++l) { size[idx++] = 9LL * cins[l] * C; for (int j = 0; j < 5; ++j) size[idx++] = C; }
    size[idx++] = F * 1024; for (int j = 0; j < 5; ++j) size[idx++] = 1024;
    size[idx++] = 1024LL * 512; for (int j = 0; j < 5; ++j) size[idx++] = 512;
    size[36] = 512 * A; size[37] = A; size[38] = 512; size[39] = 1;
    int64_t total = 0;
    for (int i = 0; i < 40; ++i) {
        const bool stat = i < 36 && (i % 6 == 4 || i % 6 == 5);
        toff[i] = stat ? -1 : total;
        if (!stat) total += (size[i] + 3) / 4 * 4;
    }
    return total;
}

static int t_alloc(oz_trainer* t, void** out, size_t bytes, bool zero = true) {
    OZ_HIP(hipMalloc(out, bytes ? bytes : 16));
    t->allocs.push_back(*out);
    if (zero) OZ_HIP(hipMemsetAsync(*out, 0, bytes ? bytes : 16, t->s));
    return OZ_OK;
}
#define T_LOCK(t) std::lock_guard<std::mutex> lock__((t)->mu)
#define T_ALLOC(ptr, count) do { if (int rc__ = t_alloc(t, (void**)&(ptr), (size_t)(count) * sizeof(*(ptr)))) return rc__; } while (0)

OZ_API int oz_trainer_create(oz_trainer** out, int n, int channels, int in_channels, int max_batch, float lr, float clipvalue,
                             float dropout, float bn_momentum, uint64_t seed, float* external_grads) {
    OZ_REQUIRE(out, "oz_trainer_create: out is NULL");
    OZ_REQUIRE(n == 6 || n == 8, "oz_trainer_create: board size %d (6 or 8: conv3/conv4 are 'valid')", n);
    OZ_REQUIRE(channels >= 128 && channels % 128 == 0, "oz_trainer_create: channels %d must be a multiple of 128", channels);
    OZ_REQUIRE(in_channels == 1 || in_channels == 2, "oz_trainer_create: in_channels %d", in_channels);
    OZ_REQUIRE(max_batch >= 1 && max_batch <= 65536, "oz_trainer_create: max_batch %d", max_batch);
    OZ_REQUIRE(dropout >= 0.f && dropout < 1.f, "oz_trainer_create: dropout %f", dropout);
    oz_trainer* t = new oz_trainer();
    t->n = n; t->C = channels; t->cin = in_channels; t->Bmax = max_batch; t->lr = lr; t->clip = clipvalue; t->rate = dropout;
    t->mom = bn_momentum; t->seed = seed; t->device = oz_current_device();
    auto fail = [&](int rc) { delete t; return rc; };
    if (hipSetDevice(t->device) != hipSuccess || hipStreamCreate(&t->s) != hipSuccess) { oz_set_error("oz_trainer_create: no GPU stream"); return fail(OZ_ERR_HIP); }
    const int C = channels, A = n * n, F = (n - 4) * (n - 4) * C;
    t->total = t_layout(n, C, in_channels, t->size, t->toff);
    int rc = [&]() -> int {
        T_ALLOC(t->P, t->total); T_ALLOC(t->M1, t->total); T_ALLOC(t->V2, t->total);
        if (external_grads) { t->G = external_grads; t->own_grads = false; } else T_ALLOC(t->G, t->total);
        for (int i = 0; i < 36; ++i) if (t->toff[i] < 0) { T_ALLOC(t->stats[i], t->size[i]); T_ALLOC(t->stats_new[i], t->size[i]); }
        const int Hs[6] = {n, n, n - 2, n - 4, 1, 1};
        for (int l = 0; l < 6; ++l) {
            t->Hout[l] = Hs[l]; t->P_[l] = Hs[l] * Hs[l]; t->Co[l] = l < 4 ? C : (l == 4 ? 1024 : 512);
            // conv3 / conv4 ('valid'): dz lives in a zero-bordered (Hout + 4)^2 buffer so that the data gradient is a plain valid conv
            t->zoff[l] = (l == 2 || l == 3) ? 2 : 0; t->Hz[l] = Hs[l] + 2 * t->zoff[l];
            const size_t rows = (size_t)max_batch * t->P_[l];
            T_ALLOC(t->z[l], rows * t->Co[l]); T_ALLOC(t->a[l], rows * t->Co[l]);
            T_ALLOC(t->dz[l], (size_t)max_batch * t->Hz[l] * t->Hz[l] * t->Co[l]);
            T_ALLOC(t->mean[l], t->Co[l]); T_ALLOC(t->rstd[l], t->Co[l]);
        }
        const size_t amax = (size_t)max_batch * (size_t)A * C > (size_t)max_batch * 1024 ? (size_t)max_batch * A * C : (size_t)max_batch * 1024;
        T_ALLOC(t->dA[0], amax); T_ALLOC(t->dA[1], amax);
        T_ALLOC(t->sums, 2 * 8192);
        size_t pmax = (size_t)RED_S * 2 * 1024;
        if ((size_t)RED_S * 2 * C > pmax) pmax = (size_t)RED_S * 2 * C;
        if ((size_t)RED_S * 9 * in_channels * C > pmax) pmax = (size_t)RED_S * 9 * in_channels * C;
        T_ALLOC(t->partial, pmax);
        const int vmax = F > 8192 ? F : 8192;
        T_ALLOC(t->ones, vmax); T_ALLOC(t->zeros, vmax);
        std::vector<float> one(vmax, 1.0f);
        OZ_HIP(hipMemcpyAsync(t->ones, one.data(), vmax * sizeof(float), hipMemcpyHostToDevice, t->s));
        OZ_HIP(hipStreamSynchronize(t->s));
        for (int l = 1; l < 6; ++l) T_ALLOC(t->Wt[l], t->size[6 * l]);
        for (int l = 1; l < 4; ++l) T_ALLOC(t->Wd[l], t->size[6 * l]);
        // the inputs of a step live in ONE device block [own | opp | pi targets | z targets | count] mirrored by a pinned host block: the step-wise
        // oz_trainer_forward_backward uploads it with one copy (five copies were five blit launches, ~30 us, in front of every step)
        t->in_bytes = (size_t)max_batch * (8 + 8 + 4 * A + 4) + 16;
        T_ALLOC(t->d_in, t->in_bytes);
        OZ_HIP(hipHostMalloc((void**)&t->h_in, t->in_bytes, hipHostMallocDefault));
        t->d_own = reinterpret_cast<uint64_t*>(t->d_in);
        t->d_opp = t->d_own + max_batch;
        t->d_pit = reinterpret_cast<float*>(t->d_opp + max_batch);
        t->d_zt = t->d_pit + (size_t)max_batch * A;
        t->d_count = reinterpret_cast<int*>(t->d_zt + max_batch);
        T_ALLOC(t->p, (size_t)max_batch * A); T_ALLOC(t->v, max_batch); T_ALLOC(t->dlogit, (size_t)max_batch * A); T_ALLOC(t->dvpre, max_batch);
        T_ALLOC(t->loss, 2 * (size_t)max_batch); T_ALLOC(t->losses, 4);
        T_ALLOC(t->gpartial, t->gpartial_floats);
        T_ALLOC(t->wpartial, t->gpartial_floats);
        OZ_HIP(hipStreamCreateWithFlags(&t->s2, hipStreamNonBlocking));
        for (int l = 0; l < 6; ++l) OZ_HIP(hipEventCreateWithFlags(&t->ev_dz[l], hipEventDisableTiming));
        OZ_HIP(hipEventCreateWithFlags(&t->ev_w, hipEventDisableTiming));
        for (hipEvent_t* e : {&t->ev_pre, &t->ev_wt, &t->ev_wd, &t->ev_wl[1], &t->ev_wl[2], &t->ev_wl[3]}) OZ_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        for (int l = 0; l < 6; ++l) T_ALLOC(t->bnb2[l], (size_t)OZ_BNB_MAX_RB * t->Co[l]);
        OZ_HIP(hipStreamSynchronize(t->s));
        return OZ_OK;
    }();
    if (rc) return fail(rc);
    *out = t;
    return OZ_OK;
}

OZ_API int oz_trainer_destroy(oz_trainer* t) { delete t; return OZ_OK; }

// sticky range flag of the f16x2 mode (an activation above the fp16 range): reported at the synchronising calls
static int t_check_range(oz_trainer* t) {
    if (!t->h2 || !t->h2flag) return OZ_OK;
    int f = 0;
    OZ_HIP(hipMemcpyAsync(&f, t->h2flag, sizeof(int), hipMemcpyDeviceToHost, t->s));
    OZ_HIP(hipStreamSynchronize(t->s));
    if (f) {
        // reported once: the flag is cleared, so the trainer is usable again after set_weights reloads good weights (the step that raised it is invalid)
        OZ_HIP(hipMemsetAsync(t->h2flag, 0, sizeof(int), t->s));
        OZ_HIP(hipStreamSynchronize(t->s));
        if (f & 2) oz_set_error("a weight or gradient tensor holds Inf / NaN in the trainer's f16x2 mode (the optimisation diverged): the step is invalid");
        else oz_set_error("an activation exceeded the fp16 range (65504) in the trainer's f16x2 mode: the step is invalid; use precision 0 (f32)");
        return OZ_ERR_STATE;
    }
    return OZ_OK;
}

OZ_API int oz_trainer_set_precision(oz_trainer* t, int mode) {
    OZ_REQUIRE(t && (mode == 0 || mode == 1), "oz_trainer_set_precision: mode 0 (f32) or 1 (f16x2)");
    T_LOCK(t);
    OZ_REQUIRE(t->C % 256 == 0 || mode == 0, "oz_trainer_set_precision: f16x2 needs channels %% 256 == 0 (got %d)", t->C);
    OZ_HIP(hipSetDevice(t->device));
    OZ_HIP(hipStreamSynchronize(t->s));
    if (mode == 1 && !t->wmax) {
        const int C = t->C;
        const size_t wq = (size_t)C * 9 * C / 4;             // uint4 units of a 3x3 kernel in the h2 layout (4 B per value)
        for (int l = 1; l < 4; ++l) {
            T_ALLOC(t->Wh[l], wq); T_ALLOC(t->Whd[l], wq);
            T_ALLOC(t->wscale[l], C); T_ALLOC(t->dscale[l], C);
            T_ALLOC(t->a_h2[l - 1], (size_t)t->Bmax * t->P_[l - 1] * C / 4);
            T_ALLOC(t->dz_h2[l], (size_t)t->Bmax * t->Hz[l] * t->Hz[l] * C / 4);      // zeroed: the border stays zero
            const size_t noct = (size_t)(t->Bmax + 7) / 8;
            T_ALLOC(t->xt_oct[l], noct * (t->Hout[l] + 2) * WH_XW * 2 * C);
            T_ALLOC(t->zt_oct[l], noct * t->Hout[l] * WH_ZW * 2 * C);
        }
        T_ALLOC(t->wmax, 4); T_ALLOC(t->dzmax, 8); T_ALLOC(t->h2flag, 1);
        OZ_HIP(hipStreamSynchronize(t->s));
    }
    t->h2 = mode;
    t->dirty = true;
    return OZ_OK;
}

OZ_API int oz_trainer_set_weight(oz_trainer* t, int index, const float* data, int64_t nelem) {
    OZ_REQUIRE(t && data && index >= 0 && index < 40, "oz_trainer_set_weight: bad argument");
    T_LOCK(t);
    OZ_REQUIRE(nelem == t->size[index], "oz_trainer_set_weight: weight %d has %lld elements, got %lld", index, (long long)t->size[index], (long long)nelem);
    OZ_HIP(hipSetDevice(t->device));
    OZ_HIP(hipMemcpyAsync(t->param(index), data, nelem * sizeof(float), hipMemcpyHostToDevice, t->s));
    OZ_HIP(hipStreamSynchronize(t->s));
    t->dirty = true;
    return OZ_OK;
}
OZ_API int oz_trainer_get_weight(oz_trainer* t, int index, float* data, int64_t nelem) {
    OZ_REQUIRE(t && data && index >= 0 && index < 40 && nelem == t->size[index], "oz_trainer_get_weight: bad argument");
    T_LOCK(t);
    OZ_HIP(hipSetDevice(t->device));
    OZ_HIP(hipMemcpyAsync(data, t->param(index), nelem * sizeof(float), hipMemcpyDeviceToHost, t->s));
    OZ_HIP(hipStreamSynchronize(t->s));
    return OZ_OK;
}
OZ_API int oz_trainer_get_grad(oz_trainer* t, int index, float* data, int64_t nelem) {
    OZ_REQUIRE(t && data && index >= 0 && index < 40 && nelem == t->size[index] && t->toff[index] >= 0, "oz_trainer_get_grad: bad argument");
    T_LOCK(t);
    OZ_HIP(hipSetDevice(t->device));
    OZ_HIP(hipMemcpyAsync(data, t->grad(index), nelem * sizeof(float), hipMemcpyDeviceToHost, t->s));
    OZ_HIP(hipStreamSynchronize(t->s));
    return OZ_OK;
}
OZ_API int oz_trainer_grad_arena(oz_trainer* t, void** device_ptr, int64_t* nelem) {
    OZ_REQUIRE(t && device_ptr && nelem, "oz_trainer_grad_arena: bad argument");
    T_LOCK(t);
    *device_ptr = t->G; *nelem = t->total;
    return OZ_OK;
}
OZ_API int oz_trainer_arena_size(int n, int channels, int in_channels, int64_t* nelem) {
    OZ_REQUIRE(nelem, "oz_trainer_arena_size: bad argument");
    int64_t size[40], toff[40];
    const int64_t tot = t_layout(n, channels, in_channels, size, toff);
    *nelem = tot;
    return OZ_OK;
}
OZ_API int oz_trainer_sync(oz_trainer* t) {
    OZ_REQUIRE(t, "oz_trainer_sync: NULL");
    T_LOCK(t);
    OZ_HIP(hipSetDevice(t->device));
    OZ_HIP(hipStreamSynchronize(t->s));
    return OZ_OK;
}
OZ_API int oz_trainer_step_count(oz_trainer* t, int64_t* step) {
    OZ_REQUIRE(t && step, "oz_trainer_step_count: bad argument");
    T_LOCK(t);
    *step = t->step;
    return OZ_OK;
}

// The weights changed (optimiser step, set_weight): rebuild the GEMM operands derived from them.  With the second stream
// available the eight launches run THERE, after everything already queued on the main stream (the optimiser step that wrote
// the weights, the previous step's readers of Wt / Wd), and the main stream only waits where it first needs them -- the
// forward operands before conv2's GEMM, the data-gradient operands before conv4's dgrad -- so the uploads, conv1 and its BN
// run beside them instead of behind them (~70 us of a 1.4 ms step at batch 32).
static int t_refresh(oz_trainer* t) {
    const int C = t->C, F = (t->n - 4) * (t->n - 4) * C;
    const int Ks[6] = {0, 9 * C, 9 * C, 9 * C, F, 1024}, Ns[6] = {0, C, C, C, 1024, 512};
    hipStream_t r = t->s;
    if (t->overlap) {
        OZ_HIP(hipEventRecord(t->ev_pre, t->s));
        OZ_HIP(hipStreamWaitEvent(t->s2, t->ev_pre, 0));
        r = t->s2;
    }
    const unsigned h2_blocks = (unsigned)(((long long)C * (9 * C / 8) + 255) / 256);
    if (t->h2) {         // f16x2: per-tensor maxima -> power-of-two scales -> the h2 forward operands, straight from the masters
        OZ_HIP(hipMemsetAsync(t->wmax, 0, 3 * sizeof(unsigned), r));
        AbsMaxArgs am;
        for (int l = 1; l < 4; ++l) { am.p[l - 1] = t->param(6 * l); am.n[l - 1] = 9LL * C * C; }
        hipLaunchKernelGGL(k_t_absmax, dim3(192, 3), dim3(256), 0, r, am, t->wmax);      // (12 x 16-byte loads per thread cover 9 x 512 x 512 floats in one trip)
        for (int l = 1; l < 4; ++l) {
            hipLaunchKernelGGL(k_t_w_to_h2<0>, dim3(h2_blocks), dim3(256), 0, r, t->param(6 * l), C, C, t->wmax + (l - 1), t->Wh[l], t->wscale[l], t->h2flag);
            if (t->overlap) { OZ_HIP(hipEventRecord(t->ev_wl[l], r)); t->wait_wl[l] = true; }      // conv2's GEMM waits for ITS operand only (round 5: it waited
        }                                                                                          // for all five forward operands, 64 us of an idle main stream per step)
        OZ_HIP(hipGetLastError());
    }
    for (int l = t->h2 ? 4 : 1; l < 6; ++l) {      // fp32 forward operands (the dense layers; the 3x3 layers too in f32 mode)
        hipLaunchKernelGGL(k_t_transpose, dim3((Ns[l] + 31) / 32, (Ks[l] + 31) / 32), dim3(256), 0, r, t->param(6 * l), Ks[l], Ns[l], t->Wt[l]);
        OZ_HIP(hipGetLastError());
    }
    if (t->overlap) OZ_HIP(hipEventRecord(t->ev_wt, r));
    for (int l = 1; l < 4; ++l) {
        if (t->h2)
            hipLaunchKernelGGL(k_t_w_to_h2<1>, dim3(h2_blocks), dim3(256), 0, r, t->param(6 * l), C, C, t->wmax + (l - 1), t->Whd[l], (float*)nullptr, t->h2flag);
        else {
            const long long cnt = 9LL * C * C;
            hipLaunchKernelGGL(k_t_dgrad_operand, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, r, t->param(6 * l), C, C, t->Wd[l]);
        }
        OZ_HIP(hipGetLastError());
    }
    if (t->overlap) OZ_HIP(hipEventRecord(t->ev_wd, r));
    t->wait_wt = t->wait_wd = t->overlap;
    t->dirty = false;
    return OZ_OK;
}

template <int MODE>
static int t_reduce(oz_trainer* t, RedArgs r) {
    hipLaunchKernelGGL(k_t_colreduce<MODE>, dim3((r.C / 4 + 63) / 64, RED_S), dim3(256), 0, t->s, r, t->d_count, t->partial);
    OZ_HIP(hipGetLastError());
    return OZ_OK;
}

// BN (training mode) + ReLU (+ dropout) of layer l: z[l] -> a[l]
static int t_bn_forward(oz_trainer* t, int l, int B) {
    const int Cc = t->Co[l], P = t->P_[l];
    if ((long long)B * P <= OZ_BN_FUSED_MAX_ROWS) {      // small batch: the whole layer in one launch (oz_train_fused.h)
        if ((long long)B * P <= 32 * OZ_BN_RL)
            hipLaunchKernelGGL(k_t_bn_fwd_fused<32>, dim3(Cc / OZ_BN_COLS), dim3(1024), 0, t->s, t->z[l], t->a[l], t->d_count, P, Cc, t->param(6 * l + 2),
                               t->param(6 * l + 3), t->mean[l], t->rstd[l], t->stats[6 * l + 4], t->stats[6 * l + 5], t->stats_new[6 * l + 4],
                               t->stats_new[6 * l + 5], t->mom, l < 4 ? 1 : 0, l >= 4 ? t->rate : 0.f, t->seed, (uint64_t)t->step, l - 4);
        else
            hipLaunchKernelGGL(k_t_bn_fwd_fused<64>, dim3(Cc / OZ_BN_COLS), dim3(1024), 0, t->s, t->z[l], t->a[l], t->d_count, P, Cc, t->param(6 * l + 2),
                               t->param(6 * l + 3), t->mean[l], t->rstd[l], t->stats[6 * l + 4], t->stats[6 * l + 5], t->stats_new[6 * l + 4],
                               t->stats_new[6 * l + 5], t->mom, l < 4 ? 1 : 0, l >= 4 ? t->rate : 0.f, t->seed, (uint64_t)t->step, l - 4);
        OZ_HIP(hipGetLastError());
        return OZ_OK;
    }
    RedArgs r = {}; r.x = t->z[l]; r.P = P; r.C = Cc;
    if (int rc = t_reduce<0>(t, r)) return rc;
    hipLaunchKernelGGL(k_t_fin_mean, dim3((Cc + 255) / 256), dim3(256), 0, t->s, t->partial, t->d_count, P, Cc, t->mean[l]);
    r.mean = t->mean[l];
    if (int rc = t_reduce<1>(t, r)) return rc;
    hipLaunchKernelGGL(k_t_fin_var, dim3((Cc + 255) / 256), dim3(256), 0, t->s, t->partial, t->d_count, P, Cc, t->mean[l], t->rstd[l],
                       t->stats[6 * l + 4], t->stats[6 * l + 5], t->stats_new[6 * l + 4], t->stats_new[6 * l + 5], t->mom, l < 4 ? 1 : 0);
    const long long total = (long long)B * P * Cc;
    hipLaunchKernelGGL(k_t_bn_fwd, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, t->s, t->z[l], t->mean[l], t->rstd[l], t->param(6 * l + 2),
                       t->param(6 * l + 3), t->a[l], t->d_count, P, Cc, l >= 4 ? t->rate : 0.f, t->seed, (uint64_t)t->step, l - 4);
    OZ_HIP(hipGetLastError());
    return OZ_OK;
}

// forward + backward of the batch staged in d_own / d_opp / d_pit / d_zt / d_count (B boards): everything is enqueued on the
// trainer's streams, nothing is waited for; gradients land in the arena, the batch-mean losses in t->losses
static int t_forward_backward_async(oz_trainer* t, int B) {
    hipStream_t s = t->s;
    const int n = t->n, C = t->C, A = n * n, F = (n - 4) * (n - 4) * C;
    if (t->dirty) if (int rc = t_refresh(t)) return rc;

    // ---- forward
    { const long long tot = (long long)B * A * (C / 4);
      hipLaunchKernelGGL(k_t_conv1_fwd, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, t->d_own, t->d_opp, t->d_count, n, C, t->cin,
                         t->param(0), t->param(1), t->z[0]);
      OZ_HIP(hipGetLastError()); }
    if (int rc = t_bn_forward(t, 0, B)) return rc;
    const int Hin[6] = {n, n, n, n - 2, 1, 1}, pad[6] = {1, 1, 0, 0, 0, 0}, Cin[6] = {t->cin, C, C, C, F, 1024}, taps[6] = {9, 9, 9, 9, 1, 1};
    if (t->wait_wt && !t->h2) { OZ_HIP(hipStreamWaitEvent(s, t->ev_wt, 0)); t->wait_wt = false; }
    for (int l = 1; l < 6; ++l) {
        if (t->wait_wt && l == 4) { OZ_HIP(hipStreamWaitEvent(s, t->ev_wt, 0)); t->wait_wt = false; }      // f16x2: the fp32 operands of the dense layers
        if (t->h2 && l < 4) {      // f16x2: the previous layer's activation in the h2 layout, then the GEMM on the fp16 matrix cores
            const long long thr = (long long)B * t->P_[l - 1] * (C / 8);
            hipLaunchKernelGGL(k_t_act_to_h2, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, s, t->a[l - 1], t->d_count, t->P_[l - 1], C, t->a_h2[l - 1], t->h2flag);
            if (t->wait_wl[l]) { OZ_HIP(hipStreamWaitEvent(s, t->ev_wl[l], 0)); t->wait_wl[l] = false; }
            if (int rc = oz_gemm_h2_launch(t->a_h2[l - 1], t->Wh[l], t->wscale[l], t->param(6 * l + 1), t->z[l], t->d_count, B, Hin[l], t->Hout[l], pad[l],
                                           Cin[l], 9, t->Co[l], s, t->gpartial, t->gpartial_floats, t->zeros, t->h2flag)) return rc;
        } else
        if (int rc = oz_gemm_f32_launch(t->a[l - 1], t->Wt[l], t->ones, t->param(6 * l + 1), t->z[l], t->d_count, B, Hin[l], t->Hout[l], pad[l],
                                        Cin[l], taps[l], t->Co[l], 0, s, t->gpartial, (t->split_mask & 1) ? t->gpartial_floats : 0)) return rc;
        if (int rc = t_bn_forward(t, l, B)) return rc;
    }
    hipLaunchKernelGGL(k_t_heads, dim3(B), dim3(64), 0, s, t->a[5], t->d_count, n, t->param(36), t->param(37), t->param(38), t->param(39),
                       t->d_pit, t->d_zt, t->p, t->v, t->dlogit, t->dvpre, t->loss);
    OZ_HIP(hipGetLastError());

    // ---- backward
    hipLaunchKernelGGL(k_t_heads_wgrad, dim3(512), dim3(128), 0, s, t->a[5], t->dlogit, t->dvpre, t->d_count, A, t->grad(36), t->grad(38));
    hipLaunchKernelGGL(k_t_heads_bias, dim3(1), dim3(128), 0, s, t->dlogit, t->dvpre, t->loss, t->d_count, A, t->grad(37), t->grad(39), t->losses,
                       t->h2 ? t->dzmax : (unsigned*)nullptr);
    hipLaunchKernelGGL(k_t_heads_dgrad, dim3((unsigned)(((long long)B * 512 + 255) / 256)), dim3(256), 0, s, t->dlogit, t->dvpre, t->param(36),
                       t->param(38), t->d_count, A, t->dA[1]);
    OZ_HIP(hipGetLastError());
    int cur = 1;                                   // dA[cur] = gradient wrt a[l]
    for (int l = 5; l >= 0; --l) {
        bool have_dzmax = false;
        const int Cc = t->Co[l], P = t->P_[l];
        const float post = l >= 4 && t->rate > 0.f ? 1.0f / (1.0f - t->rate) : 1.0f;
        if ((long long)B * P > OZ_BNB_MIN_ROWS && (long long)B * P <= OZ_BNB_MAX_ROWS) {
            // partial sums over row splits, then one launch that finishes the sums and writes dz (oz_train_fused.h)
            const long long M = (long long)B * P;
            const int Q = Cc / 4, lpr = Q < 64 ? Q : 64, rpp = 256 / lpr;
            int RS = 16; while (RS < OZ_BNB_MAX_RB && M > (long long)RS * rpp * 8) RS *= 2;          // ~8 rows per thread
            RedArgs r = {}; r.x = t->dA[cur]; r.a = t->a[l]; r.z = t->z[l]; r.mean = t->mean[l]; r.rstd = t->rstd[l]; r.post_scale = post; r.P = P; r.C = Cc;
            hipLaunchKernelGGL(k_t_colreduce<2>, dim3((Q + 63) / 64, RS), dim3(256), 0, s, r, t->d_count, t->partial);
            hipLaunchKernelGGL(k_t_bnb_apply, dim3((Q + 63) / 64, RS), dim3(256), 0, s, t->dA[cur], t->a[l], t->z[l], t->mean[l], t->rstd[l], t->param(6 * l + 2),
                               post, t->d_count, t->Hout[l], Cc, t->Hz[l], t->zoff[l], t->partial, RS, t->dz[l], t->grad(6 * l + 2), t->grad(6 * l + 3), t->bnb2[l],
                               t->h2 ? t->dzmax + l : (unsigned*)nullptr);
            have_dzmax = t->h2 != 0;
            // the bias gradient (column sums of dz) is off the dgrad chain: finished beside it
            hipStream_t sb = s;
            if (t->overlap && l > 0) {
                OZ_HIP(hipEventRecord(t->ev_dz[l], s));
                OZ_HIP(hipStreamWaitEvent(t->s2, t->ev_dz[l], 0));
                sb = t->s2;
            }
            hipLaunchKernelGGL(k_t_sum_partials, dim3((unsigned)((Cc + 255) / 256)), dim3(256), 0, sb, t->bnb2[l], RS, (long long)Cc, t->grad(6 * l + 1));
            OZ_HIP(hipGetLastError());
        } else if ((long long)B * P <= OZ_BN_FUSED_MAX_ROWS) {  // small batch: BN backward, dgamma / dbeta / bias gradient in one launch
            hipLaunchKernelGGL(k_t_bn_bwd_fused, dim3(Cc / OZ_BN_COLS), dim3(1024), 0, s, t->dA[cur], t->a[l], t->z[l], t->mean[l], t->rstd[l],
                               t->param(6 * l + 2), post, t->d_count, t->Hout[l], Cc, t->Hz[l], t->zoff[l], t->dz[l], t->grad(6 * l + 2),
                               t->grad(6 * l + 3), t->grad(6 * l + 1), t->h2 ? t->dzmax + l : (unsigned*)nullptr);
            have_dzmax = t->h2 != 0;
            OZ_HIP(hipGetLastError());
        } else {
        RedArgs r = {}; r.x = t->dA[cur]; r.a = t->a[l]; r.z = t->z[l]; r.mean = t->mean[l]; r.rstd = t->rstd[l]; r.post_scale = post; r.P = P; r.C = Cc;
        if (int rc = t_reduce<2>(t, r)) return rc;
        hipLaunchKernelGGL(k_t_fin_bnbwd, dim3((Cc + 255) / 256), dim3(256), 0, s, t->partial, Cc, t->grad(6 * l + 2), t->grad(6 * l + 3), t->sums);
        const long long total = (long long)B * P * Cc;
        hipLaunchKernelGGL(k_t_bn_bwd, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s, t->dA[cur], t->a[l], t->z[l], t->mean[l], t->rstd[l],
                           t->param(6 * l + 2), t->sums, post, t->d_count, t->Hout[l], Cc, t->Hz[l], t->zoff[l], t->dz[l]);
        OZ_HIP(hipGetLastError());
        // bias gradient = column sums of dz (mathematically 0 behind a training-mode BN; computed like autograd would)
        RedArgs rb = {}; rb.x = t->dz[l]; rb.P = P; rb.C = Cc; rb.Hout = t->Hout[l]; rb.Hz = t->Hz[l]; rb.zoff = t->zoff[l];
        if (int rc = t_reduce<3>(t, rb)) return rc;
        hipLaunchKernelGGL(k_t_fin_colsum, dim3((Cc + 255) / 256), dim3(256), 0, s, t->partial, Cc, t->grad(6 * l + 1));
        OZ_HIP(hipGetLastError());
        }
        // weight gradient
        if (l == 0) {
            int S1 = 8;                                    // row splits: ~16 rows per thread (each a dependent ~1 us load at small batch), at most RED_S (the partial buffer's capacity)
            while (S1 < RED_S && (long long)B * A > 16LL * S1) S1 *= 2;
            if (n == 8) hipLaunchKernelGGL(k_t_conv1_wgrad<8>, dim3((C + 255) / 256, S1), dim3(256), 0, s, t->d_own, t->d_opp, t->d_count, C, t->cin, t->dz[0], t->partial, S1);
            else hipLaunchKernelGGL(k_t_conv1_wgrad<6>, dim3((C + 255) / 256, S1), dim3(256), 0, s, t->d_own, t->d_opp, t->d_count, C, t->cin, t->dz[0], t->partial, S1);
            const long long cnt = 9LL * t->cin * C;
            hipLaunchKernelGGL(k_t_sum_partials, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, t->partial, S1, cnt, t->grad(0));
            OZ_HIP(hipGetLastError());
        } else {
            WgradGeom g; g.Hin = Hin[l]; g.Hout = t->Hout[l]; g.pad = pad[l]; g.Cin = Cin[l]; g.Cout = Cc; g.taps = taps[l]; g.Hz = t->Hz[l]; g.zoff = t->zoff[l];
            const int wblocks = taps[l] * (Cin[l] / 128) * (Cc / 128);
            const long long wcount = (long long)taps[l] * Cin[l] * Cc, wtiles = ((long long)B * P + 31) / 32;
            int msplit = 1;
            while (msplit < 16 && wblocks * msplit < 512 && wtiles / (msplit * 2) >= 8 && wcount * msplit * 2 <= t->gpartial_floats) msplit *= 2;
            // (round 5, measured and removed: 3 splits instead of 4 so that 144 tiles x splits stays below 512 blocks -- the launch got SLOWER, 147 -> 227 us on
            //  conv2: more than two of these 40 KB-LDS blocks share a CU, 576 blocks are one round)
            // the weight gradient only needs a[l - 1] and dz[l]: it runs on the second stream beside the data-gradient chain
            // (dgrad -> BN backward of the layer below -> ...), whose small-batch launches leave most CUs idle
            hipStream_t sw = s;
            float* wp = t->gpartial;
            if (t->overlap) {
                OZ_HIP(hipEventRecord(t->ev_dz[l], s));
                OZ_HIP(hipStreamWaitEvent(t->s2, t->ev_dz[l], 0));
                sw = t->s2; wp = t->wpartial;
            }
            // (measured on one MI355X, 8x8 / 512 filters: the board-resident kernel wins from batch 256 on -- 6.01 vs 6.36 ms per step, 17.0 vs
            //  19.9 at 1024 -- and loses 3-4 % at 32 .. 128, where the tap-per-block kernel's 144 x 4 short blocks finish sooner)
            if (taps[l] == 9 && t->h2 && have_dzmax && l >= 1 && B >= WH_MIN_BATCH && Cin[l] % WH_CI == 0 && Cc % WH_CO == 0) {
                // f16x2: octet images of a[l - 1] and of the scaled dz[l], then the MFMA kernel on the fp16 matrix cores (three products per fp32 product)
                const int noct = (B + 7) / 8, XR = t->Hout[l] + 2;
                const int tiles = (Cin[l] / WH_CI) * (Cc / WH_CO);
                msplit = 1;
                while (msplit < 32 && tiles * msplit < 256 && msplit * 2 <= noct && wcount * msplit * 2 <= t->gpartial_floats) msplit *= 2;
                const long long xthr = (long long)noct * XR * WH_XW * (Cin[l] / 4), zthr = (long long)noct * t->Hout[l] * WH_ZW * (Cc / 4);
                hipLaunchKernelGGL(k_t_x_octets, dim3((unsigned)((xthr + 255) / 256)), dim3(256), 0, sw, t->a[l - 1], t->d_count, Hin[l], pad[l], Cin[l], t->xt_oct[l]);
                hipLaunchKernelGGL(k_t_z_octets, dim3((unsigned)((zthr + 255) / 256)), dim3(256), 0, sw, t->dz[l], t->d_count, t->Hout[l], t->Hz[l], t->zoff[l], Cc,
                                   t->dzmax + l, t->zt_oct[l], t->h2flag);
                if (!t->wh_attr) {
                    OZ_HIP(hipFuncSetAttribute((const void*)k_wgrad_h2, hipFuncAttributeMaxDynamicSharedMemorySize, WH_LDS));
                    t->wh_attr = true;
                }
                WhGeom wg; wg.XR = XR; wg.Hout = t->Hout[l]; wg.Cin = Cin[l]; wg.Cout = Cc;
                hipLaunchKernelGGL(k_wgrad_h2, dim3(tiles, msplit), dim3(512), WH_LDS, sw, t->xt_oct[l], t->zt_oct[l], t->d_count, wg, t->dzmax + l, t->grad(6 * l),
                                   msplit, wp, wcount);
            } else if (taps[l] == 9 && B >= 192 && Cin[l] % WC_CI == 0 && Cc % WC_CO == 0) {
                // board-resident kernel: (Cin / 64) x (Cout / 128) tiles, boards split over blockIdx.y until every CU has a block
                WconvGeom cg; cg.Hin = Hin[l]; cg.Hout = t->Hout[l]; cg.pad = pad[l]; cg.Cin = Cin[l]; cg.Cout = Cc; cg.Hz = t->Hz[l]; cg.zoff = t->zoff[l];
                const int tiles = (Cin[l] / WC_CI) * (Cc / WC_CO);
                msplit = 1;
                while (msplit < 16 && tiles * msplit < 256 && msplit * 2 <= B && wcount * msplit * 2 <= t->gpartial_floats) msplit *= 2;
                const int XW = cg.Hin + 2 * cg.pad;
                const size_t lds_bytes = 2 * sizeof(float) * ((size_t)XW * XW * WC_CI + (size_t)P * WC_CO);
                if (!t->wconv_attr) {
                    OZ_HIP(hipFuncSetAttribute((const void*)k_wgrad_conv, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
                    t->wconv_attr = true;
                }
                hipLaunchKernelGGL(k_wgrad_conv, dim3(tiles, msplit), dim3(512), lds_bytes, sw, t->a[l - 1], t->dz[l], t->d_count, cg, t->grad(6 * l), msplit,
                                   wp, wcount);
            } else {
                hipLaunchKernelGGL(k_wgrad_f32, dim3(wblocks, msplit), dim3(256), 0, sw, t->a[l - 1], t->dz[l], t->d_count, g, t->grad(6 * l), msplit,
                                   wp, wcount);
            }
            if (msplit > 1)
                hipLaunchKernelGGL(k_t_sum_partials, dim3((unsigned)((wcount + 255) / 256)), dim3(256), 0, sw, wp, msplit, wcount, t->grad(6 * l));
            OZ_HIP(hipGetLastError());
            // data gradient -> dA[cur ^ 1] = gradient wrt a[l - 1]
            if (l >= 4) {          // dense: dX = dZ . W^T; the Keras kernel [in][out] already is the [N = in][K = out] operand
                if (int rc = oz_gemm_f32_launch(t->dz[l], t->param(6 * l), t->ones, t->zeros, t->dA[cur ^ 1], t->d_count, B, 1, 1, 0, Cc, 1, Cin[l], 0, s, t->gpartial, (t->split_mask & 2) ? t->gpartial_floats : 0)) return rc;
            } else {               // 3x3 conv: conv of dz (zero-bordered for 'valid' layers) with the reversed, channel-swapped taps
                const int same = pad[l];
                if (t->wait_wd) { OZ_HIP(hipStreamWaitEvent(s, t->ev_wd, 0)); t->wait_wd = false; }
                if (have_dzmax) {  // f16x2: dz scaled into the fp16 range by its own maximum, h2 layout, same zero-bordered geometry
                    long long thr = (long long)B * P * (Cc / 8);
                    if (thr < Cin[l]) thr = Cin[l];
                    hipLaunchKernelGGL(k_t_dz_to_h2, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, s, t->dz[l], t->d_count, t->Hout[l], t->Hz[l], t->zoff[l], Cc,
                                       t->dzmax + l, t->wmax + (l - 1), t->dz_h2[l], t->dscale[l], Cin[l], t->h2flag);
                    if (int rc = oz_gemm_h2_launch(t->dz_h2[l], t->Whd[l], t->dscale[l], t->zeros, t->dA[cur ^ 1], t->d_count, B, t->Hz[l], Hin[l], same ? 1 : 0, Cc, 9,
                                                   Cin[l], s, t->gpartial, t->gpartial_floats, t->zeros, t->h2flag)) return rc;
                } else if (t->h2) {
                    oz_set_error("trainer f16x2: no |dz| maximum for layer %d", l);        // (every BN backward path of this mode leaves one)
                    return OZ_ERR_STATE;
                } else
                // (the non-zero core of the zero-bordered dz buffer: taps that only read the border are skipped at large batch)
                if (int rc = oz_gemm_f32_launch(t->dz[l], t->Wd[l], t->ones, t->zeros, t->dA[cur ^ 1], t->d_count, B, t->Hz[l], Hin[l], same ? 1 : 0, Cc, 9,
                                                Cin[l], 0, s, t->gpartial, (t->split_mask & 4) ? t->gpartial_floats : 0, 0, t->zoff[l],
                                                t->zoff[l] + t->Hout[l])) return rc;
            }
            cur ^= 1;
        }
    }
    if (t->overlap) {                              // every gradient is complete before the caller (Adam, all-reduce, get_grad) sees the arena
        OZ_HIP(hipEventRecord(t->ev_w, t->s2));
        OZ_HIP(hipStreamWaitEvent(s, t->ev_w, 0));
    }
    return OZ_OK;
}

OZ_API int oz_trainer_forward_backward(oz_trainer* t, const uint64_t* own, const uint64_t* opp, const float* pi_target, const float* z_target,
                                       int B, float* losses3) {
    OZ_REQUIRE(t && own && opp && pi_target && z_target, "oz_trainer_forward_backward: NULL argument");
    T_LOCK(t);
    OZ_REQUIRE(B >= 1 && B <= t->Bmax, "oz_trainer_forward_backward: batch %d outside [1, %d]", B, t->Bmax);
    OZ_HIP(hipSetDevice(t->device));
    hipStream_t s = t->s;
    const int A = t->n * t->n;
    {   // one upload (the call synchronises before it returns, so the pinned block is free again at the next call)
        unsigned char* h = t->h_in;
        memcpy(h + ((unsigned char*)t->d_own - t->d_in), own, B * sizeof(uint64_t));
        memcpy(h + ((unsigned char*)t->d_opp - t->d_in), opp, B * sizeof(uint64_t));
        memcpy(h + ((unsigned char*)t->d_pit - t->d_in), pi_target, (size_t)B * A * sizeof(float));
        memcpy(h + ((unsigned char*)t->d_zt - t->d_in), z_target, B * sizeof(float));
        memcpy(h + ((unsigned char*)t->d_count - t->d_in), &B, sizeof(int));
        OZ_HIP(hipMemcpyAsync(t->d_in, h, t->in_bytes, hipMemcpyHostToDevice, s));
    }
    if (int rc = t_forward_backward_async(t, B)) {
        hipStreamSynchronize(s);                   // the upload out of h_in may still be in flight: a retrying caller's next memcpy into it must not race with that DMA
        return rc;
    }
    float h[4];
    OZ_HIP(hipMemcpyAsync(h, t->losses, 3 * sizeof(float), hipMemcpyDeviceToHost, s));
    OZ_HIP(hipStreamSynchronize(s));
    if (losses3) { losses3[0] = h[0]; losses3[1] = h[1]; losses3[2] = h[2]; }
    return t_check_range(t);
}

// ---------------------------------------------------------------- HBM-resident data set: one upload per fit, one read-back per epoch
// keras Model.fit (Net/NNet.py:67) walks the examples in shuffled batches; here the examples live on the device for the
// whole fit, a step gathers its batch by index on the device, and the optimiser steps of an epoch are enqueued back to back
// (no host copy, no synchronisation per step); the sample-weighted loss sums are accumulated on the device.
__global__ void k_t_gather_batch(const uint64_t* __restrict__ ds_own, const uint64_t* __restrict__ ds_opp, const float* __restrict__ ds_pi,
                                 const float* __restrict__ ds_z, const int* __restrict__ order, int first, int B, int A,
                                 uint64_t* __restrict__ own, uint64_t* __restrict__ opp, float* __restrict__ pit, float* __restrict__ zt,
                                 int* __restrict__ d_count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *d_count = B;
    if (i >= B * A) return;
    const int b = i / A, a = i % A, src = order[first + b];
    pit[i] = ds_pi[(size_t)src * A + a];
    if (a == 0) { own[b] = ds_own[src]; opp[b] = ds_opp[src]; zt[b] = ds_z[src]; }
}
__global__ void k_t_acc_losses(const float* __restrict__ losses, int B, double* __restrict__ acc /*[4]: 3 weighted sums + samples*/) {
    if (threadIdx.x < 3) acc[threadIdx.x] += (double)losses[threadIdx.x] * (double)B;
    if (threadIdx.x == 3) acc[3] += (double)B;
}

OZ_API int oz_trainer_set_dataset(oz_trainer* t, const uint64_t* own, const uint64_t* opp, const float* pi_target, const float* z_target, int64_t N) {
    OZ_REQUIRE(t && own && opp && pi_target && z_target && N >= 1 && N < (1ll << 31), "oz_trainer_set_dataset: bad argument");
    T_LOCK(t);
    OZ_HIP(hipSetDevice(t->device));
    const int A = t->n * t->n;
    if (N > t->ds_cap) {
        OZ_HIP(hipStreamSynchronize(t->s));
        if (t->ds_own) { hipFree(t->ds_own); hipFree(t->ds_opp); hipFree(t->ds_pi); hipFree(t->ds_z); hipFree(t->ds_order); }
        t->ds_own = t->ds_opp = nullptr; t->ds_pi = t->ds_z = nullptr; t->ds_order = nullptr; t->ds_cap = 0;
        OZ_HIP(hipMalloc((void**)&t->ds_own, N * sizeof(uint64_t))); OZ_HIP(hipMalloc((void**)&t->ds_opp, N * sizeof(uint64_t)));
        OZ_HIP(hipMalloc((void**)&t->ds_pi, (size_t)N * A * sizeof(float))); OZ_HIP(hipMalloc((void**)&t->ds_z, N * sizeof(float)));
        OZ_HIP(hipMalloc((void**)&t->ds_order, N * sizeof(int)));
        t->ds_cap = N;
    }
    if (!t->ds_acc) OZ_HIP(hipMalloc((void**)&t->ds_acc, 4 * sizeof(double)));
    OZ_HIP(hipMemcpyAsync(t->ds_own, own, N * sizeof(uint64_t), hipMemcpyHostToDevice, t->s));
    OZ_HIP(hipMemcpyAsync(t->ds_opp, opp, N * sizeof(uint64_t), hipMemcpyHostToDevice, t->s));
    OZ_HIP(hipMemcpyAsync(t->ds_pi, pi_target, (size_t)N * A * sizeof(float), hipMemcpyHostToDevice, t->s));
    OZ_HIP(hipMemcpyAsync(t->ds_z, z_target, N * sizeof(float), hipMemcpyHostToDevice, t->s));
    OZ_HIP(hipStreamSynchronize(t->s));                       // the host arrays may go away
    t->ds_n = N;
    return OZ_OK;
}

static int t_apply_locked(oz_trainer* t);

OZ_API int oz_trainer_fit_epoch(oz_trainer* t, const int32_t* order, int64_t count, int batch, float* losses3) {
    OZ_REQUIRE(t && order && count >= 1, "oz_trainer_fit_epoch: bad argument");
    T_LOCK(t);
    OZ_REQUIRE(t->ds_n > 0 && count <= t->ds_n, "oz_trainer_fit_epoch: %lld indices but the resident data set holds %lld examples (oz_trainer_set_dataset first)",
               (long long)count, (long long)t->ds_n);
    OZ_REQUIRE(batch >= 1 && batch <= t->Bmax, "oz_trainer_fit_epoch: batch %d outside [1, %d]", batch, t->Bmax);
    for (int64_t i = 0; i < count; ++i) OZ_REQUIRE(order[i] >= 0 && order[i] < t->ds_n, "oz_trainer_fit_epoch: index %d outside the data set", order[i]);
    OZ_HIP(hipSetDevice(t->device));
    hipStream_t s = t->s;
    const int A = t->n * t->n;
    OZ_HIP(hipMemcpyAsync(t->ds_order, order, count * sizeof(int), hipMemcpyHostToDevice, s));
    OZ_HIP(hipMemsetAsync(t->ds_acc, 0, 4 * sizeof(double), s));
    for (int64_t first = 0; first < count; first += batch) {
        const int B = (int)(count - first < batch ? count - first : batch);            // the last batch may be short, as in keras
        hipLaunchKernelGGL(k_t_gather_batch, dim3((B * A + 255) / 256), dim3(256), 0, s, t->ds_own, t->ds_opp, t->ds_pi, t->ds_z, t->ds_order,
                           (int)first, B, A, t->d_own, t->d_opp, t->d_pit, t->d_zt, t->d_count);
        if (int rc = t_forward_backward_async(t, B)) return rc;
        hipLaunchKernelGGL(k_t_acc_losses, dim3(1), dim3(64), 0, s, t->losses, B, t->ds_acc);
        if (int rc = t_apply_locked(t)) return rc;
    }
    double h[4];
    OZ_HIP(hipMemcpyAsync(h, t->ds_acc, sizeof h, hipMemcpyDeviceToHost, s));
    OZ_HIP(hipStreamSynchronize(s));
    if (losses3) for (int k = 0; k < 3; ++k) losses3[k] = (float)(h[k] / (h[3] > 0 ? h[3] : 1.0));
    return t_check_range(t);
}

static int t_apply_locked(oz_trainer* t) {
    t->step += 1;
    const double b1t = pow(0.9, (double)t->step), b2t = pow(0.999, (double)t->step);
    const float lr_t = (float)((double)t->lr * sqrt(1.0 - b2t) / (1.0 - b1t));
    hipLaunchKernelGGL(k_t_adam, dim3((unsigned)((t->total + 255) / 256)), dim3(256), 0, t->s, t->P, t->G, t->M1, t->V2, (long long)t->total, lr_t, t->clip);
    OZ_HIP(hipGetLastError());
    for (int i = 0; i < 36; ++i)            // the staged moving statistics become the current ones (pointer swap, stream-ordered use)
        if (t->toff[i] < 0) { float* tmp = t->stats[i]; t->stats[i] = t->stats_new[i]; t->stats_new[i] = tmp; }
    t->dirty = true;
    // the operands derived from the new weights are rebuilt now, on the second stream, behind the optimiser step -- not at the next forward's
    // first launch: what the host does between two steps (the next batch's upload) no longer delays them
    if (t->overlap) return t_refresh(t);
    return OZ_OK;
}
OZ_API int oz_trainer_apply(oz_trainer* t) {
    OZ_REQUIRE(t, "oz_trainer_apply: NULL");
    T_LOCK(t);
    OZ_HIP(hipSetDevice(t->device));
    return t_apply_locked(t);
}

OZ_API int oz_trainer_get_activation(oz_trainer* t, int layer, int B, float* data, int64_t nelem) {
    OZ_REQUIRE(t && data && layer >= 0 && layer < 6 && B >= 1 && B <= t->Bmax, "oz_trainer_get_activation: bad argument");
    T_LOCK(t);
    OZ_REQUIRE(nelem == (int64_t)B * t->P_[layer] * t->Co[layer], "oz_trainer_get_activation: expected %lld elements", (long long)B * t->P_[layer] * t->Co[layer]);
    OZ_HIP(hipSetDevice(t->device));
    OZ_HIP(hipMemcpyAsync(data, t->a[layer], nelem * sizeof(float), hipMemcpyDeviceToHost, t->s));
    OZ_HIP(hipStreamSynchronize(t->s));
    return OZ_OK;
}

OZ_API int oz_trainer_outputs(oz_trainer* t, int B, float* p, float* v) {
    OZ_REQUIRE(t && B >= 1 && B <= t->Bmax, "oz_trainer_outputs: bad argument");
    T_LOCK(t);
    OZ_HIP(hipSetDevice(t->device));
    if (p) OZ_HIP(hipMemcpyAsync(p, t->p, (size_t)B * t->n * t->n * sizeof(float), hipMemcpyDeviceToHost, t->s));
    if (v) OZ_HIP(hipMemcpyAsync(v, t->v, B * sizeof(float), hipMemcpyDeviceToHost, t->s));
    OZ_HIP(hipStreamSynchronize(t->s));
    return OZ_OK;
}
