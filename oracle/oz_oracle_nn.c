/*
 * oz_oracle_nn.c -- CPU ORACLE (NN leg). TEST INFRASTRUCTURE ONLY.
 *
 * float32 CPU restatement of the OthelloNN inference graph
 * (Net/OthelloNN.py:42-56 as driven by NNetWrapper.predict, Net/NNet.py:70-87):
 * 4 x (Conv3x3 -> BN -> ReLU), Flatten (NHWC), 2 x (Dense -> BN -> ReLU),
 * softmax policy head, tanh value head; BN with moving statistics and
 * epsilon = 1e-3, Dropout = identity.  Used (a) as the batch-1 leaf evaluator of
 * the CPU baseline that bench.py times beside the GPU numbers and (b) as a
 * cross-check of oracle/nn_numpy.py (the float64 restatement the GPU kernels
 * are compared against).
 *
 * Parity status: the NN arithmetic lives in TensorFlow 2.3.1 / Keras 2.4.3
 * (requirements.txt:16,35), which is not under /root/reference and not
 * installed here, and the reference has no tests => NN parity is UNPINNED by
 * reference vectors; this file and nn_numpy.py restate the documented Keras
 * semantics and are cross-checked against each other and torch-CPU.
 *
 * Weight list order = keras Model.get_weights(): per conv/dense [kernel, bias],
 * per BN [gamma, beta, moving_mean, moving_variance]; 40 arrays for ONN.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))
#define BN_EPS 1e-3f
#define CB 64

static void bn_fold(const float* bias, const float* g, const float* b, const float* mu, const float* var, int C,
                    float* scale, float* shift) {
    for (int c = 0; c < C; ++c) {
        float s = g[c] / sqrtf(var[c] + BN_EPS);
        scale[c] = s;
        shift[c] = (bias[c] - mu[c]) * s + b[c];
    }
}

/* in [H][H][Cin] -> out [Ho][Ho][Cout], kernel [3][3][Cin][Cout] */
static void conv3x3_bn_relu(const float* in, int H, int Cin, const float* K, int Cout, int same,
                            const float* scale, const float* shift, float* out) {
    const int Ho = same ? H : H - 2, off = same ? -1 : 0;
    const int ncb = Cout / CB;
#pragma omp parallel for collapse(2) schedule(static)
    for (int cb = 0; cb < ncb; ++cb)
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Ho; ++x) {
                float acc[CB];
                for (int j = 0; j < CB; ++j) acc[j] = 0.f;
                for (int ky = 0; ky < 3; ++ky) {
                    int iy = y + ky + off;
                    if (iy < 0 || iy >= H) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        int ix = x + kx + off;
                        if (ix < 0 || ix >= H) continue;
                        const float* ip = in + (size_t)(iy * H + ix) * Cin;
                        const float* kp = K + (size_t)((ky * 3 + kx) * Cin) * Cout + cb * CB;
                        for (int ci = 0; ci < Cin; ++ci) {
                            const float a = ip[ci];
                            const float* kr = kp + (size_t)ci * Cout;
                            for (int j = 0; j < CB; ++j) acc[j] += a * kr[j];
                        }
                    }
                }
                float* op = out + (size_t)(y * Ho + x) * Cout + cb * CB;
                for (int j = 0; j < CB; ++j) {
                    float t = acc[j] * scale[cb * CB + j] + shift[cb * CB + j];
                    op[j] = t > 0.f ? t : 0.f;
                }
            }
}

static void dense(const float* in, int I, const float* W /*[I][O]*/, int O, float* out /* raw sums */) {
    const int nb = (O + CB - 1) / CB;
#pragma omp parallel for schedule(static)
    for (int ob = 0; ob < nb; ++ob) {
        int o0 = ob * CB, on = O - o0 < CB ? O - o0 : CB;
        float acc[CB];
        for (int j = 0; j < CB; ++j) acc[j] = 0.f;
        for (int i = 0; i < I; ++i) {
            const float a = in[i];
            const float* wr = W + (size_t)i * O + o0;
            for (int j = 0; j < on; ++j) acc[j] += a * wr[j];
        }
        for (int j = 0; j < on; ++j) out[o0 + j] = acc[j];
    }
}

ORC_API int orc_nn_num_weights(void) { return 40; }

/* boards: canonical bitboards (own = channel 0, opp = channel 1), bit r*8+c.
 * pi: [B][n*n], v: [B]. */
ORC_API void orc_nn_forward_f32(const float* const* W, int n, int C, const uint64_t* own, const uint64_t* opp, int B,
                                float* pi, float* v, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    const int F = (n - 4) * (n - 4) * C, A = n * n;
    float* x0 = (float*)malloc(sizeof(float) * (size_t)n * n * 2);
    float* a1 = (float*)malloc(sizeof(float) * (size_t)n * n * C);
    float* a2 = (float*)malloc(sizeof(float) * (size_t)n * n * C);
    float* a3 = (float*)malloc(sizeof(float) * (size_t)(n - 2) * (n - 2) * C);
    float* a4 = (float*)malloc(sizeof(float) * (size_t)F);
    float *f1 = (float*)malloc(sizeof(float) * 1024), *f2 = (float*)malloc(sizeof(float) * 512);
    float* sc[6]; float* sh[6];
    const int widths[6] = {C, C, C, C, 1024, 512};
    for (int l = 0; l < 6; ++l) {
        sc[l] = (float*)malloc(sizeof(float) * (size_t)widths[l]);
        sh[l] = (float*)malloc(sizeof(float) * (size_t)widths[l]);
        const float* const* w = W + 6 * l;
        bn_fold(w[1], w[2], w[3], w[4], w[5], widths[l], sc[l], sh[l]);
    }
    for (int b = 0; b < B; ++b) {
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) {
                x0[(r * n + c) * 2 + 0] = (float)((own[b] >> (r * 8 + c)) & 1);
                x0[(r * n + c) * 2 + 1] = (float)((opp[b] >> (r * 8 + c)) & 1);
            }
        /* conv1 has Cin = 2: reuse the generic loop (C % CB == 0 is required) */
        conv3x3_bn_relu(x0, n, 2, W[0], C, 1, sc[0], sh[0], a1);
        conv3x3_bn_relu(a1, n, C, W[6], C, 1, sc[1], sh[1], a2);
        conv3x3_bn_relu(a2, n, C, W[12], C, 0, sc[2], sh[2], a3);
        conv3x3_bn_relu(a3, n - 2, C, W[18], C, 0, sc[3], sh[3], a4);
        dense(a4, F, W[24], 1024, f1);
        for (int j = 0; j < 1024; ++j) { float t = f1[j] * sc[4][j] + sh[4][j]; f1[j] = t > 0.f ? t : 0.f; }
        dense(f1, 1024, W[30], 512, f2);
        for (int j = 0; j < 512; ++j) { float t = f2[j] * sc[5][j] + sh[5][j]; f2[j] = t > 0.f ? t : 0.f; }
        float logits[64];
        dense(f2, 512, W[36], A, logits);
        float mx = -INFINITY;
        for (int a = 0; a < A; ++a) { logits[a] += W[37][a]; if (logits[a] > mx) mx = logits[a]; }
        float s = 0.f;
        for (int a = 0; a < A; ++a) { logits[a] = expf(logits[a] - mx); s += logits[a]; }
        for (int a = 0; a < A; ++a) pi[(size_t)b * A + a] = logits[a] / s;
        float vv = 0.f;
        for (int i = 0; i < 512; ++i) vv += f2[i] * W[38][i];
        v[b] = tanhf(vv + W[39][0]);
    }
    for (int l = 0; l < 6; ++l) { free(sc[l]); free(sh[l]); }
    free(x0); free(a1); free(a2); free(a3); free(a4); free(f1); free(f2);
}

/* evaluator context + callback with the orc_eval_fn signature of oz_oracle.c */
typedef struct { const float* const* W; int C; int nthreads; long calls; } orc_nn_ctx;
ORC_API void orc_nn_eval_cb(void* ctx, uint64_t own, uint64_t opp, int n, float* pi, float* v) {
    orc_nn_ctx* c = (orc_nn_ctx*)ctx;
    c->calls++;
    orc_nn_forward_f32(c->W, n, c->C, &own, &opp, 1, pi, v, c->nthreads);
}
ORC_API int orc_nn_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
