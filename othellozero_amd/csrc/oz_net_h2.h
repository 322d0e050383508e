// oz_net_h2.h -- "f32 via 2 x fp16 split" convolution kernels (included by oz_net.hip).
//
// Why: OthelloNN is fp32 and the fp32 matrix cores peak at 157 TFLOP/s (1/16 of the 16-bit MFMA rate).
// Every fp32 value x is carried as two fp16 planes  x = h1 + h2,  h1 = fp16(x), h2 = fp16(x - h1)
// (22 significand bits), and a product is evaluated as  a1*b1 + a1*b2 + a2*b1  on
// v_mfma_f32_32x32x16_f16 with fp32 accumulation: the dropped a2*b2 term and the split residuals are
// <= 2^-22 relative -- the same class as fp32 accumulation error over K = 4608 (measured: |d pi|, |d v|
// <= 1e-5 vs float64 in tests/test_gpu_parity.py, like the fp32 path).  3 MFMAs at 16x the fp32 rate.
// Range: weights are pre-scaled per layer by an exact power of two so max|w| ~ 2^9..2^10 (the inverse
// goes into the BN scale); activations (post-ReLU) must stay below 65504 -- the epilogue raises a
// sticky device flag otherwise and the host reports it (use precision f32 for such a net), never a
// silent wrong answer.
//
// Storage ("h2 layout"): for a row (pixel or output channel) every 8 consecutive k are one 32-byte
// group  [h1 x 8][h2 x 8]; a row of K values is K/8 groups = 4*K bytes (same footprint as fp32).
// A lane of a 32x32x16 MFMA needs 8 consecutive k of one plane = one aligned 16-byte read.
//
// Kernel: implicit GEMM, block tile 256 x 256 x 32, 512 threads = 8 waves (2 x 4), wave tile 128 x 64
// = 4 x 2 MFMA tiles (128 accumulator registers).  Both operand tiles (256 rows x 128 B) go global -> LDS
// with 16-byte LDS-DMA (global_load_lds_dwordx4, no staging registers, no ds_write pass): the LDS image is
// lane-linear per wave instruction (8 rows of 8 chunks), made bank-conflict-free by XOR-swizzling the
// 16-byte chunk index with (row>>1)&7 on the SOURCE address and on the read address; A rows are gathered
// per 3x3 tap, out-of-image taps read a zero line.  Two LDS buffers, one barrier per k-tile.
// Epilogue = BN scale/shift + ReLU, then fp32 rows or the h2 layout for the next layer (transposed through
// LDS so that global stores are 16-byte chunks of whole pixel rows).
#pragma once

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define H2_BM 256
#define H2_BN 256
#define H2_BK 32
#define H2_TILEB (H2_BM * 128)            // one operand tile: 256 rows x 128 B = 32 768 B
#define H2_LDS_BYTES (4 * H2_TILEB)       // [buf][A|B] = 131 072 B
#define H2_F16_MAX 65504.0f

struct H2Geom {
    int Hin, Hout, pad, Cin, taps;        // taps 9 (3x3 conv) or 1 (dense, Hin = Hout = 1)
    int N, K;                             // output channels, taps * Cin
    int out_h2;                           // 1: write the h2 layout, 0: write fp32 rows
};

__device__ __forceinline__ void h2_split(float x, _Float16& h1, _Float16& h2) {
    h1 = (_Float16)x;
    h2 = (_Float16)(x - (float)h1);
}

typedef const __attribute__((address_space(1))) void* h2_gptr;
typedef __attribute__((address_space(3))) void* h2_lptr;

// conv1 + plane unpack, output in the h2 layout: one thread per (row m, group of 8 channels)
__global__ __launch_bounds__(256) void k_conv1_h2(const uint64_t* __restrict__ own, const uint64_t* __restrict__ opp,
                                                  const int* __restrict__ d_count, int n, int C,
                                                  const float* __restrict__ W /*[9][2][C]*/, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, uint4* __restrict__ out, int* __restrict__ flag) {
    const int P = n * n, cg = C >> 3;
    const long long M = (long long)(*d_count) * P;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long m = idx / cg;
    if (m >= M) return;
    const int c8 = (int)(idx % cg) * 8;
    const int b = (int)(m / P), pix = (int)(m % P), y = pix / n, x = pix % n;
    const uint64_t o = own[b], p = opp[b];
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = y + ky - 1, ix = x + kx - 1;
            if (iy < 0 || iy >= n || ix < 0 || ix >= n) continue;
            const int sq = iy * 8 + ix;
            const float a0 = (float)((o >> sq) & 1), a1 = (float)((p >> sq) & 1);
            const float* w0 = W + (size_t)((ky * 3 + kx) * 2 + 0) * C + c8;
            const float* w1 = W + (size_t)((ky * 3 + kx) * 2 + 1) * C + c8;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(a0, w0[j], acc[j]);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(a1, w1[j], acc[j]);
        }
    f16x8 h1, h2;
    bool over = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = fmaxf(fmaf(acc[j], scale[c8 + j], shift[c8 + j]), 0.f);
        over |= v > H2_F16_MAX;
        _Float16 a, bb;
        h2_split(v, a, bb);
        h1[j] = a; h2[j] = bb;
    }
    if (over) atomicOr(flag, 1);
    uint4* dst = out + ((size_t)m * cg + (c8 >> 3)) * 2;
    dst[0] = *reinterpret_cast<uint4*>(&h1);
    dst[1] = *reinterpret_cast<uint4*>(&h2);
}

// out[M][N] = relu((A[M][K] . W[N][K]^T) * scale + shift); A and W in the h2 layout; M = *d_count * Hout^2.
// zero_line: >= 128 B of zeros in global memory (source of out-of-image taps and of rows beyond M).
__global__ __launch_bounds__(512, 2) void k_gemm_h2(const uint4* __restrict__ in, const uint4* __restrict__ Wh,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    void* __restrict__ out, const int* __restrict__ d_count, H2Geom g,
                                                    int num_mt, const uint4* __restrict__ zero_line, int* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nnt = g.N / H2_BN;
    const int id = blockIdx.x, xcd = id & 7, jb = id >> 3;
    const int mt = (jb / nnt) * 8 + xcd, nt = jb % nnt;
    const int P = g.Hout * g.Hout;
    const long long M = (long long)(*d_count) * P;
    if (mt >= num_mt || (long long)mt * H2_BM >= M) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                 // 2 x 4 waves, wave tile 128 rows x 64 cols
    const int rowq = g.Cin >> 2;                             // uint4 (16 B) units per input pixel row: Cin/8 groups * 2
    const int wrowq = g.K >> 2;                              // uint4 units per weight row

    // staging map: wave w, DMA instruction i (0..3) fills LDS rows (w*4+i)*8 .. +7; lane l -> row +(l>>3), physical
    // chunk l&7, which holds logical chunk (l&7) ^ ((row>>1)&7) of that row's 128-byte k-slice
    long long aidx[4];       // uint4 index of (row's input pixel at tap (0,0), logical chunk), or -1
    unsigned amask[4];
    unsigned bidx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        const long long m = (long long)mt * H2_BM + row;
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
        bidx[i] = (unsigned)(nt * H2_BN + row) * (unsigned)wrowq + (unsigned)lc;
    }
    const uint4* zsrc = zero_line + (lane & 7);

    auto stage = [&](int kt, int buf) {
        const int k0 = kt * H2_BK, tap = k0 / g.Cin, ci0 = k0 - tap * g.Cin;
        const long long toff = ((long long)(tap / 3) * g.Hin + (tap % 3)) * rowq + (ci0 >> 2);   // 32 ch = 8 uint4
        unsigned char* la = smem + (size_t)buf * 2 * H2_TILEB + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4* ga = ((amask[i] >> tap) & 1) ? in + (aidx[i] + toff) : zsrc;
            const uint4* gb = Wh + bidx[i] + (k0 >> 2);
            __builtin_amdgcn_global_load_lds((h2_gptr)ga, (h2_lptr)(la + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((h2_gptr)gb, (h2_lptr)(la + H2_TILEB + i * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = g.K / H2_BK;
    const int r32 = lane & 31, half = lane >> 5;
    const int swz = (r32 >> 1) & 7;                          // (row>>1)&7 of every row this lane reads (tile bases are multiples of 16)
    // byte offsets inside a 128-byte row of logical chunks c = 4s + 2*half + plane
    const int o00 = ((2 * half) ^ swz) * 16, o01 = ((2 * half + 1) ^ swz) * 16;
    const int o10 = ((4 + 2 * half) ^ swz) * 16, o11 = ((4 + 2 * half + 1) ^ swz) * 16;

    stage(0, 0);
    __syncthreads();                                         // drains the DMA (vmcnt(0)) and publishes the tile
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
        const unsigned char* At = smem + (size_t)buf * 2 * H2_TILEB + (wm * 128 + r32) * 128;
        const unsigned char* Bt = smem + (size_t)buf * 2 * H2_TILEB + H2_TILEB + (wn * 64 + r32) * 128;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int oh1 = s ? o10 : o00, oh2 = s ? o11 : o01;
            f16x8 b1[2], b2[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                b1[j] = *reinterpret_cast<const f16x8*>(Bt + j * 32 * 128 + oh1);
                b2[j] = *reinterpret_cast<const f16x8*>(Bt + j * 32 * 128 + oh2);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f16x8 a1 = *reinterpret_cast<const f16x8*>(At + i * 32 * 128 + oh1);
                const f16x8 a2 = *reinterpret_cast<const f16x8*>(At + i * 32 * 128 + oh2);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue.  C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    if (!g.out_h2) {
        float* o = reinterpret_cast<float*>(out);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = nt * H2_BN + wn * 64 + j * 32 + r32;
            const float sc = scale[col], sh = shift[col];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long m = (long long)mt * H2_BM + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (m < M) o[(size_t)m * g.N + col] = fmaxf(fmaf(acc[i][j][r], sc, sh), 0.f);
                }
        }
        return;
    }
    // h2 output: each wave transposes its 128 x 64 tile through its own 16 KB LDS slice in two halves of 64 rows:
    // slice[row][group(8)][plane(2)][8 halfs] = 256 B per row; then 16-byte chunks go out, 16 lanes per pixel row.
    bool over = false;
    _Float16* slice = reinterpret_cast<_Float16*>(smem + wave * 16384);
    uint4* o = reinterpret_cast<uint4*>(out);
    const int nq = g.N >> 2;                                // uint4 units per output pixel row (N/8 groups * 2)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = hh * 2 + ii;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int lc = j * 32 + r32;                 // column inside the wave tile
                const int col = nt * H2_BN + wn * 64 + lc;
                const float sc = scale[col], sh = shift[col];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;     // row inside the 64-row half
                    const float v = fmaxf(fmaf(acc[i][j][r], sc, sh), 0.f);
                    over |= v > H2_F16_MAX;
                    _Float16 h1, h2;
                    h2_split(v, h1, h2);
                    _Float16* p = slice + lr * 128 + (lc >> 3) * 16 + (lc & 7);
                    p[0] = h1; p[8] = h2;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): this wave's LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        // 64 rows x 256 B = 1024 chunks of 16 B; lane l takes chunks l, l+64, ...: 16 consecutive lanes = one row
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int q = c * 64 + lane, lr = q >> 4, cq = q & 15;
            const long long m = (long long)mt * H2_BM + wm * 128 + hh * 64 + lr;
            const uint4 val = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(slice) + lr * 256 + cq * 16);
            if (m < M) o[(size_t)m * nq + ((nt * H2_BN + wn * 64) >> 2) + cq] = val;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
    if (over) atomicOr(flag, 1);
}
