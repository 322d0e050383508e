// oz_net_h2.h -- "f32 via 2 x fp16 split" convolution / dense kernels (included by oz_net.hip).
//
// Why: OthelloNN is fp32 and the fp32 matrix cores peak at 157 TFLOP/s (1/16 of the 16-bit MFMA rate).
// Every fp32 value x is carried as two fp16 planes  x = h1 + h2,  h1 = fp16(x), h2 = fp16(x - h1), and a product is
// evaluated as  a1*b1 + a1*b2 + a2*b1  on the fp16 matrix cores with fp32 accumulation.  3 MFMAs at 16x the fp32 rate.
// What the split keeps (fp16: 11-bit significand, normal down to 2^-14, subnormal step 2^-24):
//   |x| in [2^-3, 65504]: h1 and h2 are both normal -> 22 significant bits, error <= 2^-24 |x|;
//   |x| < 2^-3:           h2 falls into fp16 subnormals -> an ABSOLUTE error floor of 2^-25 per element (at 1e-3 about 15 bits
//                         survive, below 6e-5 h1 itself is subnormal);   |x| > 65504: not representable.
// So the error of an element is max(2^-24 |x|, 2^-25), and it is the job of the SCALING to place every tensor in that window
// (round 4; rounds 1-3 scaled the weights per layer by their single largest |w| and the activations not at all -- they happened to
// sit around 2^-6 .. 2^-3 for Keras-initialised networks, and nothing said what happens elsewhere):
//   activations: every channel c of every h2 tensor (conv1..conv4, fc1 outputs) carries an exact power of two 2^aexp[c], chosen at
//     oz_net_commit from the channel's largest |BN output| (BEFORE the ReLU: the post-ReLU maximum of a channel that hovers around
//     zero is not a scale) over a fixed calibration set of positions (all 3^9 patterns for conv1), so that this maximum lands in
//     [2^-3, 2^-2) = H2_ACT_TOP.  The power is folded into the producing layer's BN scale and shift (exact) and divided out of the
//     consuming layer's weight rows (exact) -- a diagonal rescaling between layers, the network function is unchanged;
//   weights: after that division every output column c gets its own 2^kexp[c] (its largest |w| -> [2^-3, 2^-2) = H2_W_TOP), folded
//     into the column's BN scale: a column of small weights compensated by its BN variance is carried like any other.
//   WHERE in the window, and why at the bottom.  With the maxima just below 2^-2 every element is carried with an ABSOLUTE error of
//     2^-25 <= 2^-23 of its channel's (column's) calibration maximum -- fp32's own relative precision for the elements near the
//     maximum, which dominate a dot product, and an error far below theirs for the small ones -- which is what an absolute tolerance
//     on (pi, v) needs.  And the small elements' residual planes then hold few significant bits: the matrix pipe's power, and
//     with it the sustained clock, follows the operand bits (round 3: zeroed residual planes ran 11 % faster).  Measured on one device
//     (tools/target_probe.py: the bench's 3640-position launch, conv3 / whole forward / max error against float64 on 256 rows):
//       maxima at 2^9 (activations) and 2^10 (weights), everything normal: 1143 us / 2169 us / 4.6e-8
//       activations at 2^-2:                                            1109      / 2114    / 4.6e-8
//       activations and weights at 2^-2 (the default):                  1089      / 2082    / 6.0e-8
//       activations at 2^-4 / 2^-7 (the maxima themselves lose bits):   1094 / 1082 conv3,    1.2e-7 / 1.0e-6
//     (the unscaled round-3 build: 1098 / 2093 / 9.4e-8 -- its activations sat at 2^-3 and below by accident.)
//   guards (sticky device flag, OZ_ERR_STATE at the next synchronising call -- never a silent wrong answer): HIGH side, an
//     activation above 65504 (2^18 above its channel's calibration maximum: non-finite arithmetic upstream, in practice); LOW side, a
//     pixel row whose largest scaled activation over ALL channels is non-zero and below 2^H2_LOW_GUARD = 2^-17, i.e. a position whose
//     whole row sits 2^15 or more below the calibration maxima (the calibration does not describe it), detected per 64-channel slice
//     in the producing epilogue with a per-row counter for the rare low slices (h2_low_report);
//   self-check (oz_net.hip, run_self_check): at every commit the calibration positions go through these kernels AND the exact-fp32
//     kernels; a difference above 8e-6 in (pi, v) fails the commit -- the screen for networks that amplify rounding (conditioning),
//     which no range argument covers.
// Measured: |d pi|, |d v| <= 1e-5 vs float64 on every tested network incl. small-activation, wide-weight-range and badly scaled BN
// cases (tests/test_gpu_parity.py::test_f16x2_scaling_*), like the fp32 path; 5e-8 on the bench's network.
//
// Storage ("h2 layout"): for a row (pixel or output channel) every 8 consecutive k are one 32-byte
// group  [h1 x 8][h2 x 8]; a row of K values is K/8 groups = 4*K bytes (same footprint as fp32).
// Weight rows are stored in the kernel's k-tile order: k' = (slice*taps + tap)*32 + c  for channel 32*slice + c,
// i.e. the 9 taps of one 32-channel slice are consecutive k-tiles, so the 9 shifted re-reads of the same
// activation rows happen back to back (L1/L2 hits instead of re-fetching past L2).
//
// Kernel k_gemm_h2: implicit GEMM, k-tile 32, on v_mfma_f32_16x16x32_f16 (one MFMA consumes the whole k-tile;
// lane l holds row/column l&15 and k-group l>>4 = one aligned 16-byte LDS read per plane).  Both operand tiles
// (rows x 128 B) go global -> LDS with 16-byte LDS-DMA (global_load_lds_dwordx4: no staging registers, no
// ds_write pass); the LDS image is lane-linear per wave instruction (8 rows of 8 chunks) and made
// bank-conflict-free by XOR-swizzling the 16-byte chunk index with h2_swz(row) on the SOURCE address and on the
// read address (SQ_LDS_BANK_CONFLICT measured); A rows are gathered per 3x3 tap, out-of-image taps read a
// zero line.  Two LDS stages.  Main loop of the 3x3 convolutions: ping-pong -- the two wave rows (one wave of each per SIMD)
// run one barrier apart, one in its MFMA cluster while the other issues its LDS reads and LDS-DMA pieces, counted vmcnt
// so that pieces stay in flight across barriers; four phases per k-tile on the 256-row tile (H2BigPP), two on the 192- and
// 128-row tiles (H2MidPP / H2LowPP, round 5) -- details at the loops.  The
// dense layers keep the simple loop: one barrier per k-tile, DMA interleaved with the MFMA stream (sched_group_barrier).
// Epilogue = BN scale/shift + ReLU, then fp32 rows or the h2 layout for the
// next layer (transposed through LDS so that global stores are 16-byte chunks of whole rows).
//
// What was measured on the way (conv2 size, 4096 leaves; kept here so nobody re-discovers it):
//   32x32x16 shape, 8 waves, 2 stages ................ 2.75 ms   (no DMA: 2.20 ms, no MFMA: 1.67 ms)
//   4-stage ring, k-tile 16, counted vmcnt ........... 3.11 ms   (twice the barriers / exposed LDS latency)
//   4 waves (one per SIMD), 128x128 wave tile, register double-buffered fragments, barrier mid-k-step ... 2.78 ms
//   all three: 1.77-1.89 GHz, 66-70 % matrix-pipe busy -> clock/power bound, not issue bound
//   16x16x32 shape (this kernel) ..................... 2.53 ms   (higher sustained clock on this shape)
//   + swizzle key for the 16x16x32 operand map ....... 2.40 ms   (LDS conflict cycles 50 % -> 3 %)
//   ablation of THIS kernel (round-1 experiment builds, since deleted): 2.47 ms full, 2.06 ms with
//   no DMA in the loop, 2.00 ms with no DMA and no barrier; A staged for 1 tap of 9 only: 2.36 ms.  So the floor of
//   this structure is LDS reads + MFMA at the power-limited clock (~620 TFLOP/s fp32-equivalent); LDS-DMA issue/wait
//   costs 16 %, barriers 3 %, and tap reuse for A alone would buy <= 4.5 %.
//   4-phase ping-pong loop (H2BigPP): conv2 2.41 -> 2.15 ms on the same device (-10 %), bench 842 k -> 915 k expansions/s.
//     In-kernel s_memtime stamps (round-1 diagnostic build, since deleted): clock 2.20-2.28 GHz; per interval (one wave row in its 24-MFMA
//     cluster, the other in its L section) ~531 cycles against 395 of MFMA issue (16.4 cycles per MFMA, measured alone in
//     tools/ubench/mfma_bank.hip, independent of the operands' VGPR banks); without any DMA 469.  Variants on that loop:
//     s_setprio around the cluster -2 %; DMA before the reads of an L section +-0; chained vs product-major MFMA order
//     +-0; accumulators in AGPRs (inline asm) slower; no vmcnt wait at all +-0 (the waits are free); 2 phases per tile
//     (half the barriers) +-0 on conv2 and slower on conv3/conv4; the last 4 / 8 MFMAs of a cluster
//     issued after its closing barrier (hand-over overlap) 5 / 7 % slower; a second copy of the loop without the zero-line
//     select for the pad-0 layers pushed spills into the loop (2.19 -> 2.53 ms); s_setprio 1 for wave row 1 over the whole loop
//     -1 %; the nt cache policy on the A pieces -5 %.  What is left is the issue cost of the 8
//     LDS-DMA pieces per wave and tile (~13 %) and barrier round trips.
//   The clock is the other half of the story (round 3, conv3 at 3640 leaves, same binary otherwise): with every h2 (residual) plane forced to
//   zero -- wrong results, same instruction stream -- the launch takes 1.004 instead of 1.126 ms (-11 %); with the low 3 or 5 mantissa bits of
//   the residuals masked (19 / 17 instead of 22 significant bits) 1.096 ms (-2.7 %, mostly small residuals that become exact zeros).  The matrix
//   pipe's power, and with it the sustained clock, depends on the operand bits; the product keeps the full 22-bit split.
//
// conv1 and conv2 no longer run as convolutions in this precision mode: the input planes are discrete, so both layers
// are evaluated from tables over the 3^9 neighbourhood patterns (k_lut_ids, k_lut_build, k_conv2_lut below; switch:
// oz_net_set_tables) -- the whole-bench rate went 0.94 M -> 1.55 M expansions/s and the dominant
// GEMM is conv3.  The GEMM kernel still serves conv2 when the tables are switched off and builds the tables at commit.
#pragma once
#include <type_traits>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

#define H2_BK 32
#define H2_F16_MAX 65504.0f
#define H2_ACT_TOP (-2)                      // per-channel calibration maxima of the h2 tensors land in [2^(H2_ACT_TOP - 1), 2^H2_ACT_TOP)
#define H2_W_TOP (-2)                        // every weight column's largest |w| lands in [2^(H2_W_TOP - 1), 2^H2_W_TOP)
#define H2_LOW_GUARD (-17)                   // log2 of the low-side guard's row threshold: 2^15 below the calibration maxima
#define H2_SELF_CHECK_LIMIT 8.0e-6           // oz_net_commit: largest |d pi|, |d v| between the f16x2 and the exact-fp32 kernels on the calibration positions

// tile configurations: NWM x NWN waves, each wave NTI x NTJ blocks of 32 x 32 (= 2 x 2 MFMA tiles of 16 x 16)
//   H2Big   (3x3 convolutions): 2 x 4 waves, wave tile 128 x 64 -> block 256 x 256, 512 threads, 128 KB LDS
//   H2Small (dense layers):     2 x 2 waves, wave tile  64 x 64 -> block 128 x 128, 256 threads,  64 KB LDS
template <int NWM, int NWN, int NTI, int NTJ, int NST = 2> struct H2Cfg {
    static constexpr int WM = NWM, WN = NWN, TI = NTI, TJ = NTJ;
    static constexpr int BM = NWM * NTI * 32, BN = NWN * NTJ * 32, NW = NWM * NWN, NT = NW * 64;
    static constexpr int STAGES = NST;    // LDS stages of the one-barrier loop (> 2: k-tiles prefetched STAGES - 1 ahead, counted vmcnt)
    static constexpr int TILEA = BM * 128, TILEB = BN * 128, BUF = TILEA + TILEB, LDS = NST * BUF;
    static constexpr int IA = BM / (8 * NW), IB = BN / (8 * NW);     // LDS-DMA instructions per wave per operand tile
    // h2 epilogue: each wave transposes PB 16-row blocks (x 64 channels) per pass through a private LDS slice
    static constexpr int PB = (NTI % 2 == 0 && NW * 4 * 16 * 256 <= LDS) ? 4 : 2, SLICE = PB * 16 * 256;
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile shape");
    static_assert(NW * SLICE <= LDS, "epilogue slices must fit in the staging buffers");
    static constexpr bool PP = false;     // main loop: false = one barrier per k-tile; true = ping-pong (H2BigPP / H2MidPP / H2LowPP)
    static constexpr int PHASES = 4;      // ping-pong loop: phases per k-tile (2: the (m0, m1) x n half loop of the smaller tiles)
    static constexpr bool LUT = false;    // A rows gathered from the conv1 pattern table (H2BigPPLut), see k_lut_build
};
typedef H2Cfg<2, 4, 4, 2> H2Big;      // conv2, conv4: 256 x 256
// the same tile with the ping-pong main loop: the two wave rows (= the two waves of every SIMD) run half a phase
// apart, so one of them is in its MFMA cluster while the other issues LDS reads and LDS-DMA (see k_gemm_h2)
struct H2BigPP : H2Cfg<2, 4, 4, 2> { static constexpr bool PP = true; };
struct H2MidPP : H2Cfg<2, 4, 3, 2> { static constexpr bool PP = true; static constexpr int PHASES = 2; };
// 128 x 256: conv3 of a call whose rows fit one grid round on this tile but leave a third of the CUs idle on the 192-row one
// (the arena's <= 512-leaf batches); same k order per output element as the other two: bit-identical
struct H2LowPP : H2Cfg<2, 4, 2, 2> { static constexpr bool PP = true; static constexpr int PHASES = 2; };
// the same tile with ONE phase per k-tile and three LDS stages (round 6): a 24-MFMA M section cannot cover an L section and its barrier (the 2-phase
// loop on this tile runs at 0.63 of its MFMA time); here the whole k-tile is one M section of 48 MFMAs and one L section of 16 fragment reads + 6 DMA
// instructions, i.e. half the barriers and half the fixed L-section costs per k-tile.  A tile is staged TWO k-tiles ahead (3 x 48 KB = 144 KB), so a
// piece has a whole phase pair to land.  Same products in the same order per output element: bit-identical to every other tile.
struct H2LowPP1 : H2Cfg<2, 4, 2, 2, 3> { static constexpr bool PP = true; static constexpr int PHASES = 1; };
// conv2 with conv1 folded into a lookup: the conv1 + BN + ReLU output of a pixel depends only on the 3 x 3 neighbourhood
// of the position (9 cells, each empty / own / opponent: 3^9 = 19683 patterns), so conv2's A rows are LDS-DMA'd straight
// from a table of the 19683 possible rows (+ one zero row for taps outside the board) instead of from a conv1 output
// buffer: no conv1 kernel, no act1 round trip, and the 40 MB table is re-read from L2 / MALL instead of streamed.
struct H2BigPPLut : H2BigPP { static constexpr bool LUT = true; };
#define OZ_LUT_PATTERNS 19683
#define OZ_LUT_ROWS (OZ_LUT_PATTERNS + 1)  // row OZ_LUT_PATTERNS = zeros
typedef H2Cfg<2, 4, 3, 2> H2Mid;      // conv3 (M = B*36): 192 x 256 -> 1536 blocks = 6.0 rounds of 256 CUs (256 x 256: 4.5)
typedef H2Cfg<2, 2, 2, 2> H2Small2;   // 128 x 128, two LDS stages (the 16-way split-K launches of one-position networks)
typedef H2Cfg<1, 2, 2, 2> H2Thin2;    // 64 x 128, 2 waves, two stages
// The small tiles serve launches with few k-tiles of MFMA work per wave: with two stages every k-tile exposes an LDS-DMA
// round trip; three stages keep two k-tiles in flight (counted vmcnt, one barrier per k-tile).  Same accumulation order,
// bit-identical results (tools/pp_race_check.py).  Measured: -4 .. -6 % per forward at 128 .. 512 positions, +-0 at 4096,
// but SLOWER on the 16-way split-K launches of one-position networks (9 k-tiles per block) -- oz_net.hip picks per network.
typedef H2Cfg<2, 2, 2, 2, 3> H2Small; // 128 x 128, 3 x 32 KB of LDS
// (round 5, measured and removed: four stages -- three k-tiles in flight -- for the trainer's launches of 18 .. 36 k-tiles per block: 1.04-1.07 against
//  0.98-1.00 ms per step at the reference's batch)
typedef H2Cfg<1, 2, 2, 2, 3> H2Thin;  // dense layers (M = batch): 64 x 128, 2 waves -> 512 / 256 blocks at B = 4096; 3 x 24 KB
// fc2 at large batches (one k-slice, 228 blocks of 32 k-tiles: at most one block per CU): the same 64 x 128 tile on FOUR waves of 32 x 64 --
// every SIMD has a wave and the MFMA section of a k-tile halves.  Same accumulation order per output element as H2Thin: bit-identical.
// Measured at 3640 positions (tools/tail_layers_probe.py, round 3): H2Thin 40.2 us, six stages instead of three 41.4 (not DMA-latency
// bound), a 32 x 128 tile on two waves (twice the blocks) 33.5, this 27.8.
typedef H2Cfg<2, 2, 1, 2, 3> H2Thin4w;

// Low-side guard of an h2 tensor (see the header of this file): cnt[row] = (forward number << 6 | low slices seen) for the rows of
// the tensor a kernel writes; cnt == nullptr switches the guard off (fp32 outputs, the trainer's calls, calibration passes).
#define H2_FLAG_OVER 1        // an activation above the fp16 range
#define H2_FLAG_LOW 4         // a row whose largest scaled activation is non-zero and below the threshold (bit 2 = 2 is the trainer's)
struct H2Low {
    unsigned* cnt = nullptr;
    unsigned seq = 0;                     // 26-bit number of the forward (never 0)
    float thr = 0.f;                      // 2^H2_LOW_GUARD by default (OZ_NET_OPT_LOW_GUARD_LOG2)
    int* flag = nullptr;
};
// one 64-channel slice of row `row` came out low (0 < max < thr): count it; the row is low when all `nslices` of its slices are.
// Rare by construction (a slice of 64 channels whose values all sit 2^15 below their calibration maxima), so a CAS loop is fine;
// the forward number makes stale counts of earlier forwards harmless -- the array is never cleared.
__device__ __forceinline__ void h2_low_report(const H2Low& lo, long long row, int nslices) {
    unsigned* p = lo.cnt + row;
    unsigned seen = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), next;
    for (;;) {
        next = (seen >> 6) == lo.seq ? seen + 1 : ((lo.seq << 6) | 1u);
        const unsigned was = atomicCAS(p, seen, next);
        if (was == seen) break;
        seen = was;
    }
    if ((int)(next & 63u) >= nslices) atomicOr(lo.flag, H2_FLAG_LOW);
}
// max of |x| over the 8 lanes of an aligned lane octet (every lane gets it)
__device__ __forceinline__ float h2_octet_max(float m) {
    m = fmaxf(m, __shfl_xor(m, 1, 64));
    m = fmaxf(m, __shfl_xor(m, 2, 64));
    return fmaxf(m, __shfl_xor(m, 4, 64));
}

struct H2Geom {
    int Hin, Hout, pad, Cin, taps;        // taps 9 (3x3 conv) or 1 (dense, Hin = Hout = 1)
    int N, K;                             // output channels, taps * Cin
    int out_h2;                           // 1: write the h2 layout, 0: write fp32 rows
    int relu;
    int ksplit;                           // > 1: split-K, raw partial sums to slab[ks] (no scale/shift), see k_splitk_reduce_h2
    long long slab;                       // elements between two partial slabs
    H2Low low;                            // low-side guard of an h2 output (off by default)
};

// finishes a split-K layer: out[m][n] = act((sum_s slab[s][m][n]) * scale[n] + shift[n]) with the slices added in a
// FIXED order (bit-reproducible), written in the h2 layout.  One thread per (row, 8 channels).
__global__ __launch_bounds__(256) void k_splitk_reduce_h2(const float* __restrict__ part, long long slab, int ksplit, int N, int P,
                                                          const int* __restrict__ d_count, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int relu, uint4* __restrict__ out,
                                                          int* __restrict__ flag, H2Low low) {
    const int ng = N >> 3;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long m = idx / ng;
    if (m >= (long long)(*d_count) * P) return;          // rows = positions x output pixels
    const int c8 = (int)(idx % ng) * 8;
    float acc[8];
    const float* p = part + (size_t)m * N + c8;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = p[j];
    // added in order s = 1, 2, ... (bit-reproducible), fetched four slices (8 x 16 B per thread) at a time: see k_splitk_reduce_f32
    for (int s0 = 1; s0 < ksplit; s0 += 4) {
        f32x4v b[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (s0 + q < ksplit) {
                const f32x4v* src = reinterpret_cast<const f32x4v*>(p + (size_t)(s0 + q) * slab);
                b[q][0] = src[0]; b[q][1] = src[1];
            }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (s0 + q < ksplit) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[j] += b[q][0][j]; acc[j + 4] += b[q][1][j]; }
            }
    }
    f16x8 h1, h2;
    bool over = false;
    float vmax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = fmaf(acc[j], scale[c8 + j], shift[c8 + j]);
        if (relu) v = fmaxf(v, 0.f);
        over |= fabsf(v) > H2_F16_MAX;
        vmax = fmaxf(vmax, fabsf(v));
        const _Float16 a = (_Float16)v;
        h1[j] = a;
        h2[j] = (_Float16)(v - (float)a);
    }
    if (over) atomicOr(flag, H2_FLAG_OVER);
    if (low.cnt) {                                       // 8 adjacent threads = one 64-channel slice of row m (ng % 8 == 0: whole octets pass the row test together)
        vmax = h2_octet_max(vmax);
        if ((threadIdx.x & 7) == 0 && vmax > 0.f && vmax < low.thr) h2_low_report(low, m, N >> 6);
    }
    uint4* dst = out + ((size_t)m * ng + (c8 >> 3)) * 2;
    dst[0] = *reinterpret_cast<uint4*>(&h1);
    dst[1] = *reinterpret_cast<uint4*>(&h2);
}

__device__ __forceinline__ void h2_split(float x, _Float16& h1, _Float16& h2) {
    h1 = (_Float16)x;
    h2 = (_Float16)(x - (float)h1);
}

// XOR key of LDS row `row` (applied to the 16-byte chunk index, on the DMA source side and on the read side).
// A ds_read_b128 is served in 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS);
// with the 16x16x32 operand map (lane l: row l&15, k-group l>>4) such a group holds all 16 rows, rows 0-3 and 12-15
// with one k-group and rows 4-11 with the next one, so the key must make  {key(r)} for r in {0-3,12-15}  and
// {2 ^ key(r)} for r in {4-11}  cover all eight chunk slots of each row parity: key = bit1(r) | 6*bit3(r).
// (the natural (row>>1)&7 key is conflict-free only for the 32x32x16 map: measured 50 % conflict cycles here)
__device__ __forceinline__ int h2_swz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) * 6); }

typedef const __attribute__((address_space(1))) void* h2_gptr;
typedef __attribute__((address_space(3))) void* h2_lptr;

// conv1 + plane unpack, output in the h2 layout.  One thread = one board row (n pixels) x 8 channels: the 18
// weight vectors are loaded once per thread and applied to the row's pixels (input bits tested in registers).
__global__ __launch_bounds__(256) void k_conv1_h2(const uint64_t* __restrict__ own, const uint64_t* __restrict__ opp,
                                                  const int* __restrict__ d_count, int n, int C,
                                                  const float* __restrict__ W /*[9][2][C]*/, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, uint4* __restrict__ out, int* __restrict__ flag, H2Low low) {
    const int cg = C >> 3;
    const long long R = (long long)(*d_count) * n;                  // board rows
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long br = idx / cg;
    if (br >= R) return;
    const int c8 = (int)(idx % cg) * 8;
    const int b = (int)(br / n), y = (int)(br % n);
    const uint64_t o = own[b], p = opp[b];
    float acc[8][8];
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[x][j] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = y + ky - 1;
        if (iy < 0 || iy >= n) continue;
        const unsigned ro = (unsigned)((o >> (iy * 8)) & 0xFF), rp = (unsigned)((p >> (iy * 8)) & 0xFF);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const f32x4 w0a = *reinterpret_cast<const f32x4*>(W + (size_t)((ky * 3 + kx) * 2 + 0) * C + c8);
            const f32x4 w0b = *reinterpret_cast<const f32x4*>(W + (size_t)((ky * 3 + kx) * 2 + 0) * C + c8 + 4);
            const f32x4 w1a = *reinterpret_cast<const f32x4*>(W + (size_t)((ky * 3 + kx) * 2 + 1) * C + c8);
            const f32x4 w1b = *reinterpret_cast<const f32x4*>(W + (size_t)((ky * 3 + kx) * 2 + 1) * C + c8 + 4);
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int ix = x + kx - 1;
                if (ix < 0 || ix >= 8) continue;                    // columns >= n are never occupied
                const float a0 = (float)((ro >> ix) & 1), a1 = (float)((rp >> ix) & 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[x][j] = fmaf(a0, w0a[j], acc[x][j]);
                    acc[x][j + 4] = fmaf(a0, w0b[j], acc[x][j + 4]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[x][j] = fmaf(a1, w1a[j], acc[x][j]);
                    acc[x][j + 4] = fmaf(a1, w1b[j], acc[x][j + 4]);
                }
            }
        }
    }
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = scale[c8 + j]; sh[j] = shift[c8 + j]; }
    bool over = false;
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        if (x >= n) continue;
        f16x8 h1, h2;
        float vmax = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = fmaxf(fmaf(acc[x][j], sc[j], sh[j]), 0.f);
            over |= v > H2_F16_MAX;
            vmax = fmaxf(vmax, v);
            _Float16 a, bb;
            h2_split(v, a, bb);
            h1[j] = a; h2[j] = bb;
        }
        if (low.cnt) {                                   // cg % 8 == 0: the 8 threads of a 64-channel slice sit in one lane octet, same board row
            vmax = h2_octet_max(vmax);
            if ((threadIdx.x & 7) == 0 && vmax > 0.f && vmax < low.thr) h2_low_report(low, br * n + x, C >> 6);
        }
        uint4* dst = out + ((size_t)(br * n + x) * cg + (c8 >> 3)) * 2;
        dst[0] = *reinterpret_cast<uint4*>(&h1);
        dst[1] = *reinterpret_cast<uint4*>(&h2);
    }
    if (over) atomicOr(flag, H2_FLAG_OVER);
}

// ---- weight preparation on the device (oz_net_commit): the Keras kernels are uploaded as stored, [K][N] with k = tap * Cin + ci
// [N][K] fp32 for the precision-f32 GEMM
__global__ __launch_bounds__(256) void k_w_transpose(const float* __restrict__ src, int K, int N, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int k0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8 threads
    for (int r = ty; r < 32; r += 8)
        if (k0 + r < K && c0 + tx < N) tile[r][tx] = src[(size_t)(k0 + r) * N + c0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (c0 + r < N && k0 + tx < K) out[(size_t)(c0 + r) * K + k0 + tx] = tile[tx][r];
}
// h2 layout [N][K/8][h1 x 8 | h2 x 8] of  w[k][c] * 2^(colexp[c] - inexp[k % Cmod])  in the GEMM's k order k' = (slice * taps + tap) * 32 + c32:
// colexp[c] moves column c's largest |w| into [2^(H2_W_TOP - 1), 2^H2_W_TOP), inexp[] divides the consumed tensor's per-channel activation scale out
// (k % Cmod = the input channel of reduction index k: tap * Cin + ci for the 3x3 layers, pixel * C + c for fc1's flattened input).
// One thread per (output channel c, group of 8 k'); adjacent threads = adjacent c (coalesced reads of the [K][N] source).
__global__ __launch_bounds__(256) void k_w_to_h2(const float* __restrict__ src, int K, int N, int taps, const int* __restrict__ colexp,
                                                 const int* __restrict__ inexp, int Cmod, uint4* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(idx % N);
    const int grp = (int)(idx / N);
    if (grp >= (K >> 3)) return;
    const int Cin = K / taps;
    const int ce = colexp[c];
    f16x8 h1, h2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int kp = grp * 8 + j, tile = kp >> 5, c32 = kp & 31, slice = tile / taps, tap = tile - slice * taps;
        const int k = tap * Cin + slice * 32 + c32;
        const float x = ldexpf(src[(size_t)k * N + c], ce - inexp[k % Cmod]);
        const _Float16 a = (_Float16)x;
        h1[j] = a; h2[j] = (_Float16)(x - (float)a);
    }
    uint4* dst = out + ((size_t)c * (K >> 3) + grp) * 2;
    dst[0] = *reinterpret_cast<uint4*>(&h1);
    dst[1] = *reinterpret_cast<uint4*>(&h2);
}
// colmax[c] = max over k of |w[k][c]| * 2^-inexp[k % Cmod]  (bit pattern of a non-negative float: atomicMax on unsigned orders them)
__global__ __launch_bounds__(256) void k_w_colmax(const float* __restrict__ src, int K, int N, const int* __restrict__ inexp, int Cmod,
                                                  unsigned* __restrict__ colmax) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const int k0 = blockIdx.y * 64, k1 = k0 + 64 < K ? k0 + 64 : K;
    float m = 0.f;
    for (int k = k0; k < k1; ++k) m = fmaxf(m, fabsf(ldexpf(src[(size_t)k * N + c], -inexp[k % Cmod])));
    if (m > 0.f) atomicMax(colmax + c, __float_as_uint(m));
}
// colmax[c] = max over rows of |x[row][c]| of fp32 rows [*d_count * P][N] (calibration passes: a layer's BN output, BEFORE the ReLU, before
// its scale is chosen.  Before the ReLU because the post-ReLU maximum of a channel that hovers around zero is not a scale: it was 3e-5 on 512
// calibration positions and 300 times that on a self-play position -- measured, round 4 -- while max |z| is a few sigma either way.)
__global__ __launch_bounds__(256) void k_rows_colmax(const float* __restrict__ x, const int* __restrict__ d_count, int P, int N,
                                                     unsigned* __restrict__ colmax) {
    const long long rows = (long long)(*d_count) * P;
    const long long r0 = (long long)blockIdx.y * 64;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N || r0 >= rows) return;
    const long long r1 = r0 + 64 < rows ? r0 + 64 : rows;
    float m = 0.f;
    for (long long r = r0; r < r1; ++r) {                 // a non-finite value turns the maximum into +Inf, which the host refuses
        const float a = fabsf(x[(size_t)r * N + c]);
        m = a <= 3.0e38f ? fmaxf(m, a) : __builtin_inff();
    }
    if (m > 0.f) atomicMax(colmax + c, __float_as_uint(m));
}
// the same over the 3^9 conv1 rows (k_lut_build's fmaf sequence): the exact supremum of |conv1's BN output| per channel
__global__ __launch_bounds__(256) void k_lut_colmax(int C, const float* __restrict__ W /*[9][2][C]*/, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, unsigned* __restrict__ colmax) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int id0 = blockIdx.y * 64, id1 = id0 + 64 < OZ_LUT_PATTERNS ? id0 + 64 : OZ_LUT_PATTERNS;
    const float sc = scale[c], sh = shift[c];
    float m = 0.f;
    for (int id = id0; id < id1; ++id) {
        float acc = 0.f;
        unsigned rest = (unsigned)id;
        for (int t = 0; t < 9; ++t) {
            const unsigned cell = rest % 3; rest /= 3;
            acc = fmaf(cell == 1 ? 1.f : 0.f, W[(size_t)(t * 2 + 0) * C + c], acc);
            acc = fmaf(cell == 2 ? 1.f : 0.f, W[(size_t)(t * 2 + 1) * C + c], acc);
        }
        m = fmaxf(m, fabsf(fmaf(acc, sc, sh)));             // |BN output| before the ReLU: a stable scale also for a channel that hovers around zero
    }
    if (m > 0.f) atomicMax(colmax + c, __float_as_uint(m));
}

// pattern id of every pixel: sum over the 3 x 3 neighbourhood (ky, kx) of 3^(ky*3+kx) * {0 empty or off the board, 1 own, 2 opponent}.
// Layout: PADDED boards [batch][(n+2)][(n+2)] of 32-bit ids, the one-cell border holding OZ_LUT_PATTERNS -- the index of the all-zero
// row every table ends with -- so that the consumers (k_conv2_lut*, the H2BigPPLut gather) read the 3 x 3 window of a pixel with no
// bounds test and an off-board tap adds an exact +0.0 (wave-uniform 32-bit ids are also what the scalar unit can load).
// low_rows (optional): low_rows[id] != 0 marks a pattern whose conv1 row is LOW in the h2 layout (k_lut_rowlow, at commit); a batch that
// holds such a pixel raises the low-side flag -- the act1 leg of the guard costs one byte read per pixel.
__global__ __launch_bounds__(256) void k_lut_ids(const uint64_t* __restrict__ own, const uint64_t* __restrict__ opp,
                                                 const int* __restrict__ d_count, int n, unsigned* __restrict__ ids,
                                                 const unsigned char* __restrict__ low_rows, int* __restrict__ flag) {
    const int W = n + 2, PW = W * W;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)(*d_count) * PW) return;
    const int b = (int)(idx / PW), cell = (int)(idx % PW), y = cell / W - 1, x = cell % W - 1;
    if (y < 0 || y >= n || x < 0 || x >= n) { ids[idx] = OZ_LUT_PATTERNS; return; }
    const uint64_t o = own[b], p = opp[b];
    unsigned id = 0, pw = 1;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = y + ky - 1, ix = x + kx - 1;
            if (iy >= 0 && iy < n && ix >= 0 && ix < n) {
                const int bit = iy * 8 + ix;
                id += pw * ((unsigned)((o >> bit) & 1) + 2u * (unsigned)((p >> bit) & 1));
            }
            pw *= 3;
        }
    ids[idx] = id;
    if (low_rows && low_rows[id]) atomicOr(flag, H2_FLAG_LOW);
}
// low[id] = 1 when the conv1 table row of pattern id is non-zero and its largest h1 value is below thr (one wave per row)
__global__ __launch_bounds__(64) void k_lut_rowlow(int C, const uint4* __restrict__ table, float thr, unsigned char* __restrict__ low) {
    const int id = blockIdx.x, lane = threadIdx.x, cg = C >> 3;
    float m = 0.f;
    for (int g = lane; g < cg; g += 64) {
        const uint4 q = table[((size_t)id * cg + g) * 2];          // the h1 chunk of channel group g
        const f16x8 h = *reinterpret_cast<const f16x8*>(&q);
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf((float)h[j]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) low[id] = (m > 0.f && m < thr) ? 1 : 0;
}

// table[id] = the h2 row k_conv1_h2 writes for a pixel whose neighbourhood is pattern id: the same fmaf sequence (taps in
// ky, kx order, own plane then opponent plane; a tap k_conv1_h2 skips as off-board is an exact no-op fmaf(0, w, acc) here),
// the same scale / shift / ReLU / split -- bit-identical rows.  One thread per (pattern, 8 channels).
__global__ __launch_bounds__(256) void k_lut_build(int C, const float* __restrict__ W /*[9][2][C]*/, const float* __restrict__ scale,
                                                   const float* __restrict__ shift, uint4* __restrict__ table, int* __restrict__ flag) {
    const int cg = C >> 3;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long id = idx / cg;
    if (id >= OZ_LUT_PATTERNS) return;
    const int c8 = (int)(idx % cg) * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    unsigned rest = (unsigned)id;
    for (int t = 0; t < 9; ++t) {
        const unsigned cell = rest % 3; rest /= 3;
        const float a0 = cell == 1 ? 1.f : 0.f, a1 = cell == 2 ? 1.f : 0.f;
        const float* w0 = W + (size_t)(t * 2 + 0) * C + c8;
        const float* w1 = W + (size_t)(t * 2 + 1) * C + c8;
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
    uint4* dst = table + ((size_t)id * cg + (c8 >> 3)) * 2;
    dst[0] = *reinterpret_cast<uint4*>(&h1);
    dst[1] = *reinterpret_cast<uint4*>(&h2);
    if (over) atomicOr(flag, H2_FLAG_OVER);
}

// conv2 as a gather-sum (conv1 AND conv2 folded into tables).  With act1[q] = table[id(q)], conv2's pre-activation at
// pixel p is sum over taps t of W_t . table[id(p + t)] = sum_t T2[t][id(p + t)], where T2[t][id] = W_t . table[id] is a
// C-vector that depends on the weights only: 9 x OZ_LUT_ROWS rows of C floats (363 MB at C = 512; row OZ_LUT_PATTERNS of every
// tap = zeros), built at commit by nine [19683 x C] x [C x C] GEMMs on k_gemm_h2 (same arithmetic as the convolution: fp32 as
// 2 x fp16, fp32 accumulate).  The layer is then 9 row reads + 9 fp32 adds per pixel -- 18 KB of (mostly L2 / MALL resident)
// table per pixel instead of 4.7 MFLOP: byte work.  Taps are added in order t = 0..8 from +0.0: a fixed order, independent of batch
// size and position; an off-board tap reads the zero row (the accumulator is never -0.0, so adding +0.0 changes no bit).
// Table layout (round 3): SLICE-major, T2[slice = c / 64][tap][id][64 channels] -- a (tap, id) row of C channels is C / 64 records of
// 256 B, and all records of one 64-channel slice are contiguous (45 MB at 19684 rows).  At C = 512 there are 8 slices = the 8 XCDs:
// k_conv2_lut_xcd gives every XCD ONE slice of ALL pixels (workgroup ids go round-robin over the XCDs: slice = blockIdx.x & 7), so
// the eight private 4 MiB L2s hold eight DIFFERENT eighths of the table instead of eight copies of the same hot rows -- the L2
// capacity the gather sees is 32 MiB, not 4.  Measured on one device, 3640 mid-game leaves per launch (rocprofv3 --pmc, round 3,
// profiles/r3_conv2_gather_pmc.csv): whole rows per XCD (the round-2 mapping) 11.7 M L2->fabric read requests (1.5 GB) + 0.50 GB
// written in 0.39 ms; one slice per XCD 7.1 M requests (0.91 GB) + 0.48 GB written in 0.30 ms -- both ~4.5-5 TB/s of HBM-side
// traffic, i.e. the launch is bound by the bytes that miss L2 (the 363 MB table cannot live in the 256 MB Infinity Cache), not by
// instruction issue: a wave-per-pixel variant with 4x fewer VALU instructions (130 against the round-2 kernel's 462) measured the
// same 0.39 ms as the round-2 kernel, and half-line stores (16 B per lane at a 32-byte stride) inflated the written bytes to
// 0.83 GB and cancelled the gain until the stores became whole lines.
// Ablation of THIS kernel (same probe, 0.285 ms whole on the box of the second series): the stores compiled in but never taken 0.208 ms;
// the same with the pattern ids masked to 1024 (every record an L2 hit) and the real stores 0.200 ms; real reads with the stores folded onto
// 4096 pixels (L2-resident lines) 0.27 ms; every pixel reading the same nine records (all L1 hits) and the real stores 0.145 ms; the stores
// alone (tools/ubench/store_patterns.hip) 0.075 ms = 6.2 TB/s; random 256-byte records alone (tools/ubench/random_records.hip) 6.4 TB/s.
// So the wall is neither HBM nor the store pattern.  Counters (profiles/r3_conv2_gather_tcp_counters.txt): 2 183 TLB misses in 93 M translations;
// 93.9 M 64-byte L1 accesses = 57 % of the launch's cycles per CU, another 39 % the L1 waits on L2 data; 22.6 M L1 -> L2 read requests (2.9 GB:
// L1 hit rate 33 % -- the L1 holds 256 lines and 16 waves x 144 lines are in flight, so only same-instruction duplicates and the zero record
// hit; on positions of all plies the 128 most frequent (tap, pattern) records cover 22 % of the reads, the 8192 most frequent 64 %) at 372
// cycles average latency, 7.5 M 64-byte write requests.  What the launch time follows is the traffic through the eight L2s -- 2.9 GB of record
// reads + 0.56 GB of miss fills + 0.47 GB of writes at half the read rate, ~5 GB-equivalent at the ~17 TB/s the L2s sustain on random lines
// (half their 34 TB/s peak) = 0.28 ms; every variant that leaves that traffic alone measured +-0 (bit-identical outputs each time):
//   the off-board taps (16 % of the record loads) dropped by the range check of per-tap buffer descriptors: 0.277 -> 0.285 / 0.280 ms;
//   the 30 neighbourhood ids of a wave's board row loaded once per wave and handed out by ds_bpermute (8 lanes x 8 pixels x 9 ids before): 0.273;
//   both (25 % fewer L1 accesses): 0.268 .. 0.280;  5 instead of 4 waves per SIMD (scale / shift loaded late, 88 VGPRs): 0.291 .. 0.302;
//   HALF the waves per CU: 0.345 (so not latency-bound either);  pixels ordered by their centre pattern (host-sorted probe): 0.283 against 0.292.
// Plain instead of streaming stores: 0.297; sc0 sc1 nt / sc1 stores: +-0 / 0.307; 1, 4 or 8 pixel groups per thread: +-0 or slower.
// Hiding it instead (round 3, measured and removed): the batch as 2 / 4 position ranges, the gather of range i + 1 on a second stream under
// conv3 of range i (the gather leaves the matrix cores idle, conv3 the L2s at ~40 %) -- bit-identical, the kernels do overlap, and conv3
// slows by what the gather gains (conv3 per batch 1.094 -> 1.113 / 1.239 ms, whole step 213.4 -> 218.3 / 219.9 ms): the chip is power
// limited, the overlapped work is paid in clock.
#define OZ_C2L_SLICE 64       // channels per slice record (256 B)
__device__ __forceinline__ size_t t2_record(int slice, int t, unsigned id) {          // float index of record (slice, tap, id)
    return (((size_t)slice * 9 + t) * OZ_LUT_ROWS + id) * OZ_C2L_SLICE;
}
// one output row piece: BN + ReLU of 8 channel sums, then the h2 split (two 16-byte streaming stores) or fp32 (two 16-byte stores)
// OUT: 0 = fp32 rows, 1 = the h2 layout, 2 = the b3 layout (three bf16 planes, oz_net_b3.h)
__device__ __forceinline__ void b3_split(float x, __bf16& b1, __bf16& b2, __bf16& b3);
template <int OUT>
__device__ __forceinline__ bool c2l_finish(const f32x4& lo, const f32x4& hi, const float* __restrict__ scale, const float* __restrict__ shift, int c8,
                                           void* __restrict__ out, size_t pixel, int C, float& vmax, float relu_floor) {
    const f32x4 sc0 = *reinterpret_cast<const f32x4*>(scale + c8), sc1 = *reinterpret_cast<const f32x4*>(scale + c8 + 4);
    const f32x4 sh0 = *reinterpret_cast<const f32x4*>(shift + c8), sh1 = *reinterpret_cast<const f32x4*>(shift + c8 + 4);
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = fmaxf(fmaf(lo[j], sc0[j], sh0[j]), relu_floor);
        v[j + 4] = fmaxf(fmaf(hi[j], sc1[j], sh1[j]), relu_floor);
    }
    vmax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) vmax = fmaxf(vmax, v[j]);
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    bool over = false;
    constexpr bool OUT_H2 = OUT == 1;
    if constexpr (OUT == 2) {
        typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
        bf16x8_t p0, p1, p2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __bf16 a, b, c;
            b3_split(v[j], a, b, c);
            p0[j] = a; p1[j] = b; p2[j] = c;
        }
        v4u* dst = reinterpret_cast<v4u*>(reinterpret_cast<uint4*>(out) + pixel * (size_t)(C / 32 * 12) + (c8 >> 5) * 12 + ((c8 >> 3) & 3));
        __builtin_nontemporal_store(*reinterpret_cast<v4u*>(&p0), dst);
        __builtin_nontemporal_store(*reinterpret_cast<v4u*>(&p1), dst + 4);
        __builtin_nontemporal_store(*reinterpret_cast<v4u*>(&p2), dst + 8);
    } else if constexpr (OUT_H2) {
        f16x8 h1, h2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            over |= v[j] > H2_F16_MAX;
            _Float16 a, bb;
            h2_split(v[j], a, bb);
            h1[j] = a; h2[j] = bb;
        }
        // streaming stores: the 0.5 GB of output would otherwise evict table records from L2 / Infinity Cache
        v4u* dst = reinterpret_cast<v4u*>(reinterpret_cast<uint4*>(out) + (pixel * (size_t)(C >> 3) + (c8 >> 3)) * 2);
        __builtin_nontemporal_store(*reinterpret_cast<v4u*>(&h1), dst);
        __builtin_nontemporal_store(*reinterpret_cast<v4u*>(&h2), dst + 1);
    } else {
        float* dst = reinterpret_cast<float*>(out) + pixel * (size_t)C + c8;
        const f32x4 r0 = {v[0], v[1], v[2], v[3]}, r1 = {v[4], v[5], v[6], v[7]};
        __builtin_nontemporal_store(r0, reinterpret_cast<f32x4*>(dst));
        __builtin_nontemporal_store(r1, reinterpret_cast<f32x4*>(dst + 4));
    }
    return over;
}

// C = 512: block b works on slice b & 7 (its XCD's eighth of the table) for pixels [(b >> 3) * 32 * PPT, ...): a wave = 8 pixels x 8 lanes
// (lane j of a pixel: channels 64 * slice + 8 j .. + 7, two 16-byte loads per tap), OZ_C2L_PPT pixel groups per thread.
#define OZ_C2L_PPT 2
#define C2L_B3_PIX 448          // bytes per pixel of the b3 staging image: 384 + 64 (pixels 0..3 of a half wave start in four different bank quarters)
// pattern id of cell (cy, cx) of an N x N board straight from the bitboards (k_lut_ids' arithmetic; off the board = the all-zero row)
template <int N>
__device__ __forceinline__ unsigned lut_id_of(uint64_t o, uint64_t p, int cy, int cx) {
    if (cy < 0 || cy >= N || cx < 0 || cx >= N) return OZ_LUT_PATTERNS;
    unsigned id = 0, pw = 1;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = cy + ky - 1, ix = cx + kx - 1;
            if (iy >= 0 && iy < N && ix >= 0 && ix < N) {
                const int bit = iy * 8 + ix;
                id += pw * ((unsigned)((o >> bit) & 1) + 2u * (unsigned)((p >> bit) & 1));
            }
            pw *= 3;
        }
    return id;
}
// INLINE_IDS (few positions: the latency path of precision f32): the nine pattern ids of a pixel are computed from the bitboards here instead of
// read from the padded id boards k_lut_ids writes -- one launch less in front of a forward that is all launch latency (same ids, same sums)
template <int N, int OUT, bool INLINE_IDS = false>
__global__ __launch_bounds__(256) void k_conv2_lut_xcd(const unsigned* __restrict__ ids, const int* __restrict__ d_count,
                                                       const float* __restrict__ T2, const float* __restrict__ scale, const float* __restrict__ shift,
                                                       void* __restrict__ out, int* __restrict__ flag, H2Low low, float floor,
                                                       const uint64_t* __restrict__ own = nullptr, const uint64_t* __restrict__ opp = nullptr) {
    constexpr int P = N * N, W = N + 2, PW = W * W, C = 512;
    constexpr bool OUT_H2 = OUT == 1, OUT_B3 = OUT == 2;
    __shared__ __attribute__((aligned(16))) unsigned char c2l_b3_stage[OUT_B3 ? 4 * 8 * C2L_B3_PIX : 16];      // b3 output: a wave's 8 pixels x 384 B (+ pad)
    const float relu_floor = OUT_H2 ? 0.f : floor;            // fp32 rows: 0 = ReLU, -inf = the BN output itself (calibration passes)
    const int slice = blockIdx.x & 7, j = threadIdx.x & 7;
    const long long total = (long long)(*d_count) * P;
    // whole 128-byte lines per wave instruction: the 8 lanes of a pixel read 16 B each of line 0 of the record (channels 4 j .. 4 j + 3 of the
    // slice) and of line 1 (channels 32 + 4 j ..); the stores are whole lines too (h2 output: lane pairs swap halves, below)
    const int ca = slice * OZ_C2L_SLICE + j * 4, cb = ca + 32;
    const f32x4 sca = *reinterpret_cast<const f32x4*>(scale + ca), scb = *reinterpret_cast<const f32x4*>(scale + cb);
    const f32x4 sha = *reinterpret_cast<const f32x4*>(shift + ca), shb = *reinterpret_cast<const f32x4*>(shift + cb);
    bool over = false;
#pragma unroll 1
    for (int k = 0; k < OZ_C2L_PPT; ++k) {
        const long long pixel = ((long long)(blockIdx.x >> 3) * OZ_C2L_PPT + k) * 32 + (threadIdx.x >> 3);
        if (pixel >= total) break;
        const int b = (int)(pixel / P), pix = (int)(pixel - (long long)b * P), y = pix / N, x = pix - y * N;
        unsigned id[9];
        if constexpr (INLINE_IDS) {
            const uint64_t o = own[b], p = opp[b];
#pragma unroll
            for (int t = 0; t < 9; ++t) id[t] = lut_id_of<N>(o, p, y + t / 3 - 1, x + t % 3 - 1);
        } else {
            const unsigned* idp = ids + (size_t)b * PW + y * W + x;
#pragma unroll
            for (int t = 0; t < 9; ++t) id[t] = idp[(t / 3) * W + t % 3];
        }
        f32x4 ra[9], rb[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float* rec = T2 + t2_record(slice, t, id[t]) + j * 4;
            ra[t] = *reinterpret_cast<const f32x4*>(rec);
            rb[t] = *reinterpret_cast<const f32x4*>(rec + 32);
        }
        __builtin_amdgcn_sched_barrier(0);                    // all 18 record loads in flight before the first add (-6 %: 0.303 -> 0.285 ms; 1, 4 or 8 pixel groups per thread: +-0 or slower)
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) { lo += ra[t]; hi += rb[t]; }
        f32x4 va, vb;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            va[q] = fmaxf(fmaf(lo[q], sca[q], sha[q]), relu_floor);
            vb[q] = fmaxf(fmaf(hi[q], scb[q], shb[q]), relu_floor);
        }
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        if constexpr (OUT_H2) {
            // group g = j >> 1 of line 0 (and of line 1): lane 2 g holds channels 8 g .. 8 g + 3, lane 2 g + 1 channels 8 g + 4 .. + 7.  The h2
            // layout wants [h1 x 8][h2 x 8] per group: the even lane sends its h2 halves and receives the partner's h1 halves, so that the even lane
            // owns the 16-byte h1 chunk and the odd lane the h2 chunk -- 8 lanes x 16 B = one whole 128-byte line per store instruction
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            union Pk { f16x4 h; unsigned u[2]; };
            Pk h1a, h2a, h1b, h2b;
            if (low.cnt) {                                   // the 8 lanes of a pixel hold its whole 64-channel slice
                float vmax = fmaxf(fmaxf(fmaxf(va[0], va[1]), fmaxf(va[2], va[3])), fmaxf(fmaxf(vb[0], vb[1]), fmaxf(vb[2], vb[3])));
                vmax = h2_octet_max(vmax);
                if (j == 0 && vmax > 0.f && vmax < low.thr) h2_low_report(low, pixel, C >> 6);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                over |= va[q] > H2_F16_MAX || vb[q] > H2_F16_MAX;
                _Float16 x1, x2;
                h2_split(va[q], x1, x2); h1a.h[q] = x1; h2a.h[q] = x2;
                h2_split(vb[q], x1, x2); h1b.h[q] = x1; h2b.h[q] = x2;
            }
            const bool even = (j & 1) == 0;
            unsigned give[4] = {even ? h2a.u[0] : h1a.u[0], even ? h2a.u[1] : h1a.u[1], even ? h2b.u[0] : h1b.u[0], even ? h2b.u[1] : h1b.u[1]};
            unsigned got[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) got[q] = (unsigned)__shfl_xor((int)give[q], 1, 64);
            v4u c0, c1;       // even lane: [own h1 | partner's h1];  odd lane: [partner's h2 | own h2]
            c0[0] = even ? h1a.u[0] : got[0]; c0[1] = even ? h1a.u[1] : got[1]; c0[2] = even ? got[0] : h2a.u[0]; c0[3] = even ? got[1] : h2a.u[1];
            c1[0] = even ? h1b.u[0] : got[2]; c1[1] = even ? h1b.u[1] : got[3]; c1[2] = even ? got[2] : h2b.u[0]; c1[3] = even ? got[3] : h2b.u[1];
            // streaming stores: the 0.5 GB of output would otherwise evict table records from L2 / Infinity Cache
            v4u* dst = reinterpret_cast<v4u*>(reinterpret_cast<unsigned char*>(out) + (size_t)pixel * (C * 4) + slice * 256 + j * 16);
            __builtin_nontemporal_store(c0, dst);
            __builtin_nontemporal_store(c1, dst + 8);
        } else if constexpr (OUT_B3) {
            // b3 layout (oz_net_b3.h): the slice's 64 channels are k-tiles 2 slice (line 0: channels 4 j ..) and 2 slice + 1 (line 1) of the pixel's row, a k-tile
            // = [plane 0 | plane 1 | plane 2] x 64 B, so the slice is 384 contiguous bytes of the row.  A lane holds 4 channels (8 B) of each of the six
            // (k-tile, plane) runs; the wave's 8 pixels are transposed through a private LDS image (6 x ds_write_b64, 3 x ds_read_b128 per lane) so that every
            // store instruction writes whole 128-byte lines (8 lanes x 16 B per pixel).  Measured against the direct form (lane pairs exchange halves, two
            // 64-byte runs per pixel and instruction): see docs/HISTORY.md, round 6.
            typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
            union Pk { bf16x4_t h; uint2 u; };
            Pk a[3], b[3];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __bf16 x1, x2, x3;
                b3_split(va[q], x1, x2, x3); a[0].h[q] = x1; a[1].h[q] = x2; a[2].h[q] = x3;
                b3_split(vb[q], x1, x2, x3); b[0].h[q] = x1; b[1].h[q] = x2; b[2].h[q] = x3;
            }
            unsigned char* img = c2l_b3_stage + (threadIdx.x >> 6) * (8 * C2L_B3_PIX) + ((threadIdx.x >> 3) & 7) * C2L_B3_PIX;     // this pixel's 384 B (+ 64 B pad: bank spread)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                *reinterpret_cast<uint2*>(img + p * 64 + j * 8) = a[p].u;
                *reinterpret_cast<uint2*>(img + 192 + p * 64 + j * 8) = b[p].u;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): this wave's image is written (each wave owns its eight pixels: no block barrier)
            __builtin_amdgcn_wave_barrier();
            unsigned char* base = reinterpret_cast<unsigned char*>(out) + (size_t)pixel * (C * 6) + slice * 384 + j * 16;
#pragma unroll
            for (int k3 = 0; k3 < 3; ++k3) {
                const v4u c = *reinterpret_cast<const v4u*>(img + k3 * 128 + j * 16);
                __builtin_nontemporal_store(c, reinterpret_cast<v4u*>(base + k3 * 128));
            }
            __builtin_amdgcn_wave_barrier();
        } else {
            float* dst = reinterpret_cast<float*>(out) + (size_t)pixel * C + ca;
            __builtin_nontemporal_store(va, reinterpret_cast<f32x4*>(dst));
            __builtin_nontemporal_store(vb, reinterpret_cast<f32x4*>(dst + 32));
        }
    }
    if (OUT_H2 && over) atomicOr(flag, H2_FLAG_OVER);
}

// any channel count: one thread per (pixel, 8 channels); same sums in the same order
template <int OUT>
__global__ __launch_bounds__(256) void k_conv2_lut(const unsigned* __restrict__ ids, const int* __restrict__ d_count, int n, int C,
                                                   const float* __restrict__ T2, const float* __restrict__ scale, const float* __restrict__ shift,
                                                   void* __restrict__ out, int* __restrict__ flag, H2Low low, float floor) {
    const int cg = C >> 3, P = n * n, W = n + 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long pixel = idx / cg;
    if (pixel >= (long long)(*d_count) * P) return;
    const int c8 = (int)(idx % cg) * 8;
    const int b = (int)(pixel / P), pix = (int)(pixel % P), y = pix / n, x = pix % n;
    const unsigned* idp = ids + (size_t)b * W * W + y * W + x;
    // memory-level parallelism: all 9 pattern ids first, then all 18 record loads in flight together, then the sum in tap order
    unsigned id[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) id[t] = idp[(t / 3) * W + t % 3];
    f32x4 ra[9], rb[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float* rec = T2 + t2_record(c8 / OZ_C2L_SLICE, t, id[t]) + (c8 % OZ_C2L_SLICE);
        ra[t] = *reinterpret_cast<const f32x4*>(rec);
        rb[t] = *reinterpret_cast<const f32x4*>(rec + 4);
    }
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) { lo += ra[t]; hi += rb[t]; }
    float vmax;
    constexpr bool OUT_H2 = OUT == 1;
    if (c2l_finish<OUT>(lo, hi, scale, shift, c8, out, (size_t)pixel, C, vmax, OUT_H2 ? 0.f : floor)) atomicOr(flag, H2_FLAG_OVER);
    if (OUT_H2 && low.cnt) {                                 // cg % 8 == 0: a 64-channel slice = one lane octet of one pixel
        vmax = h2_octet_max(vmax);
        if ((threadIdx.x & 7) == 0 && vmax > 0.f && vmax < low.thr) h2_low_report(low, pixel, C >> 6);
    }
}

// commit-time re-layout: one tap's GEMM output rows [OZ_LUT_PATTERNS][C] -> the slice-major records of that tap (row OZ_LUT_PATTERNS = zeros)
__global__ __launch_bounds__(256) void k_t2_to_slices(const float* __restrict__ rows, int C, int t, float* __restrict__ T2) {
    const long long idx = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;          // 4 channels per thread
    if (idx >= (long long)OZ_LUT_ROWS * C) return;
    const unsigned id = (unsigned)(idx / C);
    const int c = (int)(idx % C);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (id < OZ_LUT_PATTERNS) v = *reinterpret_cast<const f32x4*>(rows + idx);
    *reinterpret_cast<f32x4*>(T2 + t2_record(c / OZ_C2L_SLICE, t, id) + (c % OZ_C2L_SLICE)) = v;
}

// ---- the same two tables for precision f32 (exact fp32 arithmetic: rows and sums in fp32, T2 built by the fp32 MFMA GEMM)
// table32[id][C] = the row k_conv1 writes for a pixel with neighbourhood pattern id (same fmaf sequence; off-board taps
// that k_conv1 skips are exact no-ops here)
__global__ __launch_bounds__(256) void k_lut_build_f32(int C, const float* __restrict__ W /*[9][2][C]*/, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, float* __restrict__ table) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long id = idx / C;
    if (id >= OZ_LUT_PATTERNS) return;
    const int c = (int)(idx % C);
    float acc = 0.f;
    unsigned rest = (unsigned)id;
    for (int t = 0; t < 9; ++t) {
        const unsigned cell = rest % 3; rest /= 3;
        acc = fmaf(cell == 1 ? 1.f : 0.f, W[(size_t)(t * 2 + 0) * C + c], acc);
        acc = fmaf(cell == 2 ? 1.f : 0.f, W[(size_t)(t * 2 + 1) * C + c], acc);
    }
    table[(size_t)id * C + c] = fmaxf(fmaf(acc, scale[c], shift[c]), 0.f);
}
// out[M][N] = act((A[M][K] . W[N][K]^T) * scale + shift); A and W in the h2 layout; M = *d_count * Hout^2.
// CF::LUT: `in` is the pattern table, lut_ids the per-pixel pattern ids [batch][Hin^2] (k_lut_ids).
// zero_line: >= 128 B of zeros in global memory (source of out-of-image taps and of rows beyond M).
// TAG: no effect on the code -- the throughput path instantiates the kernel once per OthelloNN layer (2 = conv2 ... 5 = fc1, 6 = fc2) so that
// every layer is its own symbol and `rocprofv3 --stats` lists conv3, conv4 and fc1 as separate rows instead of one k_gemm_h2<H2BigPP> average.
template <typename CF, int TAG = 0>
__global__ __launch_bounds__(CF::NT, 2) void k_gemm_h2(const uint4* __restrict__ in, const uint4* __restrict__ Wh,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       void* __restrict__ out, const int* __restrict__ d_count, H2Geom g,
                                                       int num_mt, const uint4* __restrict__ zero_line, int* __restrict__ flag,
                                                       const unsigned* __restrict__ lut_ids) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = CF::BM, BN = CF::BN, IA = CF::IA, IB = CF::IB;
    constexpr int RI = CF::TI * 2, RJ = CF::TJ * 2;           // 16-row / 16-column MFMA tiles per wave
    // XCD-aware tile order: the N/BN column tiles of one row tile run on the same XCD (ids b, b+8 share an L2)
    // split-K (g.ksplit > 1): the ksplit slices of one output tile are consecutive block ids of the same XCD
    // FEW row tiles (num_mt < 8: dense layers of batches up to ~1000 rows, every layer of the latency path): the mapping above would leave
    // XCDs num_mt .. 7 idle and make every working XCD stream the WHOLE weight matrix through its own L2 (one position: all blocks of a layer
    // on XCD 0 -- 32 of 256 CUs; fc1 at 512 rows: 4 XCDs, 134 MB of HBM reads for a 33.5 MB matrix).  There the WEIGHT slices q = (column
    // tile, k-slice) go round-robin over the XCDs and the row tiles of one slice share its XCD: every weight byte leaves HBM once, every XCD
    // works.  Same blocks, same sums -- only where they run changes.
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
    const int wm = wave / CF::WN, wn = wave % CF::WN;
    const int rowq = g.Cin >> 2;                             // uint4 (16 B) units per input pixel row: Cin/8 groups * 2
    const int wrowq = g.K >> 2;                              // uint4 units per weight row

    // staging map: wave w, DMA instruction i fills LDS rows (w*I + i)*8 .. +7 of the operand tile; lane l -> row
    // +(l>>3), physical chunk l&7, which holds logical chunk (l&7) ^ ((row>>1)&7) of that row's 128-byte k-slice
    // Ping-pong configs (CF::PP) stage the tile in "early" and "late" pieces instead: instructions 0,1 of a wave fill
    // rows of the FIRST half of every wave tile (A: the first RI/2 16-row blocks of each wave row; B: the first RJ/2 of
    // each wave column), instructions 2,3 rows of the second half -- the main loop reads the halves in different phases.
    // (192-row tile: 3 A pieces per wave -- piece 0 early, piece 2 late, piece 1 early for waves 0-3 and late for waves 4-7)
    // (128-row tile: 2 A pieces per wave -- piece 0 early, piece 1 late)
    auto a_late = [&](int i) -> int { return IA == 4 ? (i >> 1) : IA == 2 ? i : (i == 0 ? 0 : i == 2 ? 1 : (wave >= 4 ? 1 : 0)); };
    auto a_row0 = [&](int i) -> int {
        if constexpr (!CF::PP) return (wave * IA + i) * 8;
        else {
            constexpr int HR = BM / 4, HBLK = HR / 8;            // rows / 8-row blocks in one half of a wave row
            const int hb = IA == 4 ? 2 * wave + (i & 1) : IA == 2 ? wave : (i == 1 ? 8 + (wave & 3) : wave);
            return (hb / HBLK) * (BM / 2) + a_late(i) * HR + (hb % HBLK) * 8;
        }
    };
    auto b_row0 = [&](int i) -> int {
        if constexpr (!CF::PP) return (wave * IB + i) * 8;
        else { const int hb = 2 * wave + (i & 1); return (hb >> 2) * (BN / 4) + (i >> 1) * (BN / 8) + (hb & 3) * 8; }
    };
    long long aidx[IA];      // uint4 index of (row's input pixel at tap (0,0), logical chunk)
    unsigned amask[IA];
    unsigned bidx[IB];
    // CF::LUT: lut[tap * BM + row] = table row of (tile row, tap) -- the pattern id of the tap's pixel, or the zero row
    unsigned short* lut = reinterpret_cast<unsigned short*>(smem + CF::LDS);
    const int l8 = lane >> 3, lcb = (lane & 7) ^ ((l8 >> 1) & 1);   // piece rows start at multiples of 8: h2_swz(row0 + l8) = (row0 & 8 ? 6 : 0) | (l8 >> 1 & 1)
    if constexpr (CF::LUT) {
        for (int e = tid; e < 9 * BM; e += CF::NT) {
            const int tap = e / BM, row = e - tap * BM;
            const long long m = (long long)mt * BM + row;
            unsigned id = OZ_LUT_PATTERNS;
            if (m < M) {         // padded id boards (k_lut_ids): cell (iy + 1, ix + 1), the border = the zero row
                const int b = (int)(m / P), pix = (int)(m % P), iy = pix / g.Hout - g.pad + tap / 3, ix = pix % g.Hout - g.pad + tap % 3, Wp = g.Hin + 2;
                id = lut_ids[(size_t)b * Wp * Wp + (iy + 1) * Wp + ix + 1];
            }
            lut[e] = (unsigned short)id;
        }
        __syncthreads();
    }
    // source of A piece i for (tap, channel slice) given the piece's table row (LUT configs)
    auto lut_src = [&](int i, unsigned id, int slice) -> const uint4* {
        return in + (id * (unsigned)rowq + (unsigned)(lcb ^ (((a_row0(i) >> 3) & 1) * 6)) + (unsigned)(slice * 8));
    };
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        if constexpr (CF::LUT) { aidx[i] = 0; amask[i] = 0; continue; }
        const int row = a_row0(i) + (lane >> 3);
        const int lc = (lane & 7) ^ h2_swz(row);
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
        const int row = b_row0(i) + (lane >> 3);
        const int lc = (lane & 7) ^ h2_swz(row);
        bidx[i] = (unsigned)(nt * BN + row) * (unsigned)wrowq + (unsigned)lc;
    }
    const uint4* zsrc = zero_line + (lane & 7);
    const int nk_all = g.K / H2_BK, kbeg = (int)((long long)nk_all * ks / g.ksplit), nk = (int)((long long)nk_all * (ks + 1) / g.ksplit);

    auto stage = [&](int kt_raw, int buf) {
        const int kt = kt_raw < nk ? kt_raw : nk - 1;        // past the end: re-stage the last tile (branch-free loop body)
        const int slice = kt / g.taps, tap = kt - slice * g.taps;         // tap-inner k order
        const long long toff = ((long long)(tap / 3) * g.Hin + (tap % 3)) * rowq + slice * 8;    // 32 ch = 8 uint4
        unsigned char* la = smem + (size_t)buf * CF::BUF;
        unsigned char* lb = smem + (size_t)buf * CF::BUF + CF::TILEA;
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const uint4* ga;
            if constexpr (CF::LUT) ga = lut_src(i, lut[tap * BM + a_row0(i) + l8], slice);
            else ga = ((amask[i] >> tap) & 1) ? in + (aidx[i] + toff) : zsrc;
            __builtin_amdgcn_global_load_lds((h2_gptr)ga, (h2_lptr)(la + a_row0(i) * 128), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            const uint4* gb = Wh + bidx[i] + kt * 8;
            __builtin_amdgcn_global_load_lds((h2_gptr)gb, (h2_lptr)(lb + b_row0(i) * 128), 16, 0, 0);
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
    const int swz = h2_swz(r16);                             // tile bases are multiples of 16 rows
    const int oh1 = ((2 * kg) ^ swz) * 16, oh2 = ((2 * kg + 1) ^ swz) * 16;

    if constexpr (CF::PP && CF::PHASES == 1) {
        // ---- 1-phase ping-pong main loop (128 x 256 tile, three LDS stages): per k-tile t and wave
        //   L: all fragments of tile t (8 + 8 ds_read_b128);  DMA [A, B of tile t + 2] (2 + 4 instructions);  vmcnt(6): this wave's pieces of tile t + 1 have
        //      landed;  lgkmcnt(0);  barrier            M: 48 MFMAs (product-major);  barrier
        // The two wave rows run one barrier apart (one in L while the other is in M).  Stage (t + 2) % 3 = stage (t - 1) % 3 was last read in L(t - 1),
        // which every wave has closed (lgkmcnt(0) + barrier) before any wave enters L(t); a piece issued in L(t) is complete at the end of its wave's
        // L(t + 1) and is read in L(t + 2), after a barrier both rows have passed.
        static_assert(!CF::LUT && IA == 2 && IB == 4 && RI == 4 && RJ == 4 && CF::STAGES == 3, "1-phase ping-pong loop: 128 x 256 tile, 8 waves, 3 stages");
        constexpr int KEEP = IA + IB;
        auto put_tile = [&](int ktc, int slice, int tap, unsigned char* st) {
            const int dy = (tap * 11) >> 5, dx = tap - 3 * dy;                                      // tap / 3, tap % 3 for tap < 9
            const long long toff = ((long long)dy * g.Hin + dx) * rowq + slice * 8;
#pragma unroll
            for (int i = 0; i < IA; ++i) {
                const uint4* ga = ((amask[i] >> tap) & 1) ? in + (aidx[i] + toff) : zsrc;
                __builtin_amdgcn_global_load_lds((h2_gptr)ga, (h2_lptr)(st + a_row0(i) * 128), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < IB; ++i)
                __builtin_amdgcn_global_load_lds((h2_gptr)(Wh + bidx[i] + ktc * 8), (h2_lptr)(st + CF::TILEA + b_row0(i) * 128), 16, 0, 0);
        };
        int k2 = kbeg + 2 < nk ? kbeg + 2 : nk - 1, slice2 = k2 / g.taps, tap2 = k2 - slice2 * g.taps;      // tile kt + 2 (past the end: the last again)
        stage(kbeg, 0);
        stage(kbeg + 1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KEEP) : "memory");      // tile kbeg has landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (wm == 1) __builtin_amdgcn_s_barrier();           // stagger: wave row 1 is one barrier behind
        f16x8 fa1[RI], fa2[RI], fb1[RJ], fb2[RJ];
        int buf = 0;
        for (int kt = kbeg; kt < nk; ++kt) {
            unsigned char* la_cur = smem + (size_t)buf * CF::BUF;
            const int buf2 = buf == 0 ? 2 : buf - 1;                         // (buf + 2) % 3
            const unsigned char* At = la_cur + (wm * RI * 16 + r16) * 128;
            const unsigned char* Bt = la_cur + CF::TILEA + (wn * RJ * 16 + r16) * 128;
#pragma unroll
            for (int i = 0; i < RI; ++i) {
                fa1[i] = *reinterpret_cast<const f16x8*>(At + i * 16 * 128 + oh1);
                fa2[i] = *reinterpret_cast<const f16x8*>(At + i * 16 * 128 + oh2);
            }
#pragma unroll
            for (int j = 0; j < RJ; ++j) {
                fb1[j] = *reinterpret_cast<const f16x8*>(Bt + j * 16 * 128 + oh1);
                fb2[j] = *reinterpret_cast<const f16x8*>(Bt + j * 16 * 128 + oh2);
            }
            put_tile(k2, slice2, tap2, smem + (size_t)buf2 * CF::BUF);
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KEEP) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < RI; ++i)
#pragma unroll
                    for (int j = 0; j < RJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 0 ? fa2[i] : fa1[i], p == 1 ? fb2[j] : fb1[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            buf = buf == 2 ? 0 : buf + 1;
            if (k2 + 1 < nk) { ++k2; if (++tap2 == g.taps) { tap2 = 0; ++slice2; } }
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();           // re-align the two wave rows
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every piece has landed before the epilogue reuses the LDS
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    } else if constexpr (CF::PP && CF::PHASES == 2) {
        // ---- 2-phase ping-pong main loop (192 / 128 x 256 tiles): the same two wave rows half a phase apart, but a k-tile is two phases
        // instead of four -- (m0, m1) x n0, then (m0, m1) x n1 -- so the 12 / 18 MFMAs of a quadrant, too short to cover an L section and
        // its two barriers (the 4-phase loop on these tiles ran at 0.63 / 0.78 of its MFMA time), become 24 / 36:
        //   phase 1  L: A m0, m1 of tile t (4 * HA ds_reads);  DMA [Bl(t + 1)] [Be(t + 2)];  M: quadrants (m0, n0) (m1, n0)
        //   phase 2  L: B n1 of tile t, B n0 of tile t + 1 (8 ds_reads);  DMA [A(t + 2)];     M: quadrants (m0, n1) (m1, n1)
        // Every output element still receives its products in the same order (k-tile by k-tile, the three plane products in order).
        // DMA schedule: a region of the two-stage buffer is refilled in the phase after its last ds_read -- the L section ends with
        // lgkmcnt(0) BEFORE the barrier here, so every wave's reads of phase q are complete before any wave issues phase q + 1's DMA --
        // and is read two phases (= one k-tile) after its issue; all but the 4 + IA youngest pieces are complete at every L-section end.
        // Measured (conv3, one MI355X): 192 rows 234 -> 177 us at 512 leaves (one grid round), 1371 -> 1212 us at 4096 (6 rounds); 128 rows
        // 174 -> 151 us at 430 leaves.  The 256-row tile stays on the 4-phase loop: its m0 x (n0, n1) | m1 x (n0, n1) two-phase form (the
        // register file has no room for this one) measured -2 % in one round and +1 % at 3640 leaves, and deeper DMA schedules of the 4-phase
        // loop (a group issued 3 .. 5 phases ahead instead of 2, refilling each region right after its last read) 0 .. +16 %.
        static_assert(!CF::LUT && (IA == 3 || IA == 2) && IB == 4 && RI % 2 == 0 && RJ == 4 && CF::STAGES == 2, "2-phase ping-pong loop: 192 / 128 x 256 tile, 8 waves");
        constexpr int KEEP = 4 + IA;
        auto put_a = [&](int i, int slice, int tap, unsigned char* la) {
            const int dy = (tap * 11) >> 5, dx = tap - 3 * dy;                                      // tap / 3, tap % 3 for tap < 9
            const uint4* ga = ((amask[i] >> tap) & 1) ? in + (aidx[i] + ((long long)dy * g.Hin + dx) * rowq + slice * 8) : zsrc;
            __builtin_amdgcn_global_load_lds((h2_gptr)ga, (h2_lptr)(la + a_row0(i) * 128), 16, 0, 0);
        };
        auto put_b = [&](int i, int ktc, unsigned char* lb) {
            __builtin_amdgcn_global_load_lds((h2_gptr)(Wh + bidx[i] + ktc * 8), (h2_lptr)(lb + b_row0(i) * 128), 16, 0, 0);
        };
        int k1 = kbeg + 1 < nk ? kbeg + 1 : nk - 1, slice1 = k1 / g.taps, tap1 = k1 - slice1 * g.taps;      // tile kt + 1, kt + 2 (past the end: the last again)
        int k2 = kbeg + 2 < nk ? kbeg + 2 : nk - 1, slice2 = k2 / g.taps, tap2 = k2 - slice2 * g.taps;
        stage(kbeg, 0);
        {   // what phases 1 and 2 of the tile before the first would have issued: [Be] [A] of tile kbeg + 1
            unsigned char* st1 = smem + CF::BUF;
            put_b(0, k1, st1 + CF::TILEA); put_b(1, k1, st1 + CF::TILEA);
#pragma unroll
            for (int i = 0; i < IA; ++i) put_a(i, slice1, tap1, st1);
        }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 + IA) : "memory");     // tile kbeg has landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        f16x8 fa1[RI], fa2[RI], fb1[4], fb2[4];
        {   // B n0 of the first tile (later tiles get it in phase 2 of the tile before)
            const unsigned char* Bt0 = smem + CF::TILEA + (wn * RJ * 16 + r16) * 128;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                fb1[j] = *reinterpret_cast<const f16x8*>(Bt0 + j * 16 * 128 + oh1);
                fb2[j] = *reinterpret_cast<const f16x8*>(Bt0 + j * 16 * 128 + oh2);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // every wave's prologue reads of B n0 (stage 0) are complete before ANY wave issues phase 1's Be(kbeg + 2) into that region: without this
        // barrier wave row 0 would go straight into phase 1 while wave row 1 -- whose stagger barrier below pairs with row 0's l_end, AFTER row 0's
        // DMA issue -- could still be reading those rows (a narrow window, one k-slice of >= 3 tiles; the loop's own rule, restored)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (wm == 1) __builtin_amdgcn_s_barrier();           // stagger: wave row 1 is one barrier behind
        for (int kt = kbeg; kt < nk; ++kt) {
            const int buf = (kt - kbeg) & 1;
            unsigned char* la_cur = smem + (size_t)buf * CF::BUF;           // stage of tile kt (and of tile kt + 2)
            unsigned char* la_oth = smem + (size_t)(buf ^ 1) * CF::BUF;     // stage of tile kt + 1
            const unsigned char* At = la_cur + (wm * RI * 16 + r16) * 128;
            const unsigned char* Bt = la_cur + CF::TILEA + (wn * RJ * 16 + r16) * 128;
            const unsigned char* Btn = la_oth + CF::TILEA + (wn * RJ * 16 + r16) * 128;
            auto ldb = [&](int half, const unsigned char* base) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    fb1[half * 2 + j] = *reinterpret_cast<const f16x8*>(base + (half * 2 + j) * 16 * 128 + oh1);
                    fb2[half * 2 + j] = *reinterpret_cast<const f16x8*>(base + (half * 2 + j) * 16 * 128 + oh2);
                }
            };
            auto l_end = [&]() {
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KEEP) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            };
            auto mma = [&](int nh) {                         // (m0, nh) (m1, nh): 3 * 2 * RI MFMAs, product-major
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int i = 0; i < RI; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            f32x4v& c = acc[i][nh * 2 + j];
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 0 ? fa2[i] : fa1[i], p == 1 ? fb2[nh * 2 + j] : fb1[nh * 2 + j], c, 0, 0, 0);
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
            for (int i = 0; i < RI; ++i) {
                fa1[i] = *reinterpret_cast<const f16x8*>(At + i * 16 * 128 + oh1);
                fa2[i] = *reinterpret_cast<const f16x8*>(At + i * 16 * 128 + oh2);
            }
            put_b(2, k1, la_oth + CF::TILEA); put_b(3, k1, la_oth + CF::TILEA); put_b(0, k2, la_cur + CF::TILEA); put_b(1, k2, la_cur + CF::TILEA);
            l_end(); mma(0); m_end();
            // phase 2
            ldb(1, Bt); ldb(0, Btn);
#pragma unroll
            for (int i = 0; i < IA; ++i) put_a(i, slice2, tap2, la_cur);
            l_end(); mma(1); m_end();
            k1 = k2; slice1 = slice2; tap1 = tap2;
            if (k2 + 1 < nk) { ++k2; if (++tap2 == g.taps) { tap2 = 0; ++slice2; } }
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();           // re-align the two wave rows
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every piece has landed before the epilogue reuses the LDS
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    } else if constexpr (CF::PP) {
        // ---- 4-phase ping-pong main loop (block 256 x 256, 8 waves = 2 wave rows x 4 wave columns).
        // A k-tile is processed as four quadrants of the 128 x 64 wave tile, (m0,n0) (m0,n1) (m1,n0) (m1,n1); phase q =
        //   L section: ds_read the half-fragments the next MFMAs need (8 / 4 / 8 / 4 reads: A m0 | B n1 | A m1 | B n0 of the
        //              NEXT tile, into the registers quadrant 3 has just released), issue 2 of the 8 LDS-DMA pieces of the
        //              NEXT k-tile, s_waitcnt vmcnt(4), s_barrier
        //   M section: lgkmcnt(0), 24 MFMAs, s_barrier.
        // Wave row 1 runs one barrier behind wave row 0, and every SIMD holds one wave of each row: while one is in its
        // M section the other is in its L section, so LDS reads, DMA issue and address arithmetic hide under MFMAs.
        // LDS-DMA ordering (MI355X_MICROARCH.md: a ds_read sees DMA data only after the ISSUING waves' counted vmcnt and
        // a barrier the reader has passed; with the stagger one barrier more): pieces read in phase q are retired by every
        // wave's vmcnt in the L section of phase q-1.  Issue order per tile [Be0 Be1][Ae0 Ae1][Bl0 Bl1][Al0 Al1]
        // (e = first half, l = second half of the wave tiles): each pair is read three phases after its issue (B n0 in
        // phase 4, A m0 in the next phase 1, B n1 in phase 2, A m1 in phase 3), and at every L section all but the 4
        // youngest pieces are complete -- exactly the pairs the next phase reads.  A region is restaged four or more
        // phases after its last ds_read.
        // 192-row tile (IA = 3): a wave issues 7 pieces per tile, 2 or 1 of them per A phase depending on the wave, and
        // vmcnt(3) (the stricter of the two per-wave counts) retires what the next phase reads.
        static_assert((IA == 4 || IA == 3) && IB == 4 && RI % 2 == 0 && RJ == 4, "ping-pong loop: 256 x 256 or 192 x 256 tile, 8 waves");
        constexpr int HA = RI / 2;                           // 16-row A blocks per half of the wave tile
        stage(kbeg, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        f16x8 fa1[HA], fa2[HA], fb1[4], fb2[4];
        unsigned aid[IA] = {};
        {   // B n0 of the first tile (later tiles get it in phase 4 of the tile before)
            const unsigned char* Bt0 = smem + CF::TILEA + (wn * RJ * 16 + r16) * 128;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                fb1[j] = *reinterpret_cast<const f16x8*>(Bt0 + j * 16 * 128 + oh1);
                fb2[j] = *reinterpret_cast<const f16x8*>(Bt0 + j * 16 * 128 + oh2);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (wm == 1) __builtin_amdgcn_s_barrier();           // stagger: wave row 1 is one barrier behind
        // (slice, tap) of the tile being staged, advanced incrementally: no integer division in the loop
        int ktn = kbeg + 1 < nk ? kbeg + 1 : nk - 1;
        int slice_n = ktn / g.taps, tap_n = ktn - slice_n * g.taps;
        for (int kt = kbeg; kt < nk; ++kt) {
            const int buf = (kt - kbeg) & 1;
            const int slice = slice_n, tap = tap_n, dy = (tap * 11) >> 5, dx = tap - 3 * dy;            // tap / 3, tap % 3 for tap < 9
            const int ktc = ktn;
            const long long toff = ((long long)dy * g.Hin + dx) * rowq + slice * 8;
            unsigned char* la = smem + (size_t)(buf ^ 1) * CF::BUF;
            unsigned char* lb = la + CF::TILEA;
            const unsigned char* At = smem + (size_t)buf * CF::BUF + (wm * RI * 16 + r16) * 128;
            const unsigned char* Bt = smem + (size_t)buf * CF::BUF + CF::TILEA + (wn * RJ * 16 + r16) * 128;
            auto dma_a = [&](int i) {
                const uint4* ga;
                if constexpr (CF::LUT) ga = lut_src(i, aid[i], slice);
                else ga = ((amask[i] >> tap) & 1) ? in + (aidx[i] + toff) : zsrc;
                __builtin_amdgcn_global_load_lds((h2_gptr)ga, (h2_lptr)(la + a_row0(i) * 128), 16, 0, 0);
            };
            // LUT: the table rows of the tile being staged, read in phase 1 (retired by its lgkmcnt(0)), used in phases 2 and 4
            auto ld_ids = [&]() {
                if constexpr (CF::LUT) {
#pragma unroll
                    for (int i = 0; i < IA; ++i) aid[i] = lut[tap * BM + a_row0(i) + l8];
                }
            };
            auto dma_b = [&](int i) {
                __builtin_amdgcn_global_load_lds((h2_gptr)(Wh + bidx[i] + ktc * 8), (h2_lptr)(lb + b_row0(i) * 128), 16, 0, 0);
            };
            auto lda = [&](int half) {
#pragma unroll
                for (int i = 0; i < HA; ++i) {
                    fa1[i] = *reinterpret_cast<const f16x8*>(At + (half * HA + i) * 16 * 128 + oh1);
                    fa2[i] = *reinterpret_cast<const f16x8*>(At + (half * HA + i) * 16 * 128 + oh2);
                }
            };
            auto ldb = [&](int half, const unsigned char* base) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    fb1[half * 2 + j] = *reinterpret_cast<const f16x8*>(base + (half * 2 + j) * 16 * 128 + oh1);
                    fb2[half * 2 + j] = *reinterpret_cast<const f16x8*>(base + (half * 2 + j) * 16 * 128 + oh2);
                }
            };
            const unsigned char* Btn = smem + (size_t)(buf ^ 1) * CF::BUF + CF::TILEA + (wn * RJ * 16 + r16) * 128;
            // end of an L section: all but the `keep` youngest DMA pieces of this wave have landed, then the barrier
            auto l_end = [&](auto keep) {
                constexpr int K = decltype(keep)::value;
                if constexpr (K == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else if constexpr (K == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if constexpr (K == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else if constexpr (K == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if constexpr (K == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            };
            // quadrant (m half, n half): 3 * 2 * HA MFMAs, product-major (2 * HA independent accumulators between two
            // MFMAs of the same one; the chained order measured the same).  s_setprio around the cluster measured 2 % slower.
            auto mma = [&](int mh, int nh) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int i = 0; i < HA; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            f32x4v& c = acc[mh * HA + i][nh * 2 + j];
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 0 ? fa2[i] : fa1[i], p == 1 ? fb2[nh * 2 + j] : fb1[nh * 2 + j], c, 0, 0, 0);
                        }
                __builtin_amdgcn_sched_barrier(0);
            };
            auto m_end = [&]() {                             // end of an M section: the closing barrier
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            };
            using std::integral_constant;
            auto dma_a_early = [&]() { dma_a(0); if (IA == 4 || wave < 4) dma_a(1); };
            auto dma_a_late = [&]() { dma_a(IA - 1); if (IA == 4) dma_a(2); else if (wave >= 4) dma_a(1); };
            constexpr int KEEP = IA == 4 ? 4 : 3;
            lda(0); ld_ids(); dma_b(0); dma_b(1); l_end(integral_constant<int, KEEP>{}); mma(0, 0); m_end();     // phase 1
            ldb(1, Bt); dma_a_early(); l_end(integral_constant<int, KEEP>{}); mma(0, 1); m_end();                // phase 2
            lda(1); dma_b(2); dma_b(3); l_end(integral_constant<int, KEEP>{}); mma(1, 0); m_end();               // phase 3
            ldb(0, Btn); dma_a_late(); l_end(integral_constant<int, KEEP>{}); mma(1, 1); m_end();                // phase 4 (B n0 of the next tile)
            if (ktn + 1 < nk) { ++ktn; if (++tap_n == g.taps) { tap_n = 0; ++slice_n; } }    // past the end: re-stage the last tile
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();           // re-align the two wave rows
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every piece has landed before the epilogue reuses the LDS
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    } else {
    constexpr int S = CF::STAGES;
    if constexpr (S > 2) {
#pragma unroll
        for (int p = 0; p < S - 1; ++p) stage(kbeg + p, p);  // S - 1 k-tiles in flight (past the end: the last tile again, harmless)
    } else {
    stage(kbeg, 0);
    __syncthreads();                                         // drains the DMA (vmcnt(0)) and publishes the tile
    }
    for (int kt = kbeg; kt < nk; ++kt) {
        const int buf = S > 2 ? (kt - kbeg) % S : (kt - kbeg) & 1;
        if constexpr (S > 2) {
            // tile kt has landed once at most (S - 2) tiles' worth of younger DMA instructions of this wave are outstanding;
            // the barrier publishes it and also proves that every wave is done reading the buffer staged next (read in kt - 1)
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 2) * (IA + IB)) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stage(kt + S - 1, (kt - kbeg + S - 1) % S);
        } else {
        stage(kt + 1, buf ^ 1);
        }
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
        {   // the next tile's DMA instructions are spread over the first half of the MFMA stream
            constexpr int NDMA = IA + IB, NMF = RI * RJ * 3, PER = NMF / (2 * NDMA);
#pragma unroll
            for (int q = 0; q < NDMA; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);      // MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);        // VMEM read (global_load_lds)
            }
        }
        if constexpr (S == 2) {
        __syncthreads();
        }
    }
    if constexpr (S > 2) {                                   // the re-staged tail tiles have landed before the epilogue reuses the LDS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    }   // !CF::PP

    // ---- epilogue.  C/D layout of 16x16: col = lane&15, row = (lane>>4)*4 + reg
    if (g.ksplit > 1) {      // raw fp32 partial sums of this k-slice into slab ks; k_splitk_reduce_h2 finishes the layer
        float* o = reinterpret_cast<float*>(out) + (size_t)ks * g.slab;
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            const int col = nt * BN + wn * RJ * 16 + j * 16 + r16;
#pragma unroll
            for (int i = 0; i < RI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long m = (long long)mt * BM + wm * RI * 16 + i * 16 + kg * 4 + r;
                    if (m < M) o[(size_t)m * g.N + col] = acc[i][j][r];
                }
        }
        return;
    }
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
    // h2 output: each wave transposes its tile through its own 16 KB LDS slice, 64 rows x 64 channels at a time:
    // slice[row][group(8)][plane(2)][8 halfs] = 256 B per row; then 16-byte chunks go out, 16 lanes per row.
    static_assert(RJ == 4, "the h2 epilogue assumes a 64-channel wave tile");
    // the high-side guard as a RUNNING maximum of |v| (one register): written as `over |= |v| > 65504` the compiler turned the chain into one
    // max3 tree at the END of the epilogue and kept every v alive for it -- the kernel's 45 spilled registers, 184 B of scratch per thread,
    // 0.095 GB of extra writes (and reads) per conv3 launch (round 5)
    float amax = 0.f;
    constexpr int PB = CF::PB;
    _Float16* slice = reinterpret_cast<_Float16*>(smem + wave * CF::SLICE);
    uint4* o = reinterpret_cast<uint4*>(out);
    const int nq = g.N >> 2;                                // uint4 units per output row (N/8 groups * 2)
#pragma unroll
    for (int hh = 0; hh < RI / PB; ++hh) {
#pragma unroll
        for (int ii = 0; ii < PB; ++ii) {
            const int i = hh * PB + ii;
#pragma unroll
            for (int j = 0; j < RJ; ++j) {
                const int lc = j * 16 + r16;                 // column inside the wave tile
                const int col = nt * BN + wn * RJ * 16 + lc;
                const float sc = scale[col], sh = shift[col];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lr = ii * 16 + kg * 4 + r;     // row inside the pass
                    float v = fmaf(acc[i][j][r], sc, sh);
                    if (g.relu) v = fmaxf(v, 0.f);
                    amax = fmaxf(amax, fabsf(v));
                    _Float16 h1, h2;
                    h2_split(v, h1, h2);
                    _Float16* p = slice + lr * 128 + (lc >> 3) * 16 + (lc & 7);
                    p[0] = h1; p[8] = h2;
                }
                asm volatile("" : "+v"(amax));               // the maximum so far is MATERIALISED here (max is reassociable: without this the tree is built at the end)
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): this wave's LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        // PB*16 rows x 256 B in chunks of 16 B; lane l takes chunks l, l+64, ...: 16 consecutive lanes = one row
#pragma unroll
        for (int c = 0; c < PB * 4; ++c) {
            const int q = c * 64 + lane, lr = q >> 4, cq = q & 15;
            const long long m = (long long)mt * BM + wm * RI * 16 + hh * PB * 16 + lr;
            const uint4 val = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(slice) + lr * 256 + cq * 16);
            if (m < M) o[(size_t)m * nq + ((nt * BN + wn * RJ * 16) >> 2) + cq] = val;
            if (g.low.cnt) {     // low-side guard: the 16 lanes of a row hold its 64-channel slice; the even chunks are the h1 planes
                float vmax = 0.f;
                if ((cq & 1) == 0) {
                    const f16x8 h = *reinterpret_cast<const f16x8*>(&val);
#pragma unroll
                    for (int e = 0; e < 8; ++e) vmax = fmaxf(vmax, fabsf((float)h[e]));
                }
                vmax = h2_octet_max(vmax);
                vmax = fmaxf(vmax, __shfl_xor(vmax, 8, 64));
                if (cq == 0 && m < M && vmax > 0.f && vmax < g.low.thr) h2_low_report(g.low, m, g.N >> 6);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
    if (amax > H2_F16_MAX) atomicOr(flag, H2_FLAG_OVER);
}
