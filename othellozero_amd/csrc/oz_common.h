// oz_common.h -- shared host/device helpers of libothellozero_amd (gfx950).
//
// Board representation everywhere in the library: two uint64 bitboards per
// position, bit index = row*8 + col for EVERY board size n in {4,6,8}; the
// n x n board occupies the top-left corner of the 8x8 grid, squares outside it
// are never occupied and never playable.  "own"/"opp" = reference channel 0 /
// channel 1 of a mover-canonical state (othelo_mcts.py:22-26,43-49);
// "black"/"white" = absolute colours of a game (Othello/__init__.py:22-25).
// NN action index = row*n + col (Net/NNet.py:86 reshape order).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define OZ_HD __host__ __device__ __forceinline__

// ---------------------------------------------------------------- integer mixers
// (identical formulas in oracle/oz_oracle.c and tests/golden/gen_golden.py)
OZ_HD uint64_t oz_sm64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
OZ_HD uint64_t oz_rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

enum { OZ_RNG_COIN = 0, OZ_RNG_EXPLORE = 1, OZ_RNG_TIE = 2 };
// counter-based stream replacing random.random / np.random.choice / random.choice
// (training.py:51,56; othelo_mcts.py:59): keyed (seed, global game id, ply, purpose)
OZ_HD uint64_t oz_rng(uint64_t seed, uint64_t game, uint64_t move, uint64_t stream) {
    uint64_t a = oz_sm64(seed + 0x632BE59BD9B4E019ULL * game);
    return oz_sm64(a ^ (move * 0x9E3779B97F4A7C15ULL) ^ (stream * 0xD1B54A32D192ED03ULL));
}
OZ_HD double oz_rng_unit(uint64_t u) { return (double)(u >> 11) * (1.0 / 9007199254740992.0); }

OZ_HD uint64_t oz_stub_h(uint64_t own, uint64_t opp, uint64_t salt, uint64_t i) {
    return oz_sm64(oz_sm64(own ^ salt) ^ oz_rotl64(opp, 29) ^ ((i + 1) * 0xD6E8FEB86659FD93ULL));
}

// ---------------------------------------------------------------- bitboard rules
OZ_HD uint64_t oz_valid_mask(int n) {
    uint64_t row = (1ULL << n) - 1ULL, m = 0;
    for (int r = 0; r < n; ++r) m |= row << (8 * r);
    return m;
}

#define OZ_NOT_COL0 0xFEFEFEFEFEFEFEFEULL
#define OZ_NOT_COL7 0x7F7F7F7F7F7F7F7FULL

// one step along direction D (0..7); rows grow with bit index (south = +8), cols east = +1
template <int D> OZ_HD uint64_t oz_shift(uint64_t x) {
    if (D == 0) return x << 8;                       // S  (+1, 0)
    if (D == 1) return x >> 8;                       // N  (-1, 0)
    if (D == 2) return (x << 1) & OZ_NOT_COL0;       // E  ( 0,+1)
    if (D == 3) return (x >> 1) & OZ_NOT_COL7;       // W  ( 0,-1)
    if (D == 4) return (x << 9) & OZ_NOT_COL0;       // SE (+1,+1)
    if (D == 5) return (x >> 9) & OZ_NOT_COL7;       // NW (-1,-1)
    if (D == 6) return (x << 7) & OZ_NOT_COL7;       // SW (+1,-1)
    return (x >> 7) & OZ_NOT_COL0;                   // NE (-1,+1)
}

template <int D> OZ_HD uint64_t oz_legal_dir(uint64_t own, uint64_t opp) {
    uint64_t t = oz_shift<D>(own) & opp;
    t |= oz_shift<D>(t) & opp; t |= oz_shift<D>(t) & opp; t |= oz_shift<D>(t) & opp;
    t |= oz_shift<D>(t) & opp; t |= oz_shift<D>(t) & opp;
    return oz_shift<D>(t);
}

// R4 (Othello/__init__.py:208-214): legal-move set of the side holding `own`.
// Legality is identical to standard Othello, so the standard flood works.
OZ_HD uint64_t oz_legal(uint64_t own, uint64_t opp, uint64_t valid) {
    uint64_t m = oz_legal_dir<0>(own, opp) | oz_legal_dir<1>(own, opp) | oz_legal_dir<2>(own, opp) |
                 oz_legal_dir<3>(own, opp) | oz_legal_dir<4>(own, opp) | oz_legal_dir<5>(own, opp) |
                 oz_legal_dir<6>(own, opp) | oz_legal_dir<7>(own, opp);
    return m & ~(own | opp) & valid;
}

// R3 (Othello/__init__.py:216-235) along one ray, INCLUDING the reference's
// "flip-through": every opponent disc that lies before the LAST own disc of the
// contiguous occupied run starting at the neighbour is flipped.
//   D = walking direction, B = its opposite.
template <int D, int B> OZ_HD uint64_t oz_flips_dir(uint64_t bit, uint64_t own, uint64_t opp) {
    uint64_t x = oz_shift<D>(bit) & opp;              // first neighbour must be an opponent disc
    if (!x) return 0;
    const uint64_t occ = own | opp;
    uint64_t seg = x;                                 // contiguous occupied run from the neighbour
    seg |= oz_shift<D>(seg) & occ; seg |= oz_shift<D>(seg) & occ; seg |= oz_shift<D>(seg) & occ;
    seg |= oz_shift<D>(seg) & occ; seg |= oz_shift<D>(seg) & occ; seg |= oz_shift<D>(seg) & occ;
    uint64_t back = seg & own;                        // own discs inside the run ...
    back |= oz_shift<B>(back) & seg; back |= oz_shift<B>(back) & seg; back |= oz_shift<B>(back) & seg;
    back |= oz_shift<B>(back) & seg; back |= oz_shift<B>(back) & seg; back |= oz_shift<B>(back) & seg;
    return back & opp;                                // ... and every opponent disc before one of them
}

// flips of placing a disc of `own` on square sq (sq must be empty; no legality check, R5)
OZ_HD uint64_t oz_flips(uint64_t own, uint64_t opp, int sq) {
    const uint64_t bit = 1ULL << sq;
    return oz_flips_dir<0, 1>(bit, own, opp) | oz_flips_dir<1, 0>(bit, own, opp) |
           oz_flips_dir<2, 3>(bit, own, opp) | oz_flips_dir<3, 2>(bit, own, opp) |
           oz_flips_dir<4, 5>(bit, own, opp) | oz_flips_dir<5, 4>(bit, own, opp) |
           oz_flips_dir<6, 7>(bit, own, opp) | oz_flips_dir<7, 6>(bit, own, opp);
}

// R5 flip_board_squares (Othello/__init__.py:237-247)
OZ_HD void oz_apply(uint64_t& own, uint64_t& opp, int sq) {
    const uint64_t f = oz_flips(own, opp, sq);
    own |= f | (1ULL << sq);
    opp &= ~(f | (1ULL << sq));
}

// OthelloGame.play, Othello/__init__.py:136-159: flip, switch player; if the new mover has no move,
// either the game is over (nobody can move) or the turn passes back.  player: +1 BLACK, -1 WHITE.
OZ_HD void oz_game_play(uint64_t& black, uint64_t& white, int& player, int& finished, int sq, uint64_t valid) {
    if (player == 1) oz_apply(black, white, sq); else oz_apply(white, black, sq);
    player = -player;
    uint64_t mine = player == 1 ? black : white, theirs = player == 1 ? white : black;
    if (oz_legal(mine, theirs, valid) == 0) {
        if (oz_legal(theirs, mine, valid) == 0) finished = 1;
        else player = -player;
    }
}

OZ_HD int oz_popc(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}
OZ_HD int oz_ctz(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((unsigned long long)x) - 1;
#else
    return __builtin_ctzll(x);
#endif
}
// index of the k-th (0-based) set bit
OZ_HD int oz_kth_bit(uint64_t m, int k) {
    for (int i = 0; i < k; ++i) m &= m - 1;
    return oz_ctz(m);
}

// ---------------------------------------------------------------- error plumbing (host)
#include <stdio.h>
void oz_set_error(const char* fmt, ...);
// error codes: include/othellozero_amd.h (include it before this header)
#define OZ_HIP(call)                                                                     \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) {                                                          \
            oz_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return OZ_ERR_HIP;                                                            \
        }                                                                                 \
    } while (0)
#define OZ_REQUIRE(cond, ...)                                                             \
    do {                                                                                  \
        if (!(cond)) { oz_set_error(__VA_ARGS__); return OZ_ERR_ARG; }                    \
    } while (0)
