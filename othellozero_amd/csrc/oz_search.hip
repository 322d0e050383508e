// oz_search.hip -- PUCT search (K4-K7, K13), self-play and arena drivers for gfx950.
//
// Replaces MCTS/__init__.py:19-187 + othelo_mcts.py:9-88 (search) and training.py:26-72,
// agents.py:44-84 (drivers).  Design (MI355X-first, not a translation):
//   * every game owns one OthelloMCTS instance = one open-addressing table keyed by the exact
//     128-bit (own, opp) pair (replaces sha1, MCTS/__init__.py:7-16) plus a pool of fixed-stride node
//     records (header + visit / Q / P rows, see "tree storage"), all resident in HBM; nothing is rebuilt
//     between moves (sub-tree / transposition reuse);
//   * ONE WAVEFRONT PER GAME: lane == board square.  PUCT argmax (MCTS/__init__.py:65,168-170)
//     is a 64-lane max-reduction + ballot, lowest lane wins ties (== Python's first maximum in
//     ascending square order); a node record reaches LDS in one wave-wide load and the descent's
//     frontier stays in LDS; table probes are 64-wide ballots; the backup walks all path levels in
//     parallel (one lane per level);
//   * lock step: every active game advances exactly one simulation per step (simulations of one
//     game are sequential in the reference; no virtual loss), leaves that need the network are
//     compacted by ballot + prefix sum into one dense batch (deterministic slot order), evaluated
//     by one NN launch sequence, then expanded / backed up;
//   * float semantics that decide discrete outcomes are restated exactly (SURVEY.md R-FP):
//     U in float64 evaluated left to right with no FMA contraction, P normalised with NumPy's
//     pairwise order, Q accumulated in the dynamic type NumPy would use (q_mode).
#pragma clang fp contract(off)
#include <math.h>
#include <string.h>

#include "oz_internal.h"

enum { VT_INT = 0, VT_F32 = 1, VT_F64 = 2 };
#define OZ_MAX_DEPTH 64
#define OZ_TAG_F32 0x80000000u
#define OZ_NSTAT 5
enum { ST_SIMS = 0, ST_VISITS = 1, ST_EXPAND = 2, ST_TERMINAL = 3, ST_FALLBACK = 4 };
enum { EF_NODES = 1, EF_EDGES = 2, EF_DEPTH = 4, EF_NOMOVE = 8, EF_RECORDS = 16, EF_ROOT = 32, EF_CORRUPT = 64 };

// ---------------------------------------------------------------- tree storage (per game: one OthelloMCTS instance)
// One fixed-stride RECORD per node, addressed by the node index alone:
//     [ header 32 B: own u64 | opp u64 | Ns i32 | cnt i32 | pad ][ cnt edges x 24 B: N|tag u32, child i32, Q f64, P f64 ]
// edges in ascending square order of the legal set (rank = popcount of the legal bits below the square).  rec_bytes is a
// multiple of 64 and at most 1024, so ONE wave-wide 16-byte load (a lane per chunk) brings a whole node -- header and its
// visit / Q / P table -- into LDS in a single memory round trip, from where every lane (= board square) picks its edge.
// `child` caches the record index of the state the edge leads to (-1 = not expanded yet), so a descent follows indices and
// probes the hash table only where it leaves the known tree (the leaf) -- not at every level.  The record stride trades HBM
// capacity (22.6 GB of 288 for 4096 games x 6164 nodes; only the touched 64-byte sectors ever move) for one dependent
// memory round trip per tree level instead of four (table window, key compare, node header, edge rows).
// Hash table: 32-bit entries [tag 12 | index+1 20], 0 = empty; a 64-wide window is one 256-byte read and only tag matches
// (1 in 4096 false positives) go on to compare the 128-bit key in the record header.
struct OzEdge { uint32_t n_tag; int32_t child; double q; double p; };
#define OZ_REC_HDR 32
#define OZ_EDGE_BYTES 24
#define OZ_HT_IDX_BITS 20
#define OZ_HT_IDX_MASK ((1u << OZ_HT_IDX_BITS) - 1u)
static_assert(sizeof(OzEdge) == OZ_EDGE_BYTES, "edge layout");

struct MctsDev {
    int G, n, n2, node_cap, row_cap, rec_bytes, ht_cap;
    uint64_t valid;
    double c;
    int qmode;
    unsigned char* recs;       // [G][node_cap][rec_bytes]
    int* node_count;           // [G]
    uint32_t* ht;              // [G][ht_cap]
    int* root_node;            // [G] record index of the current root (-1: not looked up yet / not expanded)
    uint64_t *root_own, *root_opp;
    uint8_t* active;
    int* leaf_status;
    uint64_t *leaf_own, *leaf_opp, *leaf_legal;
    int *depth, *term_value, *leaf_slot, *leaf_fs;
    int2* path;                // [G][OZ_MAX_DEPTH] (node index, edge rank)
    uint64_t *batch_own, *batch_opp;
    int* batch_count;
    float *pi, *v;
    double* last_value;
    int* last_vtype;
    unsigned long long* stat;  // [G][OZ_NSTAT]
    int* error_flag;
    unsigned long long* eval_leaves;    // positions handed to the network so far (after cross-game de-duplication)
    int dedup;                 // k_compact: evaluate a board reached by several games in the same step once
    int batch_cap, batch_rot;  // k_compact (free-running driver): at most batch_cap slots (0: no cap), handed out in game order starting at game batch_rot
    EvalCacheDev ec;           // the network's persistent evaluation cache (buckets == 0: none)
    int* hit_count;            // leaves of this batch served from the cache: they take the pi / v rows G-1, G-2, ... (never the network's rows 0 ..)
    int* hit_entry;            // [G] cache entry of hit h
    unsigned batch_stamp;      // k_cache_insert: number of this batch (two inserts into one entry within a batch: the first wins)
};

__device__ __forceinline__ unsigned char* rec_ptr(const MctsDev& t, int g, int node) {
    return t.recs + ((size_t)g * t.node_cap + (size_t)node) * (size_t)t.rec_bytes;
}
__device__ __forceinline__ OzEdge* rec_edge(const MctsDev& t, int g, int node, int rank) {
    return reinterpret_cast<OzEdge*>(rec_ptr(t, g, node) + OZ_REC_HDR) + rank;
}
__device__ __forceinline__ int* rec_Ns(const MctsDev& t, int g, int node) {
    return reinterpret_cast<int*>(rec_ptr(t, g, node) + 16);
}

// per-wavefront LDS working set of the tree kernels (one wave = one game = one block of 64)
struct TreeLds {
    uint4 rec[64];             // the staged node record: header + visit / Q / P rows (<= 1024 B)
    int2 path[OZ_MAX_DEPTH];   // the frontier of this descent: (node, edge rank) per level
    double arr[64];            // expand: masked policy row for the pairwise sum
};

// ---------------------------------------------------------------- wave helpers (wave = 64 lanes)
__device__ __forceinline__ double wave_max_f64(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double y = __shfl_xor(x, off, 64);
        x = y > x ? y : x;
    }
    return x;
}
__device__ __forceinline__ int wave_max_i32(int x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        int y = __shfl_xor(x, off, 64);
        x = y > x ? y : x;
    }
    return x;
}
// Wave-uniform values (the position a wave stands on, node indices, the chosen square) belong in SGPRs: the bitboard
// arithmetic of the rules (legal set, flips: ~300 64-bit shift / and / or per tree level) then runs on the scalar unit --
// one instruction per 64-bit op per WAVE instead of two vector instructions per op per 16 lanes.  Loads through a pointer
// come back in vector registers, so the uniformity has to be stated.
// (the builtin returns a signed int: every half is cast to uint32_t before it is widened)
__device__ __forceinline__ uint32_t uni32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ int unii(int x) { return (int)__builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint64_t uni64(uint64_t x) {
    const uint32_t lo = uni32((uint32_t)x), hi = uni32((uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | (uint64_t)lo;
}
__device__ __forceinline__ int lane_get(int x, int lane) { return (int)__builtin_amdgcn_readlane(x, lane); }   // lane must be uniform

// one wave = one game: stores of one lane must have completed (and may not be reordered by the compiler) before the
// other lanes load the same locations in the next phase
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint64_t key_hash64(uint64_t own, uint64_t opp) {
    return oz_sm64(own * 0x2545F4914F6CDD1DULL ^ oz_sm64(opp));
}

// 64-wide linear probe.  Returns the node index or -1; *free_slot = first empty slot of the probe
// sequence (where an insert must go).  All lanes get the same results.
__device__ int ht_find(const MctsDev& t, int g, uint64_t own, uint64_t opp, int lane, int* free_slot) {
    const uint32_t* __restrict__ ht = t.ht + (size_t)g * t.ht_cap;
    const int cap = t.ht_cap, mask = cap - 1;
    const uint64_t hh = key_hash64(own, opp);
    int base = (int)((uint32_t)(hh >> 17) & (uint32_t)mask);
    const uint32_t tag = (uint32_t)(hh >> 52);
    for (int probed = 0; probed < cap; probed += 64, base = (base + 64) & mask) {
        const uint32_t e = ht[(base + lane) & mask];
        const bool empty = e == 0;
        const int idx = (int)(e & OZ_HT_IDX_MASK) - 1;
        bool match = false;
        if (!empty && (e >> OZ_HT_IDX_BITS) == tag && idx < t.node_cap) {
            const uint64_t* h = reinterpret_cast<const uint64_t*>(rec_ptr(t, g, idx));
            match = h[0] == own && h[1] == opp;
        }
        const uint64_t mb = __ballot(match), eb = __ballot(empty);
        if (mb) {
            const int ml = oz_ctz(mb);
            if (!eb || ml < oz_ctz(eb)) return lane_get(idx, ml);
        }
        if (eb) {
            *free_slot = (base + oz_ctz(eb)) & mask;
            return -1;
        }
    }
    *free_slot = -1;
    return -1;
}
__device__ __forceinline__ uint32_t ht_entry(uint64_t own, uint64_t opp, int node) {
    return ((uint32_t)(key_hash64(own, opp) >> 52) << OZ_HT_IDX_BITS) | (uint32_t)(node + 1);
}

// NumPy pairwise sum of a contiguous float64 vector, 8 <= len <= 128 (np.sum at MCTS/__init__.py:49-51)
__host__ __device__ inline double pairwise_sum(const double* a, int len) {
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    for (i = 8; i < len - (len % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < len; ++i) res += a[i];
    return res;
}

// ---------------------------------------------------------------- K4: select / descend
// MCTS.simulate down to the first terminal or unexpanded state (MCTS/__init__.py:39-44,58-67),
// get_next_state (othelo_mcts.py:43-49).  One wave per game; per level ONE wave-wide load stages the node's record in LDS.
// (body shared by k_select and the free-running k_advance; returns the status it stored in leaf_status[g])
__device__ __forceinline__ int select_body(const MctsDev& t, TreeLds& L, int g, int lane) {
    if (!t.active[g]) {
        if (lane == 0) t.leaf_status[g] = OZ_LEAF_IDLE;
        return OZ_LEAF_IDLE;
    }
    uint64_t own = uni64(t.root_own[g]), opp = uni64(t.root_opp[g]);
    int depth = 0, status, tval = 0, err = 0, fs = -1;
    int node = unii(t.root_node[g]);                            // record index of the state we stand on, -1 = unknown
    int pnode = -1, prank = 0;                                  // the edge we arrived through (to cache the child index)
    uint64_t legal = oz_legal(own, opp, t.valid);
    for (;;) {
        // `legal` = the mover's legal set of the state we stand on (of the root above; below, the move that led here computed it)
        if (legal == 0 && oz_legal(opp, own, t.valid) == 0) {       // is_terminal_state, othelo_mcts.py:28-29
            // get_state_reward: winner of the ch0 view, draw -> ch0; simulate returns -reward
            tval = oz_popc(own) >= oz_popc(opp) ? -1 : 1;
            status = OZ_LEAF_TERMINAL;
            break;
        }
        if (node < 0) {                                             // leaving the known tree: is the state in the table (transposition)?
            node = ht_find(t, g, own, opp, lane, &fs);
            if (node < 0) { status = OZ_LEAF_EVAL; break; }
            if (lane == 0) {
                if (pnode >= 0) rec_edge(t, g, pnode, prank)->child = node;
                else t.root_node[g] = node;
            }
        }
        if (legal == 0) { err = EF_NOMOVE; status = OZ_LEAF_IDLE; break; }   // reference: max() of empty list
        if (depth >= OZ_MAX_DEPTH) { err = EF_DEPTH; status = OZ_LEAF_IDLE; break; }
        if (node >= t.node_cap) { err = EF_CORRUPT; status = OZ_LEAF_IDLE; break; }   // never dereference an index outside the pool
        // stage the record: header + cnt edges, 16 bytes per lane
        const int cnt = oz_popc(legal);
        const int chunks = (OZ_REC_HDR + cnt * OZ_EDGE_BYTES + 15) >> 4;
        __syncthreads();                                            // the previous level's LDS reads are done
        if (lane < chunks) L.rec[lane] = reinterpret_cast<const uint4*>(rec_ptr(t, g, node))[lane];
        __syncthreads();
        const uint64_t* w = reinterpret_cast<const uint64_t*>(L.rec);
        const int Ns = unii((int)(uint32_t)w[2]);
        const bool is_legal = (legal >> lane) & 1;
        const int rank = oz_popc(legal & ((1ULL << lane) - 1ULL));
        double U = -INFINITY;
        int child = -1;
        if (is_legal) {
            const uint64_t w0 = w[4 + 3 * rank];
            const int N = (int)((uint32_t)w0 & ~OZ_TAG_F32);
            child = (int)(w0 >> 32);
            const double Q = __longlong_as_double((long long)w[5 + 3 * rank]), P = __longlong_as_double((long long)w[6 + 3 * rank]);
            const double bound = sqrt((double)Ns) / (double)(1 + N);       // MCTS/__init__.py:169
            U = Q + (t.c * P) * bound;                                     // :170, left to right
        }
        const double m = wave_max_f64(U);
        const int best = oz_ctz(__ballot(is_legal && U == m));             // first maximum
        const int brank = oz_popc(legal & ((1ULL << best) - 1ULL));
        if (lane == 0) L.path[depth] = make_int2(node, brank);
        ++depth;
        pnode = node; prank = brank;
        node = lane_get(child, best);
        oz_apply(own, opp, best);
        // swap only if the opponent can move (othelo_mcts.py:43-49); its legal set IS the next state's -- one flood per level, not two
        const uint64_t theirs = oz_legal(opp, own, t.valid);
        if (theirs != 0) { uint64_t s = own; own = opp; opp = s; legal = theirs; }
        else legal = oz_legal(own, opp, t.valid);                   // pass (or the game is over): the same side is to move again
    }
    __syncthreads();
    if (lane < depth) t.path[(size_t)g * OZ_MAX_DEPTH + lane] = L.path[lane];            // the frontier, one coalesced store
    if (lane == 0) {
        t.leaf_status[g] = status;
        t.leaf_own[g] = own; t.leaf_opp[g] = opp; t.leaf_legal[g] = legal; t.leaf_fs[g] = fs;
        t.depth[g] = depth; t.term_value[g] = tval;
        unsigned long long* st = t.stat + (size_t)g * OZ_NSTAT;
        st[ST_SIMS] += 1; st[ST_VISITS] += (unsigned long long)(depth + 1);
        if (status == OZ_LEAF_TERMINAL) st[ST_TERMINAL] += 1;
        if (err) atomicOr(t.error_flag, err);
    }
    return status;
}
__global__ __launch_bounds__(64) void k_select(MctsDev t) {
    __shared__ TreeLds L;
    select_body(t, L, blockIdx.x, threadIdx.x);
}

// ---------------------------------------------------------------- K13: leaf compaction
// ballot + prefix sum over the games; slot order = game order (deterministic).
// Leaves are also de-duplicated ACROSS games (G <= OZ_DEDUP_MAX_G): a board that several games reach in the same step
// is evaluated once and every one of those games reads the same (pi, v) row.  The network's output for a position does
// not depend on its slot or on the batch (oz_net.hip), so results are bit-identical with or without this; it removes the
// evaluations of the opening plies, where thousands of concurrent games still walk the same few positions.
// The table (LDS, open addressing) holds the LOWEST game index with a given board: slots are the first occurrences in game order.
// ---- the network's persistent evaluation cache (EvalCacheDev, oz_internal.h)
__device__ __forceinline__ unsigned ec_bucket(const EvalCacheDev& c, uint64_t own, uint64_t opp, unsigned* way_pick) {
    const uint64_t h = key_hash64(own ^ 0x6A09E667F3BCC909ULL, opp);
    *way_pick = (unsigned)(h >> 40) & (OZ_EC_WAYS - 1);
    return (unsigned)(h >> 8) & (c.buckets - 1);
}
// entry index of (own, opp), or -1 (one 64-byte line read)
__device__ __forceinline__ int ec_find(const EvalCacheDev& c, uint64_t own, uint64_t opp) {
    unsigned pick;
    const unsigned b = ec_bucket(c, own, opp, &pick);
    const ulonglong2* k = reinterpret_cast<const ulonglong2*>(c.keys) + (size_t)b * OZ_EC_WAYS;
#pragma unroll
    for (int w = 0; w < OZ_EC_WAYS; ++w) {
        const ulonglong2 e = k[w];
        if (e.x == own && e.y == opp) return (int)(b * OZ_EC_WAYS + w);
    }
    return -1;
}
// the (pi, v) rows of this batch's cache hits: hit h -> row G-1-h of the engine's pi / v arrays.  One thread per (hit, policy entry).
__global__ __launch_bounds__(256) void k_cache_copy(MctsDev t) {
    const int idx = blockIdx.x * 256 + threadIdx.x, h = idx >> 6, a = idx & 63;
    if (h >= *t.hit_count) return;
    const int e = t.hit_entry[h], row = t.G - 1 - h;
    if (a < t.n2) t.pi[(size_t)row * t.n2 + a] = t.ec.pi[(size_t)e * t.n2 + a];
    if (a == 0) t.v[row] = t.ec.v[e];
}
// after the network: every evaluated position (rows 0 .. batch_count-1) goes into the cache.  One wave per position.
__global__ __launch_bounds__(256) void k_cache_insert(MctsDev t) {
    const int lane = threadIdx.x & 63, slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= *t.batch_count) return;
    const uint64_t own = t.batch_own[slot], opp = t.batch_opp[slot];
    unsigned pick;
    const unsigned b = ec_bucket(t.ec, own, opp, &pick);
    const ulonglong2* k = reinterpret_cast<const ulonglong2*>(t.ec.keys) + (size_t)b * OZ_EC_WAYS;
    ulonglong2 e = make_ulonglong2(1, 1);
    if (lane < OZ_EC_WAYS) e = k[lane];
    const uint64_t present = __ballot(lane < OZ_EC_WAYS && e.x == own && e.y == opp);
    if (present) return;                                             // (two games evaluated the same board in one batch and the other one won)
    const uint64_t empty = __ballot(lane < OZ_EC_WAYS && e.x == 0 && e.y == 0);
    // candidate ways: the empty ones in order, else the way the hash picks; an entry takes ONE insert per batch (another position of this
    // batch may be claiming the same way right now: the stamp arbitrates, the loser tries the next candidate or stays uncached)
    uint64_t cand = empty ? empty : (1ULL << pick);
    size_t entry = 0;
    int mine = 0;
    while (cand && !mine) {
        entry = (size_t)b * OZ_EC_WAYS + (unsigned)oz_ctz(cand);
        cand &= cand - 1;
        if (lane == 0) mine = atomicExch(&t.ec.stamp[entry], t.batch_stamp) != t.batch_stamp;
        mine = __shfl(mine, 0, 64);
    }
    if (!mine) return;
    // the key goes LAST, behind a fence: a lookup of another stream that sees the key sees the whole (pi, v) of THIS insert.  (That covers an
    // entry being filled; an entry being REPLACED while another engine's k_cache_copy reads it is not covered -- engines that share a cached
    // network must not run concurrently, see oz_net_set_eval_cache in the header.)
    if (lane == 0) { t.ec.keys[entry * 2] = 0; t.ec.keys[entry * 2 + 1] = 0; }
    __threadfence();
    if (lane < t.n2) t.ec.pi[entry * t.n2 + lane] = t.pi[(size_t)slot * t.n2 + lane];
    if (lane == 0) t.ec.v[entry] = t.v[slot];
    __threadfence();
    if (lane == 0) {
        t.ec.keys[entry * 2] = own; t.ec.keys[entry * 2 + 1] = opp;
        atomicAdd(&t.ec.counters[2], 1ULL);
    }
}

#define OZ_DEDUP_MAX_G 8192
#define OZ_DEDUP_SLOTS 16384
__device__ __forceinline__ unsigned dedup_hash(uint64_t own, uint64_t opp) {
    return (unsigned)(((own * 0x9E3779B97F4A7C15ull) ^ (opp * 0xC2B2AE3D27D4EB4Full)) >> 40) & (OZ_DEDUP_SLOTS - 1);
}
__global__ __launch_bounds__(1024) void k_compact(MctsDev t) {
    __shared__ int wtot[16], whit[16];
    __shared__ int base_s, hit_s;
    __shared__ int tab[OZ_DEDUP_SLOTS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool dedup = t.dedup && t.G <= OZ_DEDUP_MAX_G;
    const bool cached = t.ec.buckets != 0;
    if (tid == 0) { base_s = 0; hit_s = 0; }
    if (dedup) {
        for (int i = tid; i < OZ_DEDUP_SLOTS; i += 1024) tab[i] = -1;
        __syncthreads();
        for (int g = tid; g < t.G; g += 1024) {
            if (t.leaf_status[g] != OZ_LEAF_EVAL) continue;
            const uint64_t own = t.leaf_own[g], opp = t.leaf_opp[g];
            unsigned h = dedup_hash(own, opp);
            for (;;) {
                int cur = atomicCAS(&tab[h], -1, g);
                if (cur == -1) break;                                      // claimed an empty slot
                if (t.leaf_own[cur] == own && t.leaf_opp[cur] == opp) {    // same board: keep the lowest game index
                    atomicMin(&tab[h], g);
                    break;
                }
                h = (h + 1) & (OZ_DEDUP_SLOTS - 1);
            }
        }
    }
    __syncthreads();
    const int cap = t.batch_cap > 0 ? t.batch_cap : t.G;
    for (int start = 0; start < t.G; start += 1024) {
        // game order, starting at game batch_rot (the start rotates from batch to batch when a cap defers leaves: every game gets served)
        const int seq = start + tid;
        int g = seq + t.batch_rot;
        if (g >= t.G) g -= t.G;
        const bool leaf = seq < t.G && t.leaf_status[g] == OZ_LEAF_EVAL;
        int first = g;
        if (leaf && dedup) {
            const uint64_t own = t.leaf_own[g], opp = t.leaf_opp[g];
            unsigned h = dedup_hash(own, opp);
            for (;;) {
                first = tab[h];
                if (t.leaf_own[first] == own && t.leaf_opp[first] == opp) break;
                h = (h + 1) & (OZ_DEDUP_SLOTS - 1);
            }
        }
        // first occurrence: gets a row -- a network slot, or (the board is in the network's evaluation cache) one of the hit rows G-1, G-2, ...
        int entry = -1;
        if (cached && leaf && first == g) entry = ec_find(t.ec, t.leaf_own[g], t.leaf_opp[g]);
        const bool hit = entry >= 0;
        const bool flag = leaf && first == g && !hit;
        const uint64_t b = __ballot(flag), bh = __ballot(hit);
        const int pre = oz_popc(b & ((1ULL << lane) - 1ULL)), preh = oz_popc(bh & ((1ULL << lane) - 1ULL));
        if (lane == 0) { wtot[w] = oz_popc(b); whit[w] = oz_popc(bh); }
        __syncthreads();
        int woff = 0, total = 0, hoff = 0, htotal = 0;
        for (int i = 0; i < 16; ++i) { int x = wtot[i], y = whit[i]; if (i < w) { woff += x; hoff += y; } total += x; htotal += y; }
        const int base = base_s, hbase = hit_s;
        if (hit) {
            const int h = hbase + hoff + preh;
            t.leaf_slot[g] = t.G - 1 - h;
            t.hit_entry[h] = entry;
        }
        if (flag) {
            const int slot = base + woff + pre;
            if (slot < cap) {
                t.leaf_slot[g] = slot;
                t.batch_own[slot] = t.leaf_own[g];
                t.batch_opp[slot] = t.leaf_opp[g];
            } else {
                t.leaf_slot[g] = 0;
                t.leaf_status[g] = OZ_LEAF_WAIT;                            // no slot in this batch: the leaf is offered again in the next one
            }
        } else if (leaf && !hit) {
            t.leaf_slot[g] = -1 - first;                                   // resolved below (the first occurrence may sit in a later chunk)
        }
        __syncthreads();
        if (tid == 0) { base_s = base + total; hit_s = hbase + htotal; }
        __syncthreads();
    }
    if (dedup) {
        __threadfence_block();
        __syncthreads();
        for (int g = tid; g < t.G; g += 1024)
            if (t.leaf_status[g] == OZ_LEAF_EVAL && t.leaf_slot[g] < 0) {
                const int first = -1 - t.leaf_slot[g];
                if (t.leaf_status[first] == OZ_LEAF_WAIT) { t.leaf_slot[g] = 0; t.leaf_status[g] = OZ_LEAF_WAIT; }     // shares the deferred board: waits with it
                else t.leaf_slot[g] = t.leaf_slot[first];
            }
    }
    if (tid == 0) {
        const int used = base_s < cap ? base_s : cap;
        *t.batch_count = used; *t.eval_leaves += (unsigned long long)used;
        *t.hit_count = hit_s;
        if (cached) { atomicAdd(&t.ec.counters[0], (unsigned long long)(base_s + hit_s)); atomicAdd(&t.ec.counters[1], (unsigned long long)hit_s); }
    }
}

// ---------------------------------------------------------------- K5 + K6: expand and backup
__device__ __forceinline__ void q_update(const MctsDev& t, OzEdge* e, double val, int vt) {
    const uint32_t nt = e->n_tag;
    const int N = (int)(nt & ~OZ_TAG_F32);
    const bool q32 = (nt & OZ_TAG_F32) != 0;
    const double Q = e->q;
    uint32_t tag = 0;
    double q;
    if (t.qmode == OZ_QMODE_F64 || (!q32 && vt != VT_F32)) {
        q = ((double)N * Q + val) / (double)(N + 1);                 // MCTS/__init__.py:68 in float64
    } else {                                                          // NumPy >= 2 weak-scalar promotion
        const float prod = q32 ? (float)N * (float)Q : (float)((double)N * Q);
        const float s = prod + (float)val;
        q = (double)(s / (float)(N + 1));
        tag = OZ_TAG_F32;
    }
    e->q = q;
    e->n_tag = (uint32_t)(N + 1) | tag;
}

// backup (MCTS/__init__.py:68-71): the level-d caller sees the leaf value negated (depth-1-d) times; all levels in
// parallel, one lane per level (a path never visits a node twice: every move adds a disc)
__device__ __forceinline__ void backup_body(const MctsDev& t, int g, int lane, double value, int vt) {
    const int depth = t.depth[g];
    if (lane < depth) {
        const int2 pe = t.path[(size_t)g * OZ_MAX_DEPTH + lane];
        const double val = ((depth - 1 - lane) & 1) ? -value : value;
        if ((unsigned)pe.x < (unsigned)t.node_cap && (unsigned)pe.y < (unsigned)t.row_cap) {
            q_update(t, rec_edge(t, g, pe.x, pe.y), val, vt);
            *rec_Ns(t, g, pe.x) += 1;
        } else atomicOr(t.error_flag, EF_CORRUPT);
    }
    if (lane == 0) {
        t.last_value[g] = (depth & 1) ? -value : value;
        t.last_vtype[g] = vt;
    }
}

// slot_is_game != 0: pi / v are indexed by game (host evaluator path); else by compacted slot.
__device__ __forceinline__ void expand_backup_body(const MctsDev& t, TreeLds& L, int g, int lane, int slot_is_game) {
    const int status = unii(t.leaf_status[g]);
    if (status == OZ_LEAF_IDLE || status == OZ_LEAF_WAIT) return;           // (WAIT: the leaf was not in the batch, nothing to expand yet)
    double value;
    int vt;
    if (status == OZ_LEAF_EVAL) {
        // first visit (MCTS/__init__.py:44-57): P = pi * mask, normalised; uniform over legal if the sum is 0
        const uint64_t own = uni64(t.leaf_own[g]), opp = uni64(t.leaf_opp[g]), legal = uni64(t.leaf_legal[g]);
        const int slot = slot_is_game ? g : unii(t.leaf_slot[g]);
        const int r = lane >> 3, c = lane & 7, n = t.n;
        const bool inb = r < n && c < n, is_legal = (legal >> lane) & 1;
        const int a = r * n + c;
        double p = is_legal ? (double)t.pi[(size_t)slot * t.n2 + a] : 0.0;     // float32 * float64 mask
        if (inb) L.arr[a] = p;
        __syncthreads();
        const double sum = pairwise_sum(L.arr, t.n2);
        const int cnt = oz_popc(legal);
        if (sum > 0) p = p / sum;
        else p = is_legal ? 1.0 / (double)cnt : 0.0;                           // mask / np.sum(mask)
        const int node = unii(t.node_count[g]), fs = unii(t.leaf_fs[g]), depth = unii(t.depth[g]);
        const bool ok = node < t.node_cap && cnt <= t.row_cap && fs >= 0;
        if (ok) {
            // build the record in LDS (header + one edge per legal square, N = 0, Q = 0, child unknown), store it with one
            // wave-wide 16-byte store, link it into the table and into the edge we came through
            uint64_t* w = reinterpret_cast<uint64_t*>(L.rec);
            if (lane == 0) { w[0] = own; w[1] = opp; w[2] = (uint64_t)(uint32_t)cnt << 32; w[3] = 0; }
            if (is_legal) {
                const int rank = oz_popc(legal & ((1ULL << lane) - 1ULL));
                w[4 + 3 * rank] = 0xFFFFFFFF00000000ULL;                       // N|tag = 0, child = -1
                w[5 + 3 * rank] = 0;                                           // Q = 0.0
                w[6 + 3 * rank] = (uint64_t)__double_as_longlong(p);
            }
            if (lane == 63 && (cnt * 3) % 2) w[4 + 3 * cnt] = 0;               // the last 16-byte chunk is half used: no stale bytes
            __syncthreads();
            const int chunks = (OZ_REC_HDR + cnt * OZ_EDGE_BYTES + 15) >> 4;
            if (lane < chunks) reinterpret_cast<uint4*>(rec_ptr(t, g, node))[lane] = L.rec[lane];
            if (lane == 0) {
                t.node_count[g] = node + 1;
                t.ht[(size_t)g * t.ht_cap + fs] = ht_entry(own, opp, node);
                if (depth > 0) {
                    const int2 pe = t.path[(size_t)g * OZ_MAX_DEPTH + depth - 1];
                    rec_edge(t, g, pe.x, pe.y)->child = node;
                } else t.root_node[g] = node;
            }
        }
        if (lane == 0) {
            unsigned long long* st = t.stat + (size_t)g * OZ_NSTAT;
            st[ST_EXPAND] += 1;
            if (!(sum > 0)) st[ST_FALLBACK] += 1;
            if (!ok) atomicOr(t.error_flag, node >= t.node_cap || fs < 0 ? EF_NODES : EF_EDGES);
        }
        value = -(double)t.v[slot];                                            // return -v (:57)
        vt = t.qmode == OZ_QMODE_F64 ? VT_F64 : VT_F32;
    } else {
        value = (double)t.term_value[g];
        vt = VT_INT;
    }
    backup_body(t, g, lane, value, vt);
}
__global__ __launch_bounds__(64) void k_expand_backup(MctsDev t, int slot_is_game) {
    __shared__ TreeLds L;
    expand_backup_body(t, L, blockIdx.x, threadIdx.x, slot_is_game);
}
// expand + backup of simulation s-1 and the descent of simulation s in ONE launch (same wave, same game): one kernel boundary
// less per simulation step, and the records the backup has just written are re-read by the descent while still in the CU's cache
__global__ __launch_bounds__(64) void k_backup_select(MctsDev t) {
    __shared__ TreeLds L;
    expand_backup_body(t, L, blockIdx.x, threadIdx.x, 0);
    wave_sync();                                           // the wave's own stores (records, child indices, table entry, root index) before its loads
    __syncthreads();
    select_body(t, L, blockIdx.x, threadIdx.x);
}

// ---------------------------------------------------------------- host object
struct oz_mcts {
    MctsDev d;
    hipStream_t stream = nullptr;
    int device = 0;
    std::mutex mu;
    bool selected = false;
    std::vector<void*> allocs;
    // HIP-event timing on the launch stream: slot TS_NN (the evaluator's launches) whenever a driver asks for it,
    // the tree kernels too when `profile` is on (oz_selfplay_profile)
    OzTimer timer{OZ_TREE_KERNELS};
    bool profile = false;
    unsigned batch_no = 0;           // batches evaluated (stamp of the evaluation-cache inserts)
    // staging of oz_mcts_root_counts (called once per move by the drop-in agents): lives with the object
    int32_t* rc_counts = nullptr; uint64_t* rc_legal = nullptr; int32_t* rc_rc = nullptr;

    template <typename T> int alloc(T** p, size_t count) {
        OZ_HIP(hipMalloc((void**)p, sizeof(T) * (count ? count : 1)));
        allocs.push_back(*p);
        return OZ_OK;
    }
};

static int next_pow2(int x) { int p = 64; while (p < x) p <<= 1; return p; }

static int mcts_reset_locked(oz_mcts* m, int game) {
    MctsDev& d = m->d;
    if (game < 0) {
        OZ_HIP(hipMemsetAsync(d.node_count, 0, sizeof(int) * d.G, m->stream));
        OZ_HIP(hipMemsetAsync(d.root_node, 0xFF, sizeof(int) * d.G, m->stream));
        OZ_HIP(hipMemsetAsync(d.ht, 0, sizeof(uint32_t) * (size_t)d.G * d.ht_cap, m->stream));
    } else {
        OZ_REQUIRE(game < d.G, "game index %d out of range", game);
        OZ_HIP(hipMemsetAsync(d.node_count + game, 0, sizeof(int), m->stream));
        OZ_HIP(hipMemsetAsync(d.root_node + game, 0xFF, sizeof(int), m->stream));
        OZ_HIP(hipMemsetAsync(d.ht + (size_t)game * d.ht_cap, 0, sizeof(uint32_t) * (size_t)d.ht_cap, m->stream));
    }
    return OZ_OK;
}

// row capacity of a node record: the largest legal-move count the board can show (n*n - 4 empty squares at most), capped
// so that a record fits one wave-wide 16-byte load (1024 B = header + 41 edges; 33 is the known maximum mobility of 8x8
// Othello positions).  A position with more legal moves raises OZ_ERR_CAPACITY, never a silent truncation.
static int row_cap_for(int n) { const int e = n * n - 4; return e < 41 ? e : 41; }

static int mcts_create(oz_mcts** out, int n, int G, int node_cap, double c, int q_mode) {
    OZ_REQUIRE(n == 4 || n == 6 || n == 8, "board size must be 4, 6 or 8 (got %d)", n);
    OZ_REQUIRE(G > 0 && node_cap > 0, "num_games and node_cap must be positive");
    OZ_REQUIRE(node_cap < (int)OZ_HT_IDX_MASK, "node_cap %d too large (max %u)", node_cap, OZ_HT_IDX_MASK - 1);
    OZ_REQUIRE(q_mode == OZ_QMODE_NEP50 || q_mode == OZ_QMODE_F64, "unknown q_mode %d", q_mode);
    oz_mcts* m = new oz_mcts();
    m->device = oz_current_device();
    MctsDev& d = m->d;
    memset(&d, 0, sizeof d);
    d.G = G; d.n = n; d.n2 = n * n; d.node_cap = node_cap;
    d.row_cap = row_cap_for(n);
    d.rec_bytes = (OZ_REC_HDR + d.row_cap * OZ_EDGE_BYTES + 63) & ~63;
    d.ht_cap = next_pow2(2 * node_cap);
    d.valid = oz_valid_mask(n); d.c = c; d.qmode = q_mode;
    int rc = OZ_OK;
#define A(ptr, cnt) if (!rc) rc = m->alloc(&d.ptr, cnt)
    A(recs, (size_t)G * node_cap * d.rec_bytes);
    A(node_count, G); A(root_node, G);
    A(ht, (size_t)G * d.ht_cap);
    A(root_own, G); A(root_opp, G); A(active, G);
    A(leaf_status, G); A(leaf_own, G); A(leaf_opp, G); A(leaf_legal, G);
    A(depth, G); A(term_value, G); A(leaf_slot, G); A(leaf_fs, G); A(path, (size_t)G * OZ_MAX_DEPTH);
    A(batch_own, G); A(batch_opp, G); A(batch_count, 1);
    A(pi, (size_t)G * d.n2); A(v, G);
    A(last_value, G); A(last_vtype, G);
    A(stat, (size_t)G * OZ_NSTAT); A(error_flag, 1); A(eval_leaves, 1);
    A(hit_count, 1); A(hit_entry, G);
#undef A
    if (!rc) rc = m->alloc(&m->rc_counts, (size_t)G * 64);
    if (!rc) rc = m->alloc(&m->rc_legal, (size_t)G);
    if (!rc) rc = m->alloc(&m->rc_rc, (size_t)G);
    d.batch_cap = 0; d.batch_rot = 0;
    d.dedup = 1;               // cross-game leaf de-duplication (oz_mcts_set_dedup / oz_selfplay_config.dedup switch it off)
    if (!rc && hipStreamCreate(&m->stream) != hipSuccess) { oz_set_error("hipStreamCreate failed"); rc = OZ_ERR_HIP; }
    if (!rc) {
        hipMemsetAsync(d.stat, 0, sizeof(unsigned long long) * (size_t)G * OZ_NSTAT, m->stream);
        hipMemsetAsync(d.error_flag, 0, sizeof(int), m->stream);
        hipMemsetAsync(d.eval_leaves, 0, sizeof(unsigned long long), m->stream);
        hipMemsetAsync(d.hit_count, 0, sizeof(int), m->stream);
        hipMemsetAsync(d.active, 0, G, m->stream);
        hipMemsetAsync(d.leaf_status, 0, sizeof(int) * G, m->stream);
        hipMemsetAsync(d.batch_count, 0, sizeof(int), m->stream);
        hipMemsetAsync(d.last_value, 0, sizeof(double) * G, m->stream);
        hipMemsetAsync(d.last_vtype, 0, sizeof(int) * G, m->stream);
        hipMemsetAsync(d.depth, 0, sizeof(int) * G, m->stream);
        rc = mcts_reset_locked(m, -1);
        if (!rc && hipStreamSynchronize(m->stream) != hipSuccess) { oz_set_error("stream sync failed"); rc = OZ_ERR_HIP; }
    }
    if (rc) {
        for (void* p : m->allocs) hipFree(p);
        if (m->stream) hipStreamDestroy(m->stream);
        delete m;
        return rc;
    }
    *out = m;
    return OZ_OK;
}

static void mcts_destroy(oz_mcts* m) {
    if (!m) return;
    hipSetDevice(m->device);
    hipStreamSynchronize(m->stream);
    m->timer.destroy();
    for (void* p : m->allocs) hipFree(p);
    hipStreamDestroy(m->stream);
    delete m;
}

static int check_error_flag(oz_mcts* m) {
    int ef = 0;
    OZ_HIP(hipMemcpyAsync(&ef, m->d.error_flag, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    OZ_HIP(hipStreamSynchronize(m->stream));
    if (ef & (EF_NODES | EF_EDGES)) {
        if (ef & EF_NODES) oz_set_error("per-game node table overflow (node_cap=%d): raise the capacity", m->d.node_cap);
        else oz_set_error("a position has more than %d legal moves (the row capacity of a node record)", m->d.row_cap);
        return OZ_ERR_CAPACITY;
    }
    if (ef & EF_RECORDS) { oz_set_error("move-record buffer overflow: raise record_cap or drain records more often"); return OZ_ERR_CAPACITY; }
    if (ef & EF_DEPTH) { oz_set_error("search path deeper than %d", OZ_MAX_DEPTH); return OZ_ERR_CAPACITY; }
    if (ef & EF_NOMOVE) { oz_set_error("a revisited state has no legal action (the reference raises ValueError: max() arg is an empty sequence)"); return OZ_ERR_STATE; }
    if (ef & EF_ROOT) { oz_set_error("root state missing from the search table after the simulations"); return OZ_ERR_KEY; }
    if (ef & EF_CORRUPT) { oz_set_error("search table corrupt: a node index outside the pool was met (internal error)"); return OZ_ERR_STATE; }
    return OZ_OK;
}

enum { TS_SELECT = 0, TS_COMPACT = 1, TS_NN = 2, TS_BACKUP = 3, TS_MOVE = 4 };
// the evaluator's part of a batch: the rows of the leaves k_compact found in the network's evaluation cache (if the search uses one),
// the network on the compacted batch (launched for up to max_count leaves), the evaluated positions into the cache
static int eval_batch_async(oz_mcts* m, oz_net* net, int max_count, bool timed) {
    MctsDev& d = m->d;
    hipStream_t s = m->stream;
    if (d.ec.buckets) hipLaunchKernelGGL(k_cache_copy, dim3((unsigned)(((long long)d.G * 64 + 255) / 256)), dim3(256), 0, s, d);
    const long long ti = timed ? m->timer.begin(TS_NN, s) : -1;
    if (int rc = oz_net_forward_device(net, d.batch_own, d.batch_opp, d.batch_count, max_count, d.pi, d.v, s)) { m->timer.cancel(ti); return rc; }
    m->timer.end(ti, s);
    if (d.ec.buckets) {
        m->batch_no += 1;
        if (m->batch_no == 0) m->batch_no = 1;               // 0 = "never written"
        d.batch_stamp = m->batch_no;
        hipLaunchKernelGGL(k_cache_insert, dim3((unsigned)((max_count + 3) / 4)), dim3(256), 0, s, d);
    }
    return OZ_OK;
}

// one lock-step simulation for every active game, leaves evaluated by `net` (all on m->stream)
static int mcts_step_async(oz_mcts* m, oz_net* net, bool time_eval) {
    MctsDev& d = m->d;
    hipStream_t s = m->stream;
    const bool all = m->profile;
    long long i = all ? m->timer.begin(TS_SELECT, s) : -1;
    hipLaunchKernelGGL(k_select, dim3(d.G), dim3(64), 0, s, d);
    m->timer.end(i, s);
    i = all ? m->timer.begin(TS_COMPACT, s) : -1;
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, s, d);
    m->timer.end(i, s);
    if (int rc = eval_batch_async(m, net, d.G, time_eval || all)) return rc;
    i = all ? m->timer.begin(TS_BACKUP, s) : -1;
    hipLaunchKernelGGL(k_expand_backup, dim3(d.G), dim3(64), 0, s, d, 0);
    m->timer.end(i, s);
    OZ_HIP(hipGetLastError());
    return OZ_OK;
}

static int mcts_collect_eval_time(oz_mcts* m) {
    if (m->timer.collect() != OZ_OK) { oz_set_error("HIP event timing failed"); return OZ_ERR_HIP; }
    return OZ_OK;
}

// `nsims` lock-step simulations: descent | compaction | evaluator, then per further simulation the previous one's expand + backup
// fused with the next descent (k_backup_select), and one closing expand + backup: nsims + 1 tree launches instead of 2 nsims
// (results are identical to the unfused sequence, which a single step still uses).
// mcts_step_k enqueues simulation k of such a sequence, mcts_steps_close the closing expand + backup: the arena interleaves the steps of
// its two searches on two streams.  max_leaves = an upper bound of the leaves a batch can hold (<= G: the active games), the size the
// evaluator's launches are made for.
static int mcts_step_k(oz_mcts* m, oz_net* net, int k, int max_leaves, bool time_eval) {
    MctsDev& d = m->d;
    hipStream_t s = m->stream;
    const bool all = m->profile;
    long long i = all ? m->timer.begin(TS_SELECT, s) : -1;
    if (k == 0) hipLaunchKernelGGL(k_select, dim3(d.G), dim3(64), 0, s, d);
    else hipLaunchKernelGGL(k_backup_select, dim3(d.G), dim3(64), 0, s, d);
    m->timer.end(i, s);
    i = all ? m->timer.begin(TS_COMPACT, s) : -1;
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, s, d);
    m->timer.end(i, s);
    return eval_batch_async(m, net, max_leaves, time_eval || all);
}
static int mcts_steps_close(oz_mcts* m) {
    const bool all = m->profile;
    const long long i = all ? m->timer.begin(TS_BACKUP, m->stream) : -1;
    hipLaunchKernelGGL(k_expand_backup, dim3(m->d.G), dim3(64), 0, m->stream, m->d, 0);
    m->timer.end(i, m->stream);
    OZ_HIP(hipGetLastError());
    return OZ_OK;
}
static int mcts_steps_async(oz_mcts* m, oz_net* net, int nsims, bool time_eval) {
    if (nsims < 2) {
        for (int i = 0; i < nsims; ++i)
            if (int rc = mcts_step_async(m, net, time_eval)) return rc;
        return OZ_OK;
    }
    for (int k = 0; k < nsims; ++k)
        if (int rc = mcts_step_k(m, net, k, m->d.G, time_eval)) return rc;
    return mcts_steps_close(m);
}

OZ_API int oz_mcts_create(oz_mcts** out, int n, int num_games, int node_cap, double c, int q_mode) {
    OZ_REQUIRE(out, "null out pointer");
    return mcts_create(out, n, num_games, node_cap, c, q_mode);
}
OZ_API int oz_mcts_destroy(oz_mcts* m) { mcts_destroy(m); return OZ_OK; }

OZ_API int oz_mcts_set_dedup(oz_mcts* m, int enable) {
    OZ_REQUIRE(m, "null mcts");
    std::lock_guard<std::mutex> lk(m->mu);
    m->d.dedup = enable ? 1 : 0;                          // takes effect at the next leaf compaction (kernels get the struct by value)
    return OZ_OK;
}

OZ_API int oz_mcts_reset(oz_mcts* m, int game) {
    OZ_REQUIRE(m, "null mcts");
    std::lock_guard<std::mutex> lk(m->mu);
    hipSetDevice(m->device);
    if (int rc = mcts_reset_locked(m, game)) return rc;
    OZ_HIP(hipStreamSynchronize(m->stream));
    m->selected = false;
    return OZ_OK;
}

OZ_API int oz_mcts_set_roots(oz_mcts* m, const uint64_t* own, const uint64_t* opp, const uint8_t* active) {
    OZ_REQUIRE(m && own && opp, "null argument");
    std::lock_guard<std::mutex> lk(m->mu);
    hipSetDevice(m->device);
    const int G = m->d.G;
    for (int i = 0; i < G; ++i)
        OZ_REQUIRE((own[i] & opp[i]) == 0 && ((own[i] | opp[i]) & ~m->d.valid) == 0, "root %d is not a valid %dx%d board", i, m->d.n, m->d.n);
    OZ_HIP(hipMemcpyAsync(m->d.root_own, own, 8ull * G, hipMemcpyHostToDevice, m->stream));
    OZ_HIP(hipMemcpyAsync(m->d.root_opp, opp, 8ull * G, hipMemcpyHostToDevice, m->stream));
    if (active) OZ_HIP(hipMemcpyAsync(m->d.active, active, G, hipMemcpyHostToDevice, m->stream));
    else OZ_HIP(hipMemsetAsync(m->d.active, 1, G, m->stream));
    OZ_HIP(hipMemsetAsync(m->d.root_node, 0xFF, sizeof(int) * G, m->stream));       // new roots: look them up again
    OZ_HIP(hipStreamSynchronize(m->stream));
    m->selected = false;
    return OZ_OK;
}

OZ_API int oz_mcts_simulate(oz_mcts* m, oz_net* net, int nsims) {
    OZ_REQUIRE(m && net, "null argument");
    OZ_REQUIRE(net->n == m->d.n, "network board size %d != search board size %d", net->n, m->d.n);
    OZ_REQUIRE(net->max_batch >= m->d.G, "network max_batch %d < num_games %d", net->max_batch, m->d.G);
    std::lock_guard<std::mutex> lk(m->mu);
    std::lock_guard<std::mutex> lkn(net->mu);
    hipSetDevice(m->device);
    if (int rc = mcts_steps_async(m, net, nsims, false)) return rc;
    m->selected = false;
    return check_error_flag(m);
}

OZ_API int oz_mcts_select(oz_mcts* m) {
    OZ_REQUIRE(m, "null mcts");
    std::lock_guard<std::mutex> lk(m->mu);
    hipSetDevice(m->device);
    hipLaunchKernelGGL(k_select, dim3(m->d.G), dim3(64), 0, m->stream, m->d);
    OZ_HIP(hipGetLastError());
    m->selected = true;
    return check_error_flag(m);
}

OZ_API int oz_mcts_leaves(oz_mcts* m, int32_t* status, uint64_t* own, uint64_t* opp) {
    OZ_REQUIRE(m && status && own && opp, "null argument");
    std::lock_guard<std::mutex> lk(m->mu);
    OZ_REQUIRE(m->selected, "oz_mcts_leaves: call oz_mcts_select first");
    hipSetDevice(m->device);
    const int G = m->d.G;
    OZ_HIP(hipMemcpyAsync(status, m->d.leaf_status, 4ull * G, hipMemcpyDeviceToHost, m->stream));
    OZ_HIP(hipMemcpyAsync(own, m->d.leaf_own, 8ull * G, hipMemcpyDeviceToHost, m->stream));
    OZ_HIP(hipMemcpyAsync(opp, m->d.leaf_opp, 8ull * G, hipMemcpyDeviceToHost, m->stream));
    OZ_HIP(hipStreamSynchronize(m->stream));
    return OZ_OK;
}

OZ_API int oz_mcts_backup(oz_mcts* m, const float* pi, const float* v) {
    OZ_REQUIRE(m && pi && v, "null argument");
    std::lock_guard<std::mutex> lk(m->mu);
    if (!m->selected) { oz_set_error("oz_mcts_backup: call oz_mcts_select first"); return OZ_ERR_STATE; }
    hipSetDevice(m->device);
    const int G = m->d.G;
    OZ_HIP(hipMemcpyAsync(m->d.pi, pi, 4ull * G * m->d.n2, hipMemcpyHostToDevice, m->stream));
    OZ_HIP(hipMemcpyAsync(m->d.v, v, 4ull * G, hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(k_expand_backup, dim3(G), dim3(64), 0, m->stream, m->d, 1);
    OZ_HIP(hipGetLastError());
    m->selected = false;
    return check_error_flag(m);
}

OZ_API int oz_mcts_last_value(oz_mcts* m, double* value, int32_t* vtype, int32_t* depth) {
    OZ_REQUIRE(m, "null mcts");
    std::lock_guard<std::mutex> lk(m->mu);
    hipSetDevice(m->device);
    const int G = m->d.G;
    if (value) OZ_HIP(hipMemcpyAsync(value, m->d.last_value, 8ull * G, hipMemcpyDeviceToHost, m->stream));
    if (vtype) OZ_HIP(hipMemcpyAsync(vtype, m->d.last_vtype, 4ull * G, hipMemcpyDeviceToHost, m->stream));
    if (depth) OZ_HIP(hipMemcpyAsync(depth, m->d.depth, 4ull * G, hipMemcpyDeviceToHost, m->stream));
    OZ_HIP(hipStreamSynchronize(m->stream));
    return OZ_OK;
}

// N(state, action) for the root of every slot (MCTS/__init__.py:73-84,172-175)
// record index of the root of slot g (cached by the descent, else looked up); -1 = not in the table
__device__ __forceinline__ int root_lookup(const MctsDev& t, int g, uint64_t own, uint64_t opp, int lane) {
    int node = unii(t.root_node[g]);
    if (node < 0) { int fs; node = ht_find(t, g, own, opp, lane, &fs); }
    return node;
}
__global__ __launch_bounds__(64) void k_root_counts(MctsDev t, int32_t* counts, uint64_t* legal_out, int32_t* rc_out) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const uint64_t own = uni64(t.root_own[g]), opp = uni64(t.root_opp[g]);
    const uint64_t legal = oz_legal(own, opp, t.valid);
    const int node = root_lookup(t, g, own, opp, lane);
    int cnt = 0, rc = 0;
    if (node < 0) rc = 1;
    else if (*rec_Ns(t, g, node) == 0) rc = 2;          // _Nsa[hash] still empty -> KeyError in the reference
    else if ((legal >> lane) & 1)
        cnt = (int)(rec_edge(t, g, node, oz_popc(legal & ((1ULL << lane) - 1ULL)))->n_tag & ~OZ_TAG_F32);
    counts[(size_t)g * 64 + lane] = cnt;
    if (lane == 0) { legal_out[g] = legal; rc_out[g] = rc; }
}

OZ_API int oz_mcts_root_counts(oz_mcts* m, int32_t* counts, uint64_t* legal, int32_t* rc) {
    OZ_REQUIRE(m && counts && legal && rc, "null argument");
    std::lock_guard<std::mutex> lk(m->mu);
    hipSetDevice(m->device);
    const int G = m->d.G;
    int32_t* dc = m->rc_counts; uint64_t* dl = m->rc_legal; int32_t* dr = m->rc_rc;
    hipLaunchKernelGGL(k_root_counts, dim3(G), dim3(64), 0, m->stream, m->d, dc, dl, dr);
    OZ_HIP(hipMemcpyAsync(counts, dc, 4ull * G * 64, hipMemcpyDeviceToHost, m->stream));
    OZ_HIP(hipMemcpyAsync(legal, dl, 8ull * G, hipMemcpyDeviceToHost, m->stream));
    OZ_HIP(hipMemcpyAsync(rc, dr, 4ull * G, hipMemcpyDeviceToHost, m->stream));
    OZ_HIP(hipStreamSynchronize(m->stream));
    return OZ_OK;
}

// OthelloMCTS.get_policy_action_probabilities (othelo_mcts.py:51-67) of every game's current root: float64 (n, n) row-major rows.
//   temperature != 0: N ** (1 / T) on the legal squares, divided by np.sum of the (n, n) array (NumPy's pairwise order) -- or by 1 if that is 0;
//   temperature == 0: one-hot of a best square: of the squares whose count equals the maximum of the (n, n) array, in row-major order (np.argwhere),
//                     the one `tie_draws[g] % (how many)` picks -- random.choice(bests) in the reference; null tie_draws: the first.
// rc[g] as oz_mcts_root_counts reports it; a root that was expanded but never selected from (rc 2: KeyError in the reference) gets a zero row.
OZ_API int oz_mcts_policy(oz_mcts* m, double temperature, const uint64_t* tie_draws, double* policy, int32_t* rc) {
    OZ_REQUIRE(m && policy && rc, "null argument");
    const int G = m->d.G, n = m->d.n, n2 = n * n;
    std::vector<int32_t> counts((size_t)G * 64);
    std::vector<uint64_t> legal((size_t)G);
    if (int r = oz_mcts_root_counts(m, counts.data(), legal.data(), rc)) return r;
    for (int g = 0; g < G; ++g) {
        double* out = policy + (size_t)g * n2;
        double arr[64];
        for (int i = 0; i < n2; ++i) arr[i] = 0.0;
        for (int i = 0; i < n2; ++i) out[i] = 0.0;
        if (rc[g] == 2) continue;
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c)
                if ((legal[g] >> (r * 8 + c)) & 1) {
                    const int cnt = counts[(size_t)g * 64 + r * 8 + c];
                    arr[r * n + c] = temperature == 0 ? (double)cnt : pow((double)cnt, 1.0 / temperature);
                }
        if (temperature == 0) {
            double mx = arr[0];
            for (int i = 1; i < n2; ++i) mx = arr[i] > mx ? arr[i] : mx;
            int nb = 0;
            for (int i = 0; i < n2; ++i) nb += arr[i] == mx;
            int pick = tie_draws ? (int)(tie_draws[g] % (uint64_t)nb) : 0;
            for (int i = 0; i < n2; ++i)
                if (arr[i] == mx && pick-- == 0) { out[i] = 1.0; break; }
        } else {
            double sum = n2 >= 8 ? pairwise_sum(arr, n2) : 0.0;
            if (n2 < 8) for (int i = 0; i < n2; ++i) sum += arr[i];
            if (sum == 0) sum = 1.0;
            for (int i = 0; i < n2; ++i) out[i] = arr[i] / sum;
        }
    }
    return OZ_OK;
}

OZ_API int oz_mcts_num_nodes(oz_mcts* m, int32_t* num_nodes) {
    OZ_REQUIRE(m && num_nodes, "null argument");
    std::lock_guard<std::mutex> lk(m->mu);
    hipSetDevice(m->device);
    OZ_HIP(hipMemcpyAsync(num_nodes, m->d.node_count, 4ull * m->d.G, hipMemcpyDeviceToHost, m->stream));
    OZ_HIP(hipStreamSynchronize(m->stream));
    return OZ_OK;
}

OZ_API int oz_mcts_dump_node(oz_mcts* m, int game, int index, uint64_t* own, uint64_t* opp, int32_t* Ns, uint64_t* legal,
                             int32_t* N, double* Q, uint8_t* qtag, double* P) {
    OZ_REQUIRE(m, "null mcts");
    std::lock_guard<std::mutex> lk(m->mu);
    hipSetDevice(m->device);
    MctsDev& d = m->d;
    OZ_REQUIRE(game >= 0 && game < d.G, "game index out of range");
    OZ_HIP(hipStreamSynchronize(m->stream));
    int nn = 0;
    OZ_HIP(hipMemcpy(&nn, d.node_count + game, 4, hipMemcpyDeviceToHost));
    OZ_REQUIRE(index >= 0 && index < nn, "node index %d out of range (%d nodes)", index, nn);
    std::vector<unsigned char> rec((size_t)d.rec_bytes);
    OZ_HIP(hipMemcpy(rec.data(), d.recs + ((size_t)game * d.node_cap + index) * (size_t)d.rec_bytes, rec.size(), hipMemcpyDeviceToHost));
    const uint64_t* h = reinterpret_cast<const uint64_t*>(rec.data());
    *own = h[0]; *opp = h[1];
    *Ns = (int32_t)(uint32_t)h[2];
    *legal = oz_legal(h[0], h[1], d.valid);                 // the legal set is a function of the key (recomputed, not stored)
    const OzEdge* e = reinterpret_cast<const OzEdge*>(rec.data() + OZ_REC_HDR);
    for (int s = 0; s < 64; ++s) { N[s] = 0; Q[s] = 0; qtag[s] = 0; P[s] = 0; }
    int k = 0;
    for (int s = 0; s < 64; ++s)
        if ((*legal >> s) & 1) { N[s] = (int)(e[k].n_tag & ~OZ_TAG_F32); qtag[s] = (e[k].n_tag & OZ_TAG_F32) ? 1 : 0; Q[s] = e[k].q; P[s] = e[k].p; ++k; }
    return OZ_OK;
}

static int mcts_stats_locked(oz_mcts* m, int64_t* out5) {
    const size_t cnt = (size_t)m->d.G * OZ_NSTAT;
    std::vector<unsigned long long> h(cnt);
    OZ_HIP(hipMemcpyAsync(h.data(), m->d.stat, 8 * cnt, hipMemcpyDeviceToHost, m->stream));
    OZ_HIP(hipStreamSynchronize(m->stream));
    for (int k = 0; k < OZ_NSTAT; ++k) out5[k] = 0;
    for (size_t i = 0; i < cnt; ++i) out5[i % OZ_NSTAT] += (int64_t)h[i];
    return OZ_OK;
}
OZ_API int oz_mcts_stats(oz_mcts* m, int64_t* out5) {
    OZ_REQUIRE(m && out5, "null argument");
    std::lock_guard<std::mutex> lk(m->mu);
    hipSetDevice(m->device);
    return mcts_stats_locked(m, out5);
}

// ================================================================ self-play driver
struct GamesDev {
    int G, n;
    uint64_t valid;
    uint64_t *black, *white, *game_id;
    int8_t* player;
    uint8_t* finished;
    int* ply;
    // per-slot move log of the game in progress
    uint64_t *log_black, *log_white;      // [G][64]
    uint8_t *log_action, *log_greedy;     // [G][64]
    int8_t* log_player;                   // [G][64]
    int32_t* last_counts;                 // [G][64]
    oz_record* records;
    unsigned long long* counters;         // [0] records [1] games completed [2] moves
    long long record_cap;
    uint64_t seed, id_stride;
    double temperature, e_greedy;
    int refill, init_black_set;
    uint64_t init_black, init_white;
};

__global__ void k_sp_roots(GamesDev gm, MctsDev t, int mover_filter /* 0 all, +1 / -1: only games with that mover */) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= gm.G) return;
    const int p = gm.player[g];
    const bool act = !gm.finished[g] && (mover_filter == 0 || mover_filter == p);
    t.active[g] = act ? 1 : 0;
    if (act) {
        t.root_node[g] = -1;
        t.root_own[g] = p == 1 ? gm.black[g] : gm.white[g];
        t.root_opp[g] = p == 1 ? gm.white[g] : gm.black[g];
    }
}

// oz_selfplay_stagger: in round r only the slots whose start offset is still ahead of r search and move
__global__ void k_sp_roots_stagger(GamesDev gm, MctsDev t, int round, int period) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= gm.G) return;
    const int offset = (int)(((long long)g * period) / gm.G);
    const int p = gm.player[g];
    const bool act = !gm.finished[g] && offset > round;
    t.active[g] = act ? 1 : 0;
    if (act) {
        t.root_node[g] = -1;
        t.root_own[g] = p == 1 ? gm.black[g] : gm.white[g];
        t.root_opp[g] = p == 1 ? gm.white[g] : gm.black[g];
    }
}

// K7: root policy extraction + action choice + OthelloGame.play + example recording
// (othelo_mcts.py:51-67, training.py:45-67 / agents.py:52-68).  One wave per game.
//   arena != 0: agents.py semantics (temperature 0, argmax over valid actions of the one-hot).
__device__ __forceinline__ void sp_move_body(const GamesDev& gm, const MctsDev& t, int g, int lane, int arena) {
    if (!t.active[g]) return;
    const uint64_t own = uni64(t.root_own[g]), opp = uni64(t.root_opp[g]);
    const uint64_t legal = oz_legal(own, opp, t.valid);
    const int node = root_lookup(t, g, own, opp, lane);
    if (node < 0 || *rec_Ns(t, g, node) == 0) {           // KeyError path of the reference (sims < 2)
        if (lane == 0) atomicOr(t.error_flag, EF_ROOT);
        return;
    }
    const bool is_legal = (legal >> lane) & 1;
    int cnt = 0;
    if (is_legal) cnt = (int)(rec_edge(t, g, node, oz_popc(legal & ((1ULL << lane) - 1ULL)))->n_tag & ~OZ_TAG_F32);
    gm.last_counts[(size_t)g * 64 + lane] = cnt;
    const int ply = unii(gm.ply[g]);
    const uint64_t gid = uni64(gm.game_id[g]);
    const int mx = unii(wave_max_i32(cnt));                      // >= 1 once the root has been selected from
    int action, greedy = 1;
    if (arena || gm.temperature == 0.0) {
        // bests = argwhere(p == p.max()) over the whole board; random.choice(bests) -> RNG_TIE stream
        const uint64_t bests = __ballot(is_legal && cnt == mx);
        const int nbests = oz_popc(bests);
        action = oz_kth_bit(bests, (int)(oz_rng(gm.seed, gid, (uint64_t)ply, OZ_RNG_TIE) % (uint64_t)nbests));
    } else {
        // N**(1/T) / sum is monotone in N: argwhere(policy == policy.max())[0] = first max-visit square
        action = oz_ctz(__ballot(is_legal && cnt == mx));
    }
    if (!arena) {
        const double coin = oz_rng_unit(oz_rng(gm.seed, gid, (uint64_t)ply, OZ_RNG_COIN));
        if (!(coin <= gm.e_greedy)) {                      // training.py:53-56
            greedy = 0;
            action = oz_kth_bit(legal, (int)(oz_rng(gm.seed, gid, (uint64_t)ply, OZ_RNG_EXPLORE) % (uint64_t)oz_popc(legal)));
        }
    }
    uint64_t black = uni64(gm.black[g]), white = uni64(gm.white[g]);
    int player = unii(gm.player[g]), fin = 0;
    const size_t lb = (size_t)g * 64;
    if (lane == 0 && ply < 64) {
        gm.log_black[lb + ply] = black; gm.log_white[lb + ply] = white;
        gm.log_action[lb + ply] = (uint8_t)action; gm.log_player[lb + ply] = (int8_t)player;
        gm.log_greedy[lb + ply] = (uint8_t)greedy;
    }
    oz_game_play(black, white, player, fin, action, gm.valid);
    const int nply = ply + 1;
    if (fin) {
        // get_winning_player: draw -> BLACK; z = +1 if winner == mover else -1 (training.py:69-72)
        const int winner = oz_popc(black) >= oz_popc(white) ? 1 : -1;
        unsigned long long base = 0;
        if (lane == 0) {
            base = atomicAdd(&gm.counters[0], (unsigned long long)nply);
            atomicAdd(&gm.counters[1], 1ULL);
        }
        base = __shfl(base, 0, 64);
        if (lane < nply) {                                 // lane i writes the record of ply i
            if ((long long)(base + lane) < gm.record_cap) {
                oz_record r;
                const bool cur = lane == ply;              // this ply's log entry is still in registers
                r.black = cur ? gm.black[g] : gm.log_black[lb + lane];
                r.white = cur ? gm.white[g] : gm.log_white[lb + lane];
                r.final_black = black; r.final_white = white; r.game_id = gid;
                r.ply = (uint8_t)lane; r.action = cur ? (uint8_t)action : gm.log_action[lb + lane];
                r.player = cur ? gm.player[g] : gm.log_player[lb + lane];
                r.z = (int8_t)(winner == r.player ? 1 : -1);
                r.greedy = cur ? (uint8_t)greedy : gm.log_greedy[lb + lane];
                r.pad[0] = r.pad[1] = r.pad[2] = 0;
                gm.records[base + lane] = r;
            } else atomicOr(t.error_flag, EF_RECORDS);
        }
    }
    if (lane == 0) atomicAdd(&gm.counters[2], 1ULL);
    if (fin && gm.refill) {
        // a fresh OthelloGame + a fresh OthelloMCTS in the same slot (training.py:30-32)
        for (int i = lane; i < t.ht_cap; i += 64) t.ht[(size_t)g * t.ht_cap + i] = 0;
        if (lane == 0) {
            t.node_count[g] = 0; t.root_node[g] = -1;
            gm.black[g] = gm.init_black; gm.white[g] = gm.init_white; gm.player[g] = 1;
            gm.finished[g] = 0; gm.ply[g] = 0; gm.game_id[g] = gid + gm.id_stride;
        }
    } else if (lane == 0) {
        gm.black[g] = black; gm.white[g] = white; gm.player[g] = (int8_t)player;
        gm.finished[g] = (uint8_t)fin; gm.ply[g] = nply;
        t.root_node[g] = -1;                               // the position changed: the next descent looks its root up again
    }
}
__global__ __launch_bounds__(64) void k_sp_move(GamesDev gm, MctsDev t, int arena) { sp_move_body(gm, t, blockIdx.x, threadIdx.x, arena); }

// ---------------------------------------------------------------- free-running self-play step
// The lock-step driver gives every game one simulation per step; simulations that end on a finished board need no
// network evaluation, so ~8 % of the slots of a leaf batch stay empty and a move round always costs `sims` network passes.
// k_advance lets every game run on by itself until its next simulation needs the network: it counts the simulation the
// previous step completed, plays the move once `sims` of them are done (k_sp_move's body: action choice, play, records,
// refill), walks further simulations that end on finished boards (k_select's body + the backup), and stops at the first
// first-visit leaf -- so (almost) every game contributes one leaf to every batch.  A game's own sequence of simulations,
// moves and random draws is exactly the lock-step one (games are independent, streams are keyed by game id and ply, a
// position's (pi, v) does not depend on the batch it sits in), so records are identical; only the interleaving of the
// games changes.  OZ_ADVANCE_CAP bounds the work of one call (late-game positions whose whole remaining tree is known
// can run many network-free simulations); a game that hits the cap simply contributes no leaf to this batch.  The launch lasts
// as long as its slowest wave, so the cap trades batch fill for launch time -- measured at 4096 games x 100 simulations
// (bench.py --driver free): cap 24 -> 4081 leaves per batch but 367 us per launch, 1.40 M expansions/s; cap 8 -> 1.56 M;
// cap 4 -> 1.585 M; cap 2 -> 3908 leaves per batch, 1.59 M (1.60 M with k_backup_advance; the lock-step driver: 1.56-1.58 M).
#define OZ_ADVANCE_CAP 2
__device__ __forceinline__ void advance_body(const GamesDev& gm, const MctsDev& t, TreeLds& L, int g, int lane, int sims, int* __restrict__ sims_done, int cap) {
    if (unii(t.leaf_status[g]) == OZ_LEAF_WAIT) {          // batch cap: the leaf of the simulation in progress found no slot -- offer it again, unchanged
        if (lane == 0) t.leaf_status[g] = OZ_LEAF_EVAL;
        return;
    }
    int done = sims_done[g];
    if (t.leaf_status[g] == OZ_LEAF_EVAL) ++done;          // the simulation whose leaf the previous step evaluated and backed up
    int status = OZ_LEAF_IDLE;
    for (int it = 0; it < cap; ++it) {
        if (gm.finished[g]) { if (lane == 0) { t.active[g] = 0; t.leaf_status[g] = OZ_LEAF_IDLE; } status = OZ_LEAF_IDLE; break; }
        // roots of the position to move in (k_sp_roots)
        const int p = gm.player[g];
        if (lane == 0) {
            t.active[g] = 1;
            t.root_own[g] = p == 1 ? gm.black[g] : gm.white[g];
            t.root_opp[g] = p == 1 ? gm.white[g] : gm.black[g];
        }
        wave_sync();                                       // lane 0's stores are visible to the wave's next loads
        if (done >= sims) {                                // training.py:42-67: the move after num_simulations simulations
            sp_move_body(gm, t, g, lane, 0);
            done = 0;
            if (lane == 0) t.leaf_status[g] = OZ_LEAF_IDLE;    // the evaluated leaf is consumed: nothing pending if the cap ends the loop here
            wave_sync();
            continue;                                      // (a finished game is refilled by the move, or goes idle above)
        }
        status = select_body(t, L, g, lane);
        if (status != OZ_LEAF_TERMINAL) break;             // EVAL: wait for the network; IDLE: error path
        wave_sync();
        backup_body(t, g, lane, (double)t.term_value[g], VT_INT);
        ++done;
        status = OZ_LEAF_IDLE;
        if (lane == 0) t.leaf_status[g] = OZ_LEAF_IDLE;    // nothing pending if the cap ends the loop here
        wave_sync();
    }
    if (lane == 0) sims_done[g] = done;
}
__global__ __launch_bounds__(64) void k_advance(GamesDev gm, MctsDev t, int sims, int* __restrict__ sims_done, int cap) {
    __shared__ TreeLds L;
    advance_body(gm, t, L, blockIdx.x, threadIdx.x, sims, sims_done, cap);
}
// expand + backup of the leaves the previous batch evaluated and the advance to the next batch's leaves in ONE launch (the free-running
// counterpart of k_backup_select: one kernel boundary less per batch, the records just written are re-read from the CU's cache)
__global__ __launch_bounds__(64) void k_backup_advance(GamesDev gm, MctsDev t, int sims, int* __restrict__ sims_done, int cap) {
    __shared__ TreeLds L;
    expand_backup_body(t, L, blockIdx.x, threadIdx.x, 0);
    wave_sync();
    __syncthreads();
    advance_body(gm, t, L, blockIdx.x, threadIdx.x, sims, sims_done, cap);
}

// RandomOthelloAgent.play (agents.py:20-24) for every live game whose mover is `side`: random.choice over the valid
// actions in ascending row-major order -> the RNG_TIE stream (the fixture generator patches random.choice to it)
__global__ void k_arena_random_move(GamesDev gm, int side) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= gm.G || gm.finished[g] || gm.player[g] != side) return;
    uint64_t black = gm.black[g], white = gm.white[g];
    int player = side, fin = 0;
    const uint64_t legal = oz_legal(side == 1 ? black : white, side == 1 ? white : black, gm.valid);
    const int ply = gm.ply[g];
    const int action = oz_kth_bit(legal, (int)(oz_rng(gm.seed, gm.game_id[g], (uint64_t)ply, OZ_RNG_TIE) % (uint64_t)oz_popc(legal)));
    const size_t lb = (size_t)g * 64;
    if (ply < 64) {
        gm.log_black[lb + ply] = black; gm.log_white[lb + ply] = white;
        gm.log_action[lb + ply] = (uint8_t)action; gm.log_player[lb + ply] = (int8_t)player; gm.log_greedy[lb + ply] = 0;
    }
    oz_game_play(black, white, player, fin, action, gm.valid);
    gm.black[g] = black; gm.white[g] = white; gm.player[g] = (int8_t)player;
    gm.finished[g] = (uint8_t)fin; gm.ply[g] = ply + 1;
    atomicAdd(&gm.counters[2], 1ULL);
}

static void initial_board(int n, uint64_t* black, uint64_t* white) {     // Othello/__init__.py:177-184
    const int h = n / 2;
    *white = (1ULL << ((h - 1) * 8 + h - 1)) | (1ULL << (h * 8 + h));
    *black = (1ULL << ((h - 1) * 8 + h)) | (1ULL << (h * 8 + h - 1));
}

struct oz_selfplay {
    oz_selfplay_config cfg;
    oz_mcts* m = nullptr;
    oz_net* net = nullptr;
    GamesDev gm;
    int* d_sims_done = nullptr;      // free-running mode: simulations completed for the move in progress, per game
    int mode = 0;                    // 0 fresh, 1 driven by oz_selfplay_run (lock step), 2 by oz_selfplay_run_steps (free-running)
    int batch_cap = 0;               // free-running driver: leaves per network batch (0: up to G)
    long long batch_no = 0;          // batches launched by the free-running driver (rotation of the slot order under a cap)
    int stagger_period = 0;
    std::vector<void*> allocs;
    std::mutex mu;
    long long records_read = 0;
    template <typename T> int alloc(T** p, size_t count) {
        OZ_HIP(hipMalloc((void**)p, sizeof(T) * (count ? count : 1)));
        allocs.push_back(*p);
        return OZ_OK;
    }
};

static int games_alloc(oz_selfplay* sp, int G, int n, long long record_cap) {
    GamesDev& gm = sp->gm;
    memset(&gm, 0, sizeof gm);
    gm.G = G; gm.n = n; gm.valid = oz_valid_mask(n); gm.record_cap = record_cap;
    int rc = OZ_OK;
#define A(ptr, cnt) if (!rc) rc = sp->alloc(&gm.ptr, cnt)
    A(black, G); A(white, G); A(game_id, G); A(player, G); A(finished, G); A(ply, G);
    A(log_black, (size_t)G * 64); A(log_white, (size_t)G * 64); A(log_action, (size_t)G * 64);
    A(log_greedy, (size_t)G * 64); A(log_player, (size_t)G * 64); A(last_counts, (size_t)G * 64);
    A(records, (size_t)record_cap); A(counters, 4);
#undef A
    return rc;
}

static int games_init(oz_selfplay* sp, uint64_t first_id, hipStream_t s) {
    GamesDev& gm = sp->gm;
    const int G = gm.G;
    uint64_t b, w;
    initial_board(gm.n, &b, &w);
    gm.init_black = b; gm.init_white = w;
    std::vector<uint64_t> hb(G, b), hw(G, w), ids(G);
    for (int i = 0; i < G; ++i) ids[i] = first_id + (uint64_t)i;
    std::vector<int8_t> pl(G, 1);
    OZ_HIP(hipMemcpyAsync(gm.black, hb.data(), 8ull * G, hipMemcpyHostToDevice, s));
    OZ_HIP(hipMemcpyAsync(gm.white, hw.data(), 8ull * G, hipMemcpyHostToDevice, s));
    OZ_HIP(hipMemcpyAsync(gm.game_id, ids.data(), 8ull * G, hipMemcpyHostToDevice, s));
    OZ_HIP(hipMemcpyAsync(gm.player, pl.data(), G, hipMemcpyHostToDevice, s));
    OZ_HIP(hipMemsetAsync(gm.finished, 0, G, s));
    OZ_HIP(hipMemsetAsync(gm.ply, 0, 4ull * G, s));
    OZ_HIP(hipMemsetAsync(gm.counters, 0, 8 * 4, s));
    OZ_HIP(hipMemsetAsync(gm.last_counts, 0, 4ull * G * 64, s));
    OZ_HIP(hipStreamSynchronize(s));
    return OZ_OK;
}

OZ_API int oz_selfplay_create(oz_selfplay** out, const oz_selfplay_config* cfg, oz_net* net) {
    OZ_REQUIRE(out && cfg && net, "null argument");
    OZ_REQUIRE(cfg->sims >= 2, "num_simulations must be >= 2 (with 1 the reference raises KeyError at othelo_mcts.py:66)");
    OZ_REQUIRE(cfg->num_games > 0, "num_games must be positive");
    OZ_REQUIRE(net->n == cfg->n, "network board size %d != %d", net->n, cfg->n);
    OZ_REQUIRE(net->max_batch >= cfg->num_games, "network max_batch %d < num_games %d", net->max_batch, cfg->num_games);
    OZ_REQUIRE(cfg->dedup == OZ_DEDUP_DEFAULT || cfg->dedup == OZ_DEDUP_ON || cfg->dedup == OZ_DEDUP_OFF, "oz_selfplay_config.dedup = %d", cfg->dedup);
    OZ_REQUIRE(cfg->eval_cache == 0 || cfg->eval_cache == 1, "oz_selfplay_config.eval_cache = %d", cfg->eval_cache);
    OZ_REQUIRE(cfg->batch_cap == 0 || cfg->batch_cap >= 8, "oz_selfplay_config.batch_cap = %d (0 = none, else >= 8 leaves)", cfg->batch_cap);
    oz_selfplay* sp = new oz_selfplay();
    sp->cfg = *cfg; sp->net = net;
    const int max_plies = cfg->n * cfg->n - 4;
    const int node_cap = cfg->node_cap > 0 ? cfg->node_cap : cfg->sims * (max_plies + 1) + 64;
    const long long rcap = cfg->record_cap > 0 ? cfg->record_cap : (long long)cfg->num_games * 64 * 4;
    int rc = mcts_create(&sp->m, cfg->n, cfg->num_games, node_cap, cfg->c, cfg->q_mode);
    if (!rc) rc = games_alloc(sp, cfg->num_games, cfg->n, rcap);
    if (!rc) sp->m->d.dedup = cfg->dedup == OZ_DEDUP_OFF ? 0 : 1;
    if (!rc) sp->batch_cap = cfg->batch_cap;
    if (!rc) rc = sp->alloc(&sp->d_sims_done, cfg->num_games);
    if (!rc && hipMemset(sp->d_sims_done, 0, sizeof(int) * cfg->num_games) != hipSuccess) rc = OZ_ERR_HIP;
    if (!rc) {
        sp->gm.seed = cfg->seed; sp->gm.id_stride = cfg->game_id_stride ? cfg->game_id_stride : (uint64_t)cfg->num_games;
        sp->gm.temperature = cfg->temperature; sp->gm.e_greedy = cfg->e_greedy; sp->gm.refill = cfg->refill;
        rc = games_init(sp, cfg->first_game_id, sp->m->stream);
    }
    if (rc) {
        for (void* p : sp->allocs) hipFree(p);
        mcts_destroy(sp->m);
        delete sp;
        return rc;
    }
    *out = sp;
    return OZ_OK;
}

OZ_API int oz_selfplay_destroy(oz_selfplay* sp) {
    if (!sp) return OZ_OK;
    hipSetDevice(sp->m->device);
    hipStreamSynchronize(sp->m->stream);
    for (void* p : sp->allocs) hipFree(p);
    mcts_destroy(sp->m);
    delete sp;
    return OZ_OK;
}

// cfg.eval_cache: the engine's leaves go through the network's evaluation cache as it is NOW (oz_net_set_eval_cache may have been called,
// or called again, since the engine was created)
static void selfplay_attach_cache(oz_selfplay* sp) {
    sp->m->d.ec = (sp->cfg.eval_cache && sp->net->ec.buckets && sp->net->ec.n2 == sp->m->d.n2) ? sp->net->ec : EvalCacheDev();
}

// one move round on m->stream: roots, `sims` lock-step simulations, move.  stagger_round >= 0: only slots whose start
// offset (oz_selfplay_stagger) is still ahead of that round take part
static int selfplay_round_async(oz_selfplay* sp, int sims, int stagger_round) {
    oz_mcts* m = sp->m;
    const int G = sp->gm.G;
    hipStream_t s = m->stream;
    long long ti = m->profile ? m->timer.begin(TS_MOVE, s) : -1;
    if (stagger_round >= 0) hipLaunchKernelGGL(k_sp_roots_stagger, dim3((G + 255) / 256), dim3(256), 0, s, sp->gm, m->d, stagger_round, sp->stagger_period);
    else hipLaunchKernelGGL(k_sp_roots, dim3((G + 255) / 256), dim3(256), 0, s, sp->gm, m->d, 0);
    m->timer.end(ti, s);
    if (int rc = mcts_steps_async(m, sp->net, sims, true)) return rc;
    ti = m->profile ? m->timer.begin(TS_MOVE, s) : -1;
    hipLaunchKernelGGL(k_sp_move, dim3(G), dim3(64), 0, s, sp->gm, m->d, 0);
    m->timer.end(ti, s);
    OZ_HIP(hipGetLastError());
    if (m->timer.backlog() > 4096) m->timer.drain();      // completed pairs only: never a host stall inside the enqueue loop
    return OZ_OK;
}

OZ_API int oz_selfplay_run(oz_selfplay* sp, int rounds) {
    OZ_REQUIRE(sp, "null selfplay");
    std::lock_guard<std::mutex> lk(sp->mu);
    std::lock_guard<std::mutex> lkn(sp->net->mu);
    hipSetDevice(sp->m->device);
    OZ_REQUIRE(sp->mode != 2, "oz_selfplay_run after oz_selfplay_run_steps: moves are in progress (use one driver per engine)");
    sp->mode = 1;
    selfplay_attach_cache(sp);
    for (int r = 0; r < rounds; ++r)
        if (int rc = selfplay_round_async(sp, sp->cfg.sims, -1)) return rc;
    return OZ_OK;
}

// Spread the slots of a continuous self-play engine over the plies of a game BEFORE measuring it: a service that has been
// running for a while holds games at every stage, whereas a fresh engine holds num_games openings that would all finish in
// the same move round.  Slot g gets the start offset (g * period) / num_games, period = the longest game (n*n - 4 plies):
// in round r (r = 0 .. period-2) the slots whose offset is > r play one move with `sims_pre` simulations -- real searched
// self-play moves by the same kernels and RNG streams, recorded like any other -- the others wait.  Afterwards slot g is
// offset(g) plies into its first game, and the refills keep the spread.  Asynchronous, like oz_selfplay_run.
OZ_API int oz_selfplay_stagger(oz_selfplay* sp, int sims_pre) {
    OZ_REQUIRE(sp, "null selfplay");
    OZ_REQUIRE(sims_pre >= 2, "sims_pre must be >= 2 (the reference raises KeyError with one simulation)");
    std::lock_guard<std::mutex> lk(sp->mu);
    std::lock_guard<std::mutex> lkn(sp->net->mu);
    hipSetDevice(sp->m->device);
    OZ_REQUIRE(sp->mode == 0, "oz_selfplay_stagger must be the first driver call on an engine");
    OZ_REQUIRE(sp->cfg.refill, "oz_selfplay_stagger is for continuous self-play (cfg.refill = 1)");
    sp->mode = 1;
    selfplay_attach_cache(sp);
    sp->stagger_period = sp->cfg.n * sp->cfg.n - 4;
    for (int r = 0; r + 1 < sp->stagger_period; ++r)
        if (int rc = selfplay_round_async(sp, sims_pre, r)) return rc;
    return OZ_OK;
}

OZ_API int oz_selfplay_profile(oz_selfplay* sp, int enable) {
    OZ_REQUIRE(sp, "null selfplay");
    std::lock_guard<std::mutex> lk(sp->mu);
    sp->m->profile = enable != 0;
    return OZ_OK;
}

OZ_API int oz_selfplay_profile_read(oz_selfplay* sp, double* ms_total, int64_t* launches, int reset) {
    OZ_REQUIRE(sp, "null selfplay");
    std::lock_guard<std::mutex> lk(sp->mu);
    hipSetDevice(sp->m->device);
    if (int rc = mcts_collect_eval_time(sp->m)) return rc;
    for (int i = 0; i < OZ_TREE_KERNELS; ++i) {
        if (ms_total) ms_total[i] = sp->m->timer.ms[i];
        if (launches) launches[i] = sp->m->timer.count[i];
    }
    if (reset) sp->m->timer.reset();
    return OZ_OK;
}

OZ_API int oz_selfplay_run_steps(oz_selfplay* sp, int steps) {
    OZ_REQUIRE(sp, "null selfplay");
    std::lock_guard<std::mutex> lk(sp->mu);
    std::lock_guard<std::mutex> lkn(sp->net->mu);
    oz_mcts* m = sp->m;
    hipSetDevice(m->device);
    MctsDev& d = m->d;
    if (sp->mode == 1) OZ_HIP(hipMemsetAsync(d.leaf_status, 0, sizeof(int) * d.G, m->stream));    // no simulation is pending after whole rounds
    sp->mode = 2;
    selfplay_attach_cache(sp);
    const bool fuse = true;
    for (int i = 0; i < steps; ++i) {
        const bool all = m->profile;
        hipStream_t s = m->stream;
        long long ti = all ? m->timer.begin(TS_SELECT, s) : -1;
        // under a batch cap the leaves on offer exceed the slots anyway (waiting games re-offer theirs), so one descent per game and call is
        // enough to keep the batches full and the launch is as short as the lock-step one (measured: +1.2 % expansions/s, +3 % games/s over 2)
        const int adv_cap = (sp->batch_cap > 0 && sp->batch_cap < d.G ? 1 : OZ_ADVANCE_CAP);
        // (from the second batch of a call on, the previous batch's expand + backup rides in the same launch; one closing k_expand_backup per call)
        if (fuse && i > 0) hipLaunchKernelGGL(k_backup_advance, dim3(d.G), dim3(64), 0, s, sp->gm, d, sp->cfg.sims, sp->d_sims_done, adv_cap);
        else hipLaunchKernelGGL(k_advance, dim3(d.G), dim3(64), 0, s, sp->gm, d, sp->cfg.sims, sp->d_sims_done, adv_cap);
        m->timer.end(ti, s);
        ti = all ? m->timer.begin(TS_COMPACT, s) : -1;
        MctsDev dc = d;                                    // this batch's cap and slot order (the kernels take the struct by value)
        const int cap = sp->batch_cap > 0 && sp->batch_cap < d.G ? sp->batch_cap : 0;
        dc.batch_cap = cap;
        dc.batch_rot = cap ? (int)((sp->batch_no * (long long)cap) % d.G) : 0;
        sp->batch_no += 1;
        hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, s, dc);
        m->timer.end(ti, s);
        // (the network's launches are sized for the cap: it picks its tile shapes from the batch it is asked to hold)
        if (int rc = eval_batch_async(m, sp->net, cap ? cap : d.G, true)) return rc;
        if (!fuse || i == steps - 1) {
            ti = all ? m->timer.begin(TS_BACKUP, s) : -1;
            hipLaunchKernelGGL(k_expand_backup, dim3(d.G), dim3(64), 0, s, d, 0);
            m->timer.end(ti, s);
        }
        OZ_HIP(hipGetLastError());
        if (m->timer.backlog() > 4096) m->timer.drain();      // completed pairs only: never a host stall inside the enqueue loop
    }
    return OZ_OK;
}

OZ_API int oz_selfplay_set_batch_cap(oz_selfplay* sp, int cap) {
    OZ_REQUIRE(sp, "null selfplay");
    std::lock_guard<std::mutex> lk(sp->mu);
    OZ_REQUIRE(cap >= 0, "oz_selfplay_set_batch_cap: cap %d", cap);
    OZ_REQUIRE(cap == 0 || cap >= 8, "oz_selfplay_set_batch_cap: a cap below 8 leaves (got %d)", cap);
    sp->batch_cap = cap;
    return OZ_OK;
}

OZ_API int oz_selfplay_set_dedup(oz_selfplay* sp, int enable) {
    OZ_REQUIRE(sp, "null selfplay");
    std::lock_guard<std::mutex> lk(sp->mu);
    sp->m->d.dedup = enable ? 1 : 0;
    return OZ_OK;
}

OZ_API int oz_selfplay_sync(oz_selfplay* sp) {
    OZ_REQUIRE(sp, "null selfplay");
    std::lock_guard<std::mutex> lk(sp->mu);
    hipSetDevice(sp->m->device);
    if (int rc = check_error_flag(sp->m)) return rc;
    if (int rc = mcts_collect_eval_time(sp->m)) return rc;
    std::lock_guard<std::mutex> lkn(sp->net->mu);
    return sp->net->check();
}

OZ_API int oz_selfplay_get_stats(oz_selfplay* sp, oz_selfplay_stats* out) {
    OZ_REQUIRE(sp && out, "null argument");
    std::lock_guard<std::mutex> lk(sp->mu);
    hipSetDevice(sp->m->device);
    int64_t s5[OZ_NSTAT];
    if (int rc = mcts_stats_locked(sp->m, s5)) return rc;
    unsigned long long c[4];
    OZ_HIP(hipMemcpy(c, sp->gm.counters, sizeof c, hipMemcpyDeviceToHost));
    std::vector<uint8_t> fin(sp->gm.G);
    OZ_HIP(hipMemcpy(fin.data(), sp->gm.finished, sp->gm.G, hipMemcpyDeviceToHost));
    int ef = 0;
    OZ_HIP(hipMemcpy(&ef, sp->m->d.error_flag, 4, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof *out);
    out->simulations = s5[ST_SIMS]; out->node_visits = s5[ST_VISITS]; out->expansions = s5[ST_EXPAND];
    out->terminal_hits = s5[ST_TERMINAL]; out->fallbacks = s5[ST_FALLBACK];
    out->records = (int64_t)c[0]; out->games_completed = (int64_t)c[1]; out->moves = (int64_t)c[2];
    int live = 0;
    for (uint8_t f : fin) live += f ? 0 : 1;
    out->live_games = live; out->overflow = ef;
    unsigned long long ev = 0;
    OZ_HIP(hipMemcpy(&ev, sp->m->d.eval_leaves, sizeof ev, hipMemcpyDeviceToHost));
    out->leaves_evaluated = (int64_t)ev;
    return OZ_OK;
}

OZ_API int oz_selfplay_state(oz_selfplay* sp, uint64_t* black, uint64_t* white, int8_t* player, uint8_t* finished,
                             int32_t* ply, uint64_t* game_id) {
    OZ_REQUIRE(sp, "null selfplay");
    std::lock_guard<std::mutex> lk(sp->mu);
    hipSetDevice(sp->m->device);
    OZ_HIP(hipStreamSynchronize(sp->m->stream));
    const int G = sp->gm.G;
    if (black) OZ_HIP(hipMemcpy(black, sp->gm.black, 8ull * G, hipMemcpyDeviceToHost));
    if (white) OZ_HIP(hipMemcpy(white, sp->gm.white, 8ull * G, hipMemcpyDeviceToHost));
    if (player) OZ_HIP(hipMemcpy(player, sp->gm.player, G, hipMemcpyDeviceToHost));
    if (finished) OZ_HIP(hipMemcpy(finished, sp->gm.finished, G, hipMemcpyDeviceToHost));
    if (ply) OZ_HIP(hipMemcpy(ply, sp->gm.ply, 4ull * G, hipMemcpyDeviceToHost));
    if (game_id) OZ_HIP(hipMemcpy(game_id, sp->gm.game_id, 8ull * G, hipMemcpyDeviceToHost));
    return OZ_OK;
}

static int selfplay_records(oz_selfplay* sp, void* dst, int64_t max_records, int64_t* written, hipMemcpyKind kind) {
    OZ_REQUIRE(sp && written, "null argument");
    std::lock_guard<std::mutex> lk(sp->mu);
    hipSetDevice(sp->m->device);
    OZ_HIP(hipStreamSynchronize(sp->m->stream));
    unsigned long long total = 0;
    OZ_HIP(hipMemcpy(&total, sp->gm.counters, 8, hipMemcpyDeviceToHost));
    long long n = (long long)total;
    if (n > sp->gm.record_cap) n = sp->gm.record_cap;
    if (n > max_records) n = max_records;
    if (n > 0) {
        OZ_REQUIRE(dst, "null destination");
        OZ_HIP(hipMemcpy(dst, sp->gm.records, sizeof(oz_record) * (size_t)n, kind));
    }
    *written = n;
    return OZ_OK;
}
OZ_API int oz_selfplay_records(oz_selfplay* sp, oz_record* out, int64_t max_records, int64_t* written) {
    return selfplay_records(sp, out, max_records, written, hipMemcpyDeviceToHost);
}
OZ_API int oz_selfplay_records_device(oz_selfplay* sp, void* dst_device, int64_t max_records, int64_t* written) {
    return selfplay_records(sp, dst_device, max_records, written, hipMemcpyDeviceToDevice);
}

OZ_API int oz_selfplay_last_counts(oz_selfplay* sp, int32_t* counts) {
    OZ_REQUIRE(sp && counts, "null argument");
    std::lock_guard<std::mutex> lk(sp->mu);
    hipSetDevice(sp->m->device);
    OZ_HIP(hipStreamSynchronize(sp->m->stream));
    OZ_HIP(hipMemcpy(counts, sp->gm.last_counts, 4ull * sp->gm.G * 64, hipMemcpyDeviceToHost));
    return OZ_OK;
}

OZ_API int oz_selfplay_eval_time(oz_selfplay* sp, double* ms_total, int64_t* launches, int64_t* leaves) {
    OZ_REQUIRE(sp, "null selfplay");
    std::lock_guard<std::mutex> lk(sp->mu);
    hipSetDevice(sp->m->device);
    if (int rc = mcts_collect_eval_time(sp->m)) return rc;
    if (ms_total) *ms_total = sp->m->timer.ms[TS_NN];
    if (launches) *launches = sp->m->timer.count[TS_NN];
    if (leaves) {
        int64_t s5[OZ_NSTAT];
        if (int rc = mcts_stats_locked(sp->m, s5)) return rc;
        *leaves = s5[ST_EXPAND];
    }
    return OZ_OK;
}

// ================================================================ arena
struct oz_arena {
    oz_selfplay games;           // reuses the game-state arrays (games.m = agent A's search)
    oz_mcts* mb = nullptr;       // agent B's search
    oz_net *na = nullptr, *nb = nullptr;
    int sims = 0;
    std::mutex mu;
    // full move lists (arena games are not refilled): [G][128]
    uint8_t* d_actions = nullptr; int8_t* d_players = nullptr;
    int* d_nmoves = nullptr;
    int* d_movers = nullptr;     // [2] live games with BLACK / WHITE to move (k_arena_movers)
    int eval_cache = 0;          // oz_arena_set_eval_cache: the two searches look their leaves up in (and insert them into) their networks' evaluation caches
};

// live games per mover: counts[0] = BLACK to move, counts[1] = WHITE to move (an agent with nothing to move this round is not launched)
__global__ void k_arena_movers(GamesDev gm, int* counts) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = g < gm.G && !gm.finished[g];
    const uint64_t b = __ballot(live && gm.player[g] == 1), w = __ballot(live && gm.player[g] == -1);
    if ((threadIdx.x & 63) == 0) {
        if (b) atomicAdd(&counts[0], oz_popc(b));
        if (w) atomicAdd(&counts[1], oz_popc(w));
    }
}

// copies the slot's move log into the arena move list before k_sp_move overwrites nothing (log is per ply)
__global__ void k_arena_collect(GamesDev gm, uint8_t* actions, int8_t* players, int* nmoves) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= gm.G) return;
    const int k = gm.ply[g];
    nmoves[g] = k;
    for (int i = 0; i < k && i < 64; ++i) {
        actions[(size_t)g * 128 + i] = gm.log_action[(size_t)g * 64 + i];
        players[(size_t)g * 128 + i] = gm.log_player[(size_t)g * 64 + i];
    }
}

OZ_API int oz_arena_create(oz_arena** out, int n, int num_games, int sims, double c, int q_mode, uint64_t seed,
                           uint64_t first_game_id, oz_net* net_a, oz_net* net_b, int node_cap) {
    OZ_REQUIRE(out && (net_a || net_b), "null argument (at most one of the two networks may be NULL = RandomOthelloAgent)");
    OZ_REQUIRE(sims >= 2, "num_simulations must be >= 2");
    OZ_REQUIRE((!net_a || net_a->n == n) && (!net_b || net_b->n == n), "network board size mismatch");
    OZ_REQUIRE((!net_a || net_a->max_batch >= num_games) && (!net_b || net_b->max_batch >= num_games), "network max_batch < num_games");
    oz_arena* a = new oz_arena();
    a->na = net_a; a->nb = net_b; a->sims = sims;
    const int max_plies = n * n - 4;
    // each agent searches only on its own turns: about half the plies
    const int ncap = node_cap > 0 ? node_cap : sims * (max_plies / 2 + 2) + 64;
    oz_selfplay* sp = &a->games;
    memset(&sp->cfg, 0, sizeof sp->cfg);
    sp->cfg.n = n; sp->cfg.num_games = num_games; sp->cfg.sims = sims; sp->cfg.c = c; sp->cfg.q_mode = q_mode; sp->cfg.seed = seed;
    int rc = mcts_create(&sp->m, n, num_games, ncap, c, q_mode);
    if (!rc) rc = mcts_create(&a->mb, n, num_games, ncap, c, q_mode);
    if (!rc) rc = games_alloc(sp, num_games, n, (long long)num_games * 64);
    if (!rc) {
        sp->gm.seed = seed; sp->gm.id_stride = 0; sp->gm.temperature = 0; sp->gm.e_greedy = 1.0; sp->gm.refill = 0;
        rc = games_init(sp, first_game_id, sp->m->stream);
    }
    if (!rc && hipMalloc((void**)&a->d_actions, (size_t)num_games * 128) != hipSuccess) rc = OZ_ERR_HIP;
    if (!rc && hipMalloc((void**)&a->d_players, (size_t)num_games * 128) != hipSuccess) rc = OZ_ERR_HIP;
    if (!rc && hipMalloc((void**)&a->d_nmoves, 4ull * num_games) != hipSuccess) rc = OZ_ERR_HIP;
    if (!rc && hipMalloc((void**)&a->d_movers, 8) != hipSuccess) rc = OZ_ERR_HIP;
    if (!rc && (hipMemset(a->d_actions, 0, (size_t)num_games * 128) != hipSuccess || hipMemset(a->d_players, 0, (size_t)num_games * 128) != hipSuccess)) rc = OZ_ERR_HIP;   // entries beyond n_moves read 0
    if (rc) {
        for (void* p : sp->allocs) hipFree(p);
        mcts_destroy(sp->m); mcts_destroy(a->mb);
        if (a->d_actions) hipFree(a->d_actions);
        if (a->d_players) hipFree(a->d_players);
        if (a->d_nmoves) hipFree(a->d_nmoves);
        if (a->d_movers) hipFree(a->d_movers);
        delete a;
        if (rc == OZ_ERR_HIP) oz_set_error("arena allocation failed");
        return rc;
    }
    *out = a;
    return OZ_OK;
}

OZ_API int oz_arena_destroy(oz_arena* a) {
    if (!a) return OZ_OK;
    hipSetDevice(a->games.m->device);
    hipStreamSynchronize(a->games.m->stream);
    hipStreamSynchronize(a->mb->stream);
    for (void* p : a->games.allocs) hipFree(p);
    mcts_destroy(a->games.m); mcts_destroy(a->mb);
    hipFree(a->d_actions); hipFree(a->d_players); hipFree(a->d_nmoves); hipFree(a->d_movers);
    delete a;
    return OZ_OK;
}

OZ_API int oz_arena_run(oz_arena* a) { return oz_arena_run_rounds(a, 0); }

// counters of the two agents' searches since creation (layout of oz_mcts_stats); a RandomOthelloAgent side reads zeros
OZ_API int oz_arena_stats(oz_arena* a, int64_t* black5, int64_t* white5) {
    OZ_REQUIRE(a && black5 && white5, "null argument");
    std::lock_guard<std::mutex> lk(a->mu);
    hipSetDevice(a->games.m->device);
    if (int rc = mcts_stats_locked(a->games.m, black5)) return rc;
    return mcts_stats_locked(a->mb, white5);
}

// cross-game leaf de-duplication of both agents' searches (default on, as everywhere in the library; results are identical either way)
OZ_API int oz_arena_set_dedup(oz_arena* a, int enable) {
    OZ_REQUIRE(a, "null arena");
    std::lock_guard<std::mutex> lk(a->mu);
    a->games.m->d.dedup = enable ? 1 : 0;
    a->mb->d.dedup = enable ? 1 : 0;
    return OZ_OK;
}
// HIP-event timing of both agents' tree kernels (slots of oz_selfplay_profile; the sums of the two searches)
OZ_API int oz_arena_profile(oz_arena* a, int enable) {
    OZ_REQUIRE(a, "null arena");
    std::lock_guard<std::mutex> lk(a->mu);
    a->games.m->profile = enable != 0;
    a->mb->profile = enable != 0;
    return OZ_OK;
}
OZ_API int oz_arena_profile_read(oz_arena* a, double* ms_total, int64_t* launches, int reset) {
    OZ_REQUIRE(a, "null arena");
    std::lock_guard<std::mutex> lk(a->mu);
    hipSetDevice(a->games.m->device);
    oz_mcts* both[2] = {a->games.m, a->mb};
    for (oz_mcts* m : both) if (int rc = mcts_collect_eval_time(m)) return rc;
    for (int i = 0; i < OZ_TREE_KERNELS; ++i) {
        if (ms_total) ms_total[i] = both[0]->timer.ms[i] + both[1]->timer.ms[i];
        if (launches) launches[i] = both[0]->timer.count[i] + both[1]->timer.count[i];
    }
    if (reset) for (oz_mcts* m : both) m->timer.reset();
    return OZ_OK;
}
// positions the two agents' networks have evaluated so far (<= expansions when boards are shared between games)
OZ_API int oz_arena_leaves_evaluated(oz_arena* a, int64_t* black, int64_t* white) {
    OZ_REQUIRE(a && black && white, "null argument");
    std::lock_guard<std::mutex> lk(a->mu);
    hipSetDevice(a->games.m->device);
    OZ_HIP(hipStreamSynchronize(a->games.m->stream));
    unsigned long long eb = 0, ew = 0;
    OZ_HIP(hipMemcpy(&eb, a->games.m->d.eval_leaves, sizeof eb, hipMemcpyDeviceToHost));
    OZ_HIP(hipMemcpy(&ew, a->mb->d.eval_leaves, sizeof ew, hipMemcpyDeviceToHost));
    *black = (int64_t)eb; *white = (int64_t)ew;
    return OZ_OK;
}

// the persistent evaluation caches of the two networks (oz_net_set_eval_cache) serve the arena's searches too: arena games start from ONE opening
// and play deterministically, so across games, plies and steps most boards recur -- the reference's own per-search _predict_cache
// (othelo_mcts.py:13,82-88) generalised to everything that uses the network.  A hit changes no bit (a position's (pi, v) does not depend on the
// batch it is evaluated in): moves, boards and results are those of the uncached arena.  Default off, like oz_selfplay_config.eval_cache.
OZ_API int oz_arena_set_eval_cache(oz_arena* a, int enable) {
    OZ_REQUIRE(a, "null arena");
    std::lock_guard<std::mutex> lk(a->mu);
    a->eval_cache = enable ? 1 : 0;
    return OZ_OK;
}

OZ_API int oz_arena_run_rounds(oz_arena* a, int max_rounds_arg) {
    OZ_REQUIRE(a, "null arena");
    OZ_REQUIRE(max_rounds_arg >= 0, "oz_arena_run_rounds: max_rounds %d", max_rounds_arg);
    std::lock_guard<std::mutex> lk(a->mu);
    oz_selfplay* sp = &a->games;
    oz_mcts *ma = sp->m, *mb = a->mb;
    hipSetDevice(ma->device);
    // (the caches as they are NOW: oz_net_set_eval_cache may have been called since the arena was created.  The two searches run one after
    //  the other on one stream, so even ONE network playing itself never has two users of its cache at a time.)
    ma->d.ec = (a->eval_cache && a->na && a->na->ec.buckets && a->na->ec.n2 == ma->d.n2) ? a->na->ec : EvalCacheDev();
    mb->d.ec = (a->eval_cache && a->nb && a->nb->ec.buckets && a->nb->ec.n2 == mb->d.n2) ? a->nb->ec : EvalCacheDev();
    const int G = sp->gm.G;
    hipStream_t s = ma->stream;          // both searches are driven on agent A's stream (they share game state)
    hipStream_t sb_saved = mb->stream;
    mb->stream = s;
    int rc = OZ_OK;
    // every round plays one ply in every live game (two where a RandomOthelloAgent side moves first); n*n rounds end every game
    const int max_rounds = max_rounds_arg > 0 && max_rounds_arg < sp->gm.n * sp->gm.n ? max_rounds_arg : sp->gm.n * sp->gm.n;
    std::vector<uint8_t> fin(G);
    for (int round = 0; round < max_rounds && !rc; ++round) {
        // BLACK movers search in agent A's tables with net A, WHITE movers in agent B's with net B
        // a NULL network = RandomOthelloAgent on that colour: it moves first in the round (a game may then play two plies
        // in one round, which changes nothing: games are independent and every ply is keyed by (game id, ply))
        if (!a->na) hipLaunchKernelGGL(k_arena_random_move, dim3((G + 255) / 256), dim3(256), 0, s, sp->gm, 1);
        if (!a->nb) hipLaunchKernelGGL(k_arena_random_move, dim3((G + 255) / 256), dim3(256), 0, s, sp->gm, -1);
        // who has to move?  Without passes every game of a round has the same mover, so one of the two agents has nothing to search: its
        // a->sims steps (11 launches each over zero leaves) are skipped -- one 8-byte read-back per round buys ~10 % at 800 sims per move.
        // (Round 5, measured and removed: BOTH agents in every round -- the odd slots held back one ply, agent A's and agent B's chains on two
        //  streams, k-splits sized for half batches -- 350.6 against 348.7 ms per ply of 512 games in an A/B on one device: the second chain's
        //  kernels run beside the first one's and slow them by what they gain, the chip is power limited)
        int movers[2] = {0, 0};
        hipMemsetAsync(a->d_movers, 0, sizeof movers, s);
        hipLaunchKernelGGL(k_arena_movers, dim3((G + 255) / 256), dim3(256), 0, s, sp->gm, a->d_movers);
        hipMemcpyAsync(movers, a->d_movers, sizeof movers, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) { oz_set_error("arena: stream failed"); rc = OZ_ERR_HIP; break; }
        if (movers[0] + movers[1] == 0) break;               // every game is over
        const bool run_a = a->na && movers[0] > 0, run_b = a->nb && movers[1] > 0;
        if (run_a) hipLaunchKernelGGL(k_sp_roots, dim3((G + 255) / 256), dim3(256), 0, s, sp->gm, ma->d, 1);
        if (run_b) hipLaunchKernelGGL(k_sp_roots, dim3((G + 255) / 256), dim3(256), 0, s, sp->gm, mb->d, -1);
        // the evaluator's launches are made for the movers of the round (an upper bound of the leaves a batch can hold), not for all G slots
        if (run_a) {
            std::lock_guard<std::mutex> la(a->na->mu);
            for (int k = 0; k < a->sims && !rc; ++k) rc = mcts_step_k(ma, a->na, k, movers[0], false);
            if (!rc) rc = mcts_steps_close(ma);
        }
        if (!rc && run_b) {
            std::lock_guard<std::mutex> lb(a->nb->mu);
            for (int k = 0; k < a->sims && !rc; ++k) rc = mcts_step_k(mb, a->nb, k, movers[1], false);
            if (!rc) rc = mcts_steps_close(mb);
        }
        if (rc) break;
        {
            const long long ta = (run_a && ma->profile) ? ma->timer.begin(TS_MOVE, s) : -1;
            if (run_a) hipLaunchKernelGGL(k_sp_move, dim3(G), dim3(64), 0, s, sp->gm, ma->d, 1);
            ma->timer.end(ta, s);
            const long long tb = (run_b && mb->profile) ? mb->timer.begin(TS_MOVE, s) : -1;
            if (run_b) hipLaunchKernelGGL(k_sp_move, dim3(G), dim3(64), 0, s, sp->gm, mb->d, 1);
            mb->timer.end(tb, s);
        }
        if (hipGetLastError() != hipSuccess) { oz_set_error("arena kernel launch failed"); rc = OZ_ERR_HIP; break; }
        if ((round & 3) == 3 || round + 1 == max_rounds) {
            if ((rc = check_error_flag(ma))) break;
            if ((rc = check_error_flag(mb))) break;
            if (hipMemcpy(fin.data(), sp->gm.finished, G, hipMemcpyDeviceToHost) != hipSuccess) { oz_set_error("memcpy failed"); rc = OZ_ERR_HIP; break; }
            bool all = true;
            for (uint8_t f : fin) all = all && f;
            if (all) break;
        }
    }
    if (!rc) rc = check_error_flag(ma);
    if (!rc) rc = check_error_flag(mb);
    mb->stream = sb_saved;
    if (rc) return rc;
    OZ_HIP(hipStreamSynchronize(s));
    return OZ_OK;
}

OZ_API int oz_arena_results(oz_arena* a, int8_t* winner, int32_t* points, int32_t* n_moves, uint8_t* actions,
                            int8_t* players, uint64_t* final_black, uint64_t* final_white) {
    OZ_REQUIRE(a, "null arena");
    std::lock_guard<std::mutex> lk(a->mu);
    oz_selfplay* sp = &a->games;
    hipSetDevice(sp->m->device);
    const int G = sp->gm.G;
    hipLaunchKernelGGL(k_arena_collect, dim3((G + 255) / 256), dim3(256), 0, sp->m->stream, sp->gm, a->d_actions, a->d_players, a->d_nmoves);
    OZ_HIP(hipStreamSynchronize(sp->m->stream));
    std::vector<uint64_t> b(G), w(G);
    OZ_HIP(hipMemcpy(b.data(), sp->gm.black, 8ull * G, hipMemcpyDeviceToHost));
    OZ_HIP(hipMemcpy(w.data(), sp->gm.white, 8ull * G, hipMemcpyDeviceToHost));
    for (int g = 0; g < G; ++g) {
        const int pb = __builtin_popcountll(b[g]), pw = __builtin_popcountll(w[g]);
        if (winner) winner[g] = pb >= pw ? 1 : -1;               // draw -> BLACK agent (agents.py:83-84)
        if (points) points[g] = pb >= pw ? pb : pw;
        if (final_black) final_black[g] = b[g];
        if (final_white) final_white[g] = w[g];
    }
    if (n_moves) OZ_HIP(hipMemcpy(n_moves, a->d_nmoves, 4ull * G, hipMemcpyDeviceToHost));
    if (actions) OZ_HIP(hipMemcpy(actions, a->d_actions, (size_t)G * 128, hipMemcpyDeviceToHost));
    if (players) OZ_HIP(hipMemcpy(players, a->d_players, (size_t)G * 128, hipMemcpyDeviceToHost));
    return OZ_OK;
}

// ================================================================ exchange step over RCCL (SURVEY.md 8(b) gather_examples(comm), 8(e))
// The path's ONE collective -- the all-gather of the move records of the games a self-play batch completed -- behind the C ABI, for
// hosts that are not Python (othellozero_amd/distributed.py does the same through torch.distributed).  RCCL is bound at run time
// (dlopen of librccl.so.1: a process that already carries an RCCL -- PyTorch ships its own -- keeps using that one; no link-time
// dependency for single-GPU users).  Replaces WorkerManager.get_results' list concatenation (workers.py:180-184) and the ssh / pickle
// return path (workers.py:147-159).
#include <dlfcn.h>
namespace {
struct Rccl {
    struct Id { char b[OZ_COMM_ID_BYTES]; };                 // ncclUniqueId: 128 bytes, passed BY VALUE to ncclCommInitRank
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;
int rccl_load() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return OZ_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* nm : names) if ((h = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) { oz_set_error("RCCL not available: %s", dlerror()); return OZ_ERR_STATE; }
    Rccl r;
    r.h = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) {
        oz_set_error("the RCCL library lacks an expected symbol");
        dlclose(h);
        return OZ_ERR_STATE;
    }
    g_rccl = r;
    return OZ_OK;
}
}  // namespace
#define OZ_NCCL(call) do { int rc__ = (call); if (rc__ != 0) { oz_set_error("RCCL: %s (%s)", g_rccl.GetErrorString(rc__), #call); return OZ_ERR_HIP; } } while (0)
enum { OZ_NCCL_UINT8 = 1, OZ_NCCL_INT64 = 4 };              // ncclDataType_t (rccl.h)

struct oz_comm {
    void* comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    long long* d_counts = nullptr;        // [2 + 2 * world]: this rank's (count, room), then every rank's pair
    unsigned char *d_send = nullptr, *d_recv = nullptr;
    long long cap = 0;                    // records per rank the two buffers hold
    std::mutex mu;
};

OZ_API int oz_comm_unique_id(uint8_t* id) {
    OZ_REQUIRE(id, "null id");
    if (int rc = rccl_load()) return rc;
    oz_current_device();
    OZ_NCCL(g_rccl.GetUniqueId(id));
    return OZ_OK;
}

OZ_API int oz_comm_create(oz_comm** out, const uint8_t* id, int rank, int world) {
    OZ_REQUIRE(out && id, "null argument");
    OZ_REQUIRE(world >= 1 && rank >= 0 && rank < world, "oz_comm_create: rank %d of %d", rank, world);
    if (int rc = rccl_load()) return rc;
    oz_comm* c = new oz_comm();
    c->rank = rank; c->world = world; c->device = oz_current_device();
    Rccl::Id uid;
    memcpy(uid.b, id, OZ_COMM_ID_BYTES);
    int nrc = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (nrc != 0) { oz_set_error("RCCL: %s (ncclCommInitRank, rank %d of %d)", g_rccl.GetErrorString(nrc), rank, world); delete c; return OZ_ERR_HIP; }
    if (hipStreamCreate(&c->stream) != hipSuccess || hipMalloc((void**)&c->d_counts, sizeof(long long) * (2 * world + 2)) != hipSuccess) {
        oz_set_error("oz_comm_create: stream / buffer allocation failed");
        g_rccl.CommDestroy(c->comm);
        delete c;
        return OZ_ERR_HIP;
    }
    *out = c;
    return OZ_OK;
}

OZ_API int oz_comm_destroy(oz_comm* c) {
    if (!c) return OZ_OK;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (c->comm) g_rccl.CommDestroy(c->comm);
    hipFree(c->d_counts); hipFree(c->d_send); hipFree(c->d_recv);
    hipStreamDestroy(c->stream);
    delete c;
    return OZ_OK;
}

// COLLECTIVE: every rank of `comm` calls it with its own engine.  The records [first_record, records completed so far) of every rank's
// engine, concatenated in rank order, into `out` (host, caller-owned, max_records long; per_rank[world] gets every rank's count).
// A (count, room) pair all-gather (16 bytes per rank) + ONE padded all-gather of 48-byte records, device to device out of the engines' HBM
// buffers.  Every decision that could end the call on one rank is taken from the GATHERED pairs, i.e. identically on every rank (a
// rank-local capacity check let one rank leave while its peers entered the payload all-gather; ADVICE r4: so did a rank-local argument or HIP
// error BEFORE the first all-gather): a rank whose own arguments are bad (first_record < 0, a null buffer with room, the communicator on
// another device) or whose record counter cannot be read joins the pair all-gather with room = -2, "cannot take part", and EVERY rank fails
// together -- that rank with its own message, the others naming it; too little room on ANY rank fails the call on EVERY rank before the payload
// moves; a failed buffer allocation is agreed on by one more 8-byte all-gather.  What stays rank-local: null sp / comm / written (there is no
// communicator to say so with) and an RCCL / HIP failure of the collectives themselves (the communicator is then broken for everybody).
// out == NULL with max_records == 0 on every rank = the counts only (*written = the pooled number, per_rank filled): size the buffer, call again.
OZ_API int oz_selfplay_gather_records(oz_selfplay* sp, oz_comm* c, int64_t first_record, oz_record* out, int64_t max_records, int64_t* written,
                                      int64_t* per_rank) {
    OZ_REQUIRE(sp && c && written, "null argument");
    std::lock_guard<std::mutex> lk(sp->mu);
    std::lock_guard<std::mutex> lkc(c->mu);
    hipSetDevice(c->device);
    char local_bad[256] = "";
    if (first_record < 0) snprintf(local_bad, sizeof local_bad, "first_record %lld", (long long)first_record);
    else if (!out && max_records != 0) snprintf(local_bad, sizeof local_bad, "null output buffer with room for %lld records", (long long)max_records);
    else if (c->device != sp->m->device) snprintf(local_bad, sizeof local_bad, "the communicator lives on device %d, the engine on device %d", c->device, sp->m->device);
    unsigned long long total = 0;
    if (!local_bad[0]) {
        hipError_t e = hipStreamSynchronize(sp->m->stream);
        if (e == hipSuccess) e = hipMemcpy(&total, sp->gm.counters, 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { snprintf(local_bad, sizeof local_bad, "reading the engine's record counter failed: %s", hipGetErrorString(e)); (void)hipGetLastError(); }
    }
    long long have = (long long)total < sp->gm.record_cap ? (long long)total : sp->gm.record_cap;
    const int W = c->world;
    // room: >= 0 = records `out` holds, -1 = "counts only", -2 = "this rank cannot take part"
    long long pair[2] = {have > first_record ? have - first_record : 0, out ? (long long)max_records : -1};
    if (local_bad[0]) { pair[0] = 0; pair[1] = -2; }
    const long long mine = pair[0];
    OZ_HIP(hipMemcpyAsync(c->d_counts, pair, sizeof pair, hipMemcpyHostToDevice, c->stream));
    OZ_NCCL(g_rccl.AllGather(c->d_counts, c->d_counts + 2, 2, OZ_NCCL_INT64, c->comm, c->stream));
    std::vector<long long> pairs((size_t)2 * W);
    OZ_HIP(hipMemcpyAsync(pairs.data(), c->d_counts + 2, sizeof(long long) * 2 * W, hipMemcpyDeviceToHost, c->stream));
    OZ_HIP(hipStreamSynchronize(c->stream));
    *written = 0;
    if (local_bad[0]) { oz_set_error("oz_selfplay_gather_records: %s (every rank fails together)", local_bad); return OZ_ERR_ARG; }
    for (int r = 0; r < W; ++r)
        if (pairs[2 * r + 1] == -2) { oz_set_error("oz_selfplay_gather_records: rank %d could not take part (bad arguments or a local error there): every rank fails together", r); return OZ_ERR_STATE; }
    long long mx = 0, sum = 0;
    int counts_only = 0, short_rank = -1;
    for (int r = 0; r < W; ++r) {
        const long long cnt = pairs[2 * r];
        mx = cnt > mx ? cnt : mx; sum += cnt;
        if (per_rank) per_rank[r] = cnt;
        if (pairs[2 * r + 1] < 0) ++counts_only;
    }
    for (int r = 0; r < W && short_rank < 0; ++r) if (pairs[2 * r + 1] >= 0 && pairs[2 * r + 1] < sum) short_rank = r;
    if (counts_only == W) { *written = sum; return OZ_OK; }
    OZ_REQUIRE(counts_only == 0, "oz_selfplay_gather_records: %d of %d ranks asked for the counts only (out == NULL), the others for the records", counts_only, W);
    OZ_REQUIRE(short_rank < 0, "oz_selfplay_gather_records: %lld pooled records, rank %d has room for %lld (every rank fails together)", sum, short_rank,
               pairs[2 * short_rank + 1]);
    if (mx == 0) return OZ_OK;
    if (mx > c->cap) {                                        // grow the padded send / receive buffers (kept with the communicator); every rank's cap
        hipFree(c->d_send); hipFree(c->d_recv);               // has followed the same mx history, so all of them are here together
        c->d_send = c->d_recv = nullptr; c->cap = 0;
        const long long cap = mx + mx / 4 + 1024;
        long long failed = hipMalloc((void**)&c->d_send, (size_t)cap * sizeof(oz_record)) != hipSuccess ||
                           hipMalloc((void**)&c->d_recv, (size_t)cap * sizeof(oz_record) * W) != hipSuccess;
        if (!failed) c->cap = cap;
        else { hipFree(c->d_send); hipFree(c->d_recv); c->d_send = c->d_recv = nullptr; (void)hipGetLastError(); }
        OZ_HIP(hipMemcpyAsync(c->d_counts, &failed, sizeof failed, hipMemcpyHostToDevice, c->stream));
        OZ_NCCL(g_rccl.AllGather(c->d_counts, c->d_counts + 2, 1, OZ_NCCL_INT64, c->comm, c->stream));
        std::vector<long long> st((size_t)W);
        OZ_HIP(hipMemcpyAsync(st.data(), c->d_counts + 2, sizeof(long long) * W, hipMemcpyDeviceToHost, c->stream));
        OZ_HIP(hipStreamSynchronize(c->stream));
        for (int r = 0; r < W; ++r)
            if (st[r]) { oz_set_error("oz_selfplay_gather_records: rank %d could not allocate its exchange buffers (%lld records x %d ranks)", r, cap, W); return OZ_ERR_HIP; }
    }
    if (mine) OZ_HIP(hipMemcpyAsync(c->d_send, sp->gm.records + first_record, (size_t)mine * sizeof(oz_record), hipMemcpyDeviceToDevice, c->stream));
    if (mine < mx) OZ_HIP(hipMemsetAsync(c->d_send + (size_t)mine * sizeof(oz_record), 0, (size_t)(mx - mine) * sizeof(oz_record), c->stream));
    OZ_NCCL(g_rccl.AllGather(c->d_send, c->d_recv, (size_t)mx * sizeof(oz_record), OZ_NCCL_UINT8, c->comm, c->stream));
    long long pos = 0;
    for (int r = 0; r < W; ++r) {
        const long long cnt = pairs[2 * r];
        if (cnt) OZ_HIP(hipMemcpyAsync(out + pos, c->d_recv + (size_t)r * mx * sizeof(oz_record), (size_t)cnt * sizeof(oz_record),
                                       hipMemcpyDeviceToHost, c->stream));
        pos += cnt;
    }
    OZ_HIP(hipStreamSynchronize(c->stream));
    *written = sum;
    return OZ_OK;
}

// ================================================================ diagnostics
// Device arithmetic used by the PUCT / backup formulas, exposed so the parity tests can compare it with the
// host libm / IEEE results bit for bit: sqrt (float64), division (float64, float32), float32 multiply-add chain.
__global__ void k_selftest_arith(const double* a, const double* b, int count, double* sq, double* dv, float* fdv, float* fq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    sq[i] = sqrt(a[i]);
    dv[i] = a[i] / b[i];
    const float x = (float)a[i], y = (float)b[i];
    fdv[i] = x / y;
    fq[i] = (x * y + x) / y;       // must stay mul, add, div (no FMA contraction)
}
OZ_API int oz_selftest_arith(const double* a, const double* b, int count, double* sqrt_a, double* div_ab, float* fdiv_ab,
                             float* fchain) {
    OZ_REQUIRE(a && b && sqrt_a && div_ab && fdiv_ab && fchain && count > 0, "bad argument");
    oz_current_device();
    double *da, *db, *ds, *dd; float *df, *dq;
    OZ_HIP(hipMalloc((void**)&da, 8ull * count)); OZ_HIP(hipMalloc((void**)&db, 8ull * count));
    OZ_HIP(hipMalloc((void**)&ds, 8ull * count)); OZ_HIP(hipMalloc((void**)&dd, 8ull * count));
    OZ_HIP(hipMalloc((void**)&df, 4ull * count)); OZ_HIP(hipMalloc((void**)&dq, 4ull * count));
    hipMemcpy(da, a, 8ull * count, hipMemcpyHostToDevice); hipMemcpy(db, b, 8ull * count, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_selftest_arith, dim3((count + 255) / 256), dim3(256), 0, 0, da, db, count, ds, dd, df, dq);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(sqrt_a, ds, 8ull * count, hipMemcpyDeviceToHost); hipMemcpy(div_ab, dd, 8ull * count, hipMemcpyDeviceToHost);
    hipMemcpy(fdiv_ab, df, 4ull * count, hipMemcpyDeviceToHost); hipMemcpy(fchain, dq, 4ull * count, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(ds); hipFree(dd); hipFree(df); hipFree(dq);
    OZ_HIP(e);
    return OZ_OK;
}

// Sustained matrix-pipe rate of THIS device, right now: a pure-MFMA loop (no LDS, no loads, no barriers; one block per CU, one wave per SIMD,
// four independent accumulators back to back -- the issue pattern of the GEMM kernels' clusters) for about `target_ms`.  kind 0 =
// v_mfma_f32_32x32x2_f32 (what k_gemm_f32 runs on), kind 1 = v_mfma_f32_16x16x32_f16 on operands with every mantissa bit busy (what k_gemm_h2
// runs on; the sustained clock of that pipe follows the operand bits, docs/HISTORY.md section 4).  bench.py puts both numbers into its line as
// `device_calibration`, so that a reader can tell a slow box (or a power-capped one) from a regression: the GEMM kernels' own rate moves
// with this one.  tflops = FLOP of the issued MFMAs / HIP-event time; clock_ghz = the clock at which back-to-back issue (64 / 16 cycles per
// MFMA and SIMD) gives that rate.
typedef float dg_f32x16 __attribute__((ext_vector_type(16)));
typedef float dg_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 dg_f16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void k_diag_mfma_f32(float* out, int iters, float a0, float b0) {
    dg_f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float a = a0 + threadIdx.x * 1.0009765625e-3f, b = b0 + threadIdx.x * 2.001953125e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_diag_mfma_f16(float* out, int iters, unsigned seed) {
    dg_f32x4 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    dg_f16x8 a, b;
    uint64_t r = oz_sm64(seed + 977u * (blockIdx.x * blockDim.x + threadIdx.x));
    for (int j = 0; j < 8; ++j) {                            // values in [2^-3, 2^-2) with random mantissas and signs: the window the network's tensors sit in
        r = oz_sm64(r);
        const unsigned short ha = (unsigned short)(0x3000u | (r & 0x3FFu) | ((r >> 10) & 1u) << 15);
        const unsigned short hb = (unsigned short)(0x3000u | ((r >> 16) & 0x3FFu) | ((r >> 26) & 1u) << 15);
        a[j] = __builtin_bit_cast(_Float16, ha); b[j] = __builtin_bit_cast(_Float16, hb);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r2 = 0; r2 < 4; ++r2) s += acc[i][r2];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
typedef __bf16 dg_bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void k_diag_mfma_bf16(float* out, int iters, unsigned seed) {
    dg_f32x4 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    dg_bf16x8 a, b;
    uint64_t r = oz_sm64(seed + 977u * (blockIdx.x * blockDim.x + threadIdx.x));
    for (int j = 0; j < 8; ++j) {                            // values in [2^-3, 2^-2) with random 7-bit mantissas and signs: what the planes of precision bf16x3 hold
        r = oz_sm64(r);
        const unsigned short ha = (unsigned short)(0x3E00u | (r & 0x7Fu) | ((r >> 10) & 1u) << 15);
        const unsigned short hb = (unsigned short)(0x3E00u | ((r >> 16) & 0x7Fu) | ((r >> 26) & 1u) << 15);
        a[j] = __builtin_bit_cast(__bf16, ha); b[j] = __builtin_bit_cast(__bf16, hb);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r2 = 0; r2 < 4; ++r2) s += acc[i][r2];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
OZ_API int oz_selftest_mfma_rate(int kind, double target_ms, double* tflops, double* clock_ghz, double* ms_measured) {
    OZ_REQUIRE((kind == 0 || kind == 1 || kind == 2) && tflops, "oz_selftest_mfma_rate: kind must be 0 (f32), 1 (f16) or 2 (bf16), tflops non-null");
    OZ_REQUIRE(target_ms > 0 && target_ms <= 2000, "oz_selftest_mfma_rate: target_ms %.1f outside (0, 2000]", target_ms);
    const int dev = oz_current_device();
    hipDeviceProp_t prop;
    OZ_HIP(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    float* out = nullptr;
    OZ_HIP(hipMalloc((void**)&out, sizeof(float) * 256 * (size_t)cus));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop_per_mfma = kind == 0 ? 32.0 * 32 * 2 * 2 : 16.0 * 16 * 32 * 2;
    const double cycles_per_mfma = kind == 0 ? 64.0 : 16.0;  // back-to-back issue on one SIMD
    auto launch = [&](int iters) {
        if (kind == 0) hipLaunchKernelGGL(k_diag_mfma_f32, dim3(cus), dim3(256), 0, 0, out, iters, 0.5f, 0.25f);
        else if (kind == 1) hipLaunchKernelGGL(k_diag_mfma_f16, dim3(cus), dim3(256), 0, 0, out, iters, 12345u);
        else hipLaunchKernelGGL(k_diag_mfma_bf16, dim3(cus), dim3(256), 0, 0, out, iters, 12345u);
    };
    // a short run sizes the long one (and warms the clocks up)
    int iters = kind == 0 ? 400 : 1600;
    float ms = 0.f;
    hipError_t err = hipSuccess;
    for (int pass = 0; pass < 2 && err == hipSuccess; ++pass) {
        hipEventRecord(e0, 0);
        launch(iters);
        hipEventRecord(e1, 0);
        err = hipEventSynchronize(e1);
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
        if (pass == 0 && err == hipSuccess) {
            const double scale = target_ms / (ms > 1e-3f ? ms : 1e-3f);
            const double want = iters * scale;
            iters = (int)(want < 1 ? 1 : want > 2.0e8 ? 2.0e8 : want);
        }
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(out);
    OZ_HIP(err);
    const double mfmas = (double)cus * 4 /* waves = SIMDs */ * iters * 64.0;
    const double rate = mfmas * flop_per_mfma / (ms * 1e-3);
    *tflops = rate / 1e12;
    if (clock_ghz) *clock_ghz = mfmas / ((double)cus * 4) * cycles_per_mfma / (ms * 1e-3) / 1e9;
    if (ms_measured) *ms_measured = ms;
    return OZ_OK;
}
