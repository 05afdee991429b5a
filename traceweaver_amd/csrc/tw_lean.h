// tw_lean.h -- k_enumerate_lean<W>: the wavefront enumeration of the deep call graphs (included by tw_kernels.h).
//
// Same reference functions as k_enumerate_heavy (DfsTraverseX / DfsTraverse3 traceweaver_v3.py:236-351, feasibility :328-347,
// ScoreAssignmentAsPerInvocationGraph traceweaver_v1.py:305-361, the size-5 heap of traceweaver_v3.py:304-307), same work lists, same
// result slots -- another layout of the wavefront's state.  k_enumerate_heavy<E, W> keeps everything that is per endpoint in registers
// indexed at compile time: for seven and eight endpoints that is 256 vector registers + 256 accumulation registers and 2 000 spilled
// scalar registers (the call-order DAG, counts, cut-offs, table offsets of eight endpoints are wave-uniform but far more than the 100
// scalar registers hold), one wavefront per SIMD with nothing beside it, and E^3 select chains per tuple because a register array
// cannot be indexed by a (uniform) run-time value.  Here
//   * the endpoint count is a run-time value; per-endpoint loops are loops;
//   * what is per endpoint and wave-uniform (counts, cut-offs, predecessor masks) lives in LDS and is read at uniform addresses;
//   * a lane's tuple is ONE 64-bit register, eight bits per endpoint (the format of the listed tuples): the position at endpoint p
//     is a shift by a uniform amount, the span's times and terms are LDS look-ups at (endpoint, position);
//   * a tuple's score is a short PROGRAM in LDS, built once per item: one word per addition of ScoreAssignmentAsPerInvocationGraph
//     in the reference's order (pair term of a primary in-edge / root term / closing term), so that the scoring loop is a loop over
//     additions instead of E x E unrolled cases.
// Routes (each visits the feasible tuples in the reference's order; which one an item takes changes nothing in its result):
//   grid      few grid points: every grid point a lane, feasibility tested per tuple;
//   list      call-order constraints and many grid points: the feasible tuples are listed level by level (one lane per prefix, the
//             admissible candidates of the next endpoint are a tail of its start-ordered list), the last level counted first;
//             few tuples: written and scored from the list; many (deferring classes): cut into LIST PARTS by tuples or prefixes
//             (tw_kernels.h, kListSplitFlag), served by the launch with part = 3;
//   fallback  what this kernel leaves to k_enumerate_heavy (part = 4): pruned walks (enumerations without call-order constraints or
//             too long to list), lists that outgrow the wavefront's buffers, pair tables that do not fit their LDS pool, parts cut
//             by the first endpoint's candidate, requests too long for 32-bit... -- rare at scale, listed in P.fb_*.
// The five largest tuples: first under the strict part of Python's order with a watch for undecided ties, then (ties, twin candidates,
// log-mode parts) the replay of CPython's heapq / list.sort push by push -- as in k_enumerate_heavy.
#pragma once

namespace tw {

template <int W>
struct LeanLds {
    unsigned long long sbits[kMaxEp][kCandWords];
    int64_t ls[kMaxEp][W], le[kMaxEp][W];          // staged candidates: start / end
    uint8_t lr[kMaxEp][W];                         // position of the staged candidate in the cut-off window
    int32_t cn[kMaxEp], lo[kMaxEp], wd[kMaxEp];    // staged candidates, first span of the cut-off window, its width
    uint32_t pm[kMaxEp], magic[kMaxEp];            // predecessor masks; reciprocals of the counts (grid decoding)
    uint32_t ops[kMaxEp * kMaxEp + 2 * kMaxEp];    // the scoring program (see lean_op)
    int32_t edge_first[kMaxEp * kMaxEp + 1];       // first pair of every primary in-edge in the item's index space of terms
    uint8_t edge_op[kMaxEp * kMaxEp];              // ... and its word of the program
    int32_t n_ops;
    double hs[kTopK + 1];                          // CPython heap replay: scores / tuples (staged positions, 8 bits each)
    unsigned long long hx[kTopK + 1];
    unsigned long long keep[kTopK];                // strict attempt: the kept tuples by slot
    int32_t defer_lo[kMaxParts + 1];               // first listed prefix of every list part (deferred spans)
};

// one addition of a tuple's score: the term tabs[off + position at p * factor + position at e] -- kind 0 always (pair term of the primary
// in-edge p -> e: factor = the count of e's staged candidates; root term of e: factor 0), kind 1 by the lanes whose last-ending span is
// at e (closing term), kind 2 nothing (padding), kind 3 a pair term from the wavefront's slot of P.pair_pool (off within the slot)
// The LDS pool of pair tables is what limits the wavefronts per CU (an eight-endpoint item with four candidates per endpoint needs
// ~200 doubles, one with fourteen 2 000): the launch asks for a pool that holds the common case (TW_LEAN_POOL, 512 doubles: 11
// wavefronts per CU instead of 6 with 2 048), tables beyond it go to a slot of kPairSpill doubles in global memory (L2-resident).
constexpr int kPairSlots = 4096, kPairSpill = 2048;
__device__ __forceinline__ uint32_t lean_op(int kind, int e, int p, int cn_e, int off) {
    return (uint32_t)kind | ((uint32_t)e << 2) | ((uint32_t)p << 5) | ((uint32_t)cn_e << 8) | ((uint32_t)off << 16);
}
__device__ __forceinline__ int lean_pos(unsigned long long pk, int e) { return (int)((pk >> (8 * e)) & 255ull); }
// CPython's heapq / list.sort on (score, tuple) entries in LDS (lane 0): LdsHeap of k_enumerate_heavy with packed tuples and a
// run-time endpoint count.  Python's order on equal scores: start_mus of the first differing span.
template <int W>
struct LeanHeap {
    double* hs;
    unsigned long long* hx;
    int n, E;
    const int64_t (*ls)[W];
    __device__ bool lt(double sa, unsigned long long xa, double sb, unsigned long long xb) const {
        if (sa != sb) return sa < sb;
        for (int e = 0; e < E; e++) {
            const int a = lean_pos(xa, e), b = lean_pos(xb, e);
            if (a != b) return ls[e][a] < ls[e][b];
        }
        return false;
    }
    __device__ void siftdown(int startpos, int pos) {
        const double is = hs[pos]; const unsigned long long ix = hx[pos];
        while (pos > startpos) {
            const int parent = (pos - 1) >> 1;
            if (lt(is, ix, hs[parent], hx[parent])) { hs[pos] = hs[parent]; hx[pos] = hx[parent]; pos = parent; continue; }
            break;
        }
        hs[pos] = is; hx[pos] = ix;
    }
    __device__ void siftup(int pos) {
        const int startpos = pos;
        const double is = hs[pos]; const unsigned long long ix = hx[pos];
        int child = 2 * pos + 1;
        while (child < n) {
            const int right = child + 1;
            if (right < n && !lt(hs[child], hx[child], hs[right], hx[right])) child = right;
            hs[pos] = hs[child]; hx[pos] = hx[child];
            pos = child;
            child = 2 * pos + 1;
        }
        hs[pos] = is; hx[pos] = ix;
        siftdown(startpos, pos);
    }
    __device__ void push(double s, unsigned long long x) {
        hs[n] = s; hx[n] = x; n++;
        siftdown(0, n - 1);
        if (n > kTopK) {
            n--;
            const double ls2 = hs[n]; const unsigned long long lx = hx[n];
            if (n > 0) { hs[0] = ls2; hx[0] = lx; siftup(0); }
        }
    }
    __device__ void reverse(int m) {
        for (int i = 0, j = m - 1; i < j; i++, j--) {
            const double a = hs[i]; hs[i] = hs[j]; hs[j] = a;
            const unsigned long long b = hx[i]; hx[i] = hx[j]; hx[j] = b;
        }
    }
    __device__ void sort_desc() {   // list.sort(reverse=True) of <= 5 entries: reverse, count_run + binary insertion, reverse
        if (n < 2) return;
        reverse(n);
        int run = 2;
        if (lt(hs[1], hx[1], hs[0], hx[0])) {
            for (int i = 2; i < n; i++, run++) if (!lt(hs[i], hx[i], hs[i - 1], hx[i - 1])) break;
            reverse(run);
        } else {
            for (int i = 2; i < n; i++, run++) if (lt(hs[i], hx[i], hs[i - 1], hx[i - 1])) break;
        }
        for (int start = run; start < n; start++) {
            int l = 0, r = start;
            const double ps = hs[start]; const unsigned long long px = hx[start];
            do {
                const int p = l + ((r - l) >> 1);
                if (lt(ps, px, hs[p], hx[p])) r = p; else l = p + 1;
            } while (l < r);
            for (int p = start; p > l; p--) { hs[p] = hs[p - 1]; hx[p] = hx[p - 1]; }
            hs[l] = ps; hx[l] = px;
        }
        reverse(n);
    }
};

// Enumerations of up to P.lean_grid grid points (TW_LEAN_GRID) are walked as a grid, no list: a batch of 64 grid points costs about what
// a batch of 64 listed tuples costs plus the call-order tests, and with call-order constraints a few per cent of the grid are tuples --
// listing visits only the feasible prefixes (measured on the Alibaba-shape slice: items of 100-200 tuples in grids of 4 096 points
// took 300 us as grids).
#ifndef TW_LEAN_GRID
#define TW_LEAN_GRID 256
#endif
#ifndef TW_LEAN_ATTR
#ifdef TW_HOST_EMULATION
#define TW_LEAN_ATTR
#else
#define TW_LEAN_ATTR __attribute__((amdgpu_waves_per_eu(2, 4)))
#endif
#endif

template <int W>
__global__ void __launch_bounds__(kHeavyThreads) TW_LEAN_ATTR k_enumerate_lean(Dev P, int pass, int mode, int part, int pool, int E_lo, int E_hi) {
    // part: 0 the classes' lists (long enumerations first), 1 the spans k_merge_parts lists again, 3 the list parts of the deferred spans.
    // ONE launch serves the classes E_lo .. E_hi (the endpoint count is a run-time value here): the GPU runs about four kernels at a time
    // whatever their sizes (measured: with a chain of kernels per class, the fifth class' first kernel started when another class'
    // kernel ended), so the classes' work is put into few large launches instead of many small ones.  The launch's work list = the lists
    // of long enumerations of its classes, most endpoints first, then their other lists.
    if (*P.err != 0) return;
    static_assert(W == kNarrow || W == 64 * kCandWords, "one instantiation per half of the work list");
    constexpr bool kWide = W != kNarrow;
    __shared__ LeanLds<W> L;
    __shared__ int32_t seg_first[2 * kMaxEp + 1];   // first item of every segment of the launch's work list (segment = one list of one class)
    // every term table in ONE array of dynamic LDS, so that a word of the scoring program addresses its term by an index (a pointer
    // that may point at one of several LDS arrays or at global memory is a generic pointer: flat loads, a wait on both counters per term):
    // [0, kMaxEp W) root terms by (endpoint, staged position), [kMaxEp W, 2 kMaxEp W) closing terms, from kTabPairs on `pool` doubles of
    // pair tables
    HIP_DYNAMIC_SHARED(double, tabs)
    constexpr int kTabClose = kMaxEp * W, kTabPairs = 2 * kMaxEp * W;
    const int t = threadIdx.x, nt = blockDim.x, lane = t & 63;
    const int ncls = E_hi - E_lo + 1;
    if (t == 0) {
        int acc = 0;
        for (int k = 0; k < ncls; k++) { const int Ec = E_hi - k; seg_first[k] = acc; acc += part == 3 ? P.defer_count[Ec] : (part == 1 ? P.redo_count[Ec] : P.heavy_big_count[Ec]); }
        for (int k = 0; k < ncls; k++) { const int Ec = E_hi - k; seg_first[ncls + k] = acc; acc += part == 0 ? P.heavy_in_count[kWide ? kMaxEp + 1 + Ec : Ec] : 0; }
        seg_first[2 * ncls] = acc;
    }
    wave_sync();
    const int count = seg_first[2 * ncls];
    int32_t* next_counter = enum_cursor(P, part, kWide, E_lo);
    const int nstatic = (int)gridDim.x * kWorkChunk;
    if ((int)blockIdx.x >= count) return;
    int chunk_pos = 0, chunk_end = 0;
    int front_slot = -1;   // this wavefront's pair of tuple-list buffers: -1 not claimed yet, -2 none left
    int pair_slot = -1;    // ... slot of P.pair_pool (pair tables beyond the LDS pool)
    bool first_chunk = true;
#ifdef TW_PROFILE
    const int E = E_hi;   // (the timers of a -DTW_PROFILE build are switched on by the launch's largest class; shadowed by the item's own class below)
#endif
    TW_PROF_DECL();
    while (true) {
        if (chunk_pos == chunk_end) {
            if (first_chunk) { chunk_pos = (int)blockIdx.x * kWorkChunk; first_chunk = false; }
            else {
                if (t == 0) chunk_pos = nstatic + atomicAdd(next_counter, kWorkChunk);
                chunk_pos = __shfl(chunk_pos, 0);
            }
            const int limit = chunk_pos < nstatic ? nstatic : count;
            chunk_end = chunk_pos + kWorkChunk < limit ? chunk_pos + kWorkChunk : limit;
            if (chunk_pos >= count && chunk_pos >= nstatic) { TW_PROF_FLUSH(); break; }
        }
        int item = chunk_pos++;
        if (item < nstatic) item = (item % kWorkChunk) * (int)gridDim.x + item / kWorkChunk;   // the long spans at the front: one to a wavefront
        if (item >= count) continue;
        int sg = 0;   // the segment of the item: which list of which class
        while (sg + 1 < 2 * ncls && seg_first[sg + 1] <= item) sg++;
        const bool from_big = sg < ncls;
        const int E = __builtin_amdgcn_readfirstlane(E_hi - (from_big ? sg : sg - ncls));
        const int at = item - seg_first[sg];
        const int pos = from_big ? P.heavy_big_off[E] + (part == 3 ? P.heavy_big_count[E] : 0) + at : (kWide ? P.heavy_in_off[E + 1] - 1 - at : P.heavy_in_off[E] + at);
        const int unit = __builtin_amdgcn_readfirstlane((from_big ? P.heavy_big_unit : P.heavy_in_unit)[pos]);
        const int i_raw = __builtin_amdgcn_readfirstlane((from_big ? P.heavy_big_idx : P.heavy_in_idx)[pos]);
        const int i = i_raw & ~kIdxReplayFlag;
        const int part_info = from_big ? __builtin_amdgcn_readfirstlane(P.heavy_big_part[pos]) : 1;
        if (from_big && (((part_info >> 24) & 1) != 0) != kWide) continue;   // the other instantiation's
        const int nparts = part_info & 255, part_no = (part_info >> 8) & 0xffff;
        const bool part_log = nparts > 1 && (part_info & kPartLogFlag) != 0;
        const bool replay_first = nparts == 1 && (from_big ? (part_info & kReplayFlag) != 0 : (i_raw & kIdxReplayFlag) != 0);
        const bool list_part = nparts > 1 && (part_info & kListSplitFlag) != 0;
        const int part_slot = from_big ? __builtin_amdgcn_readfirstlane(P.heavy_big_slot[pos]) : 0;
        const UnitDev& U = P.units[unit];
        // hands the item to k_enumerate_heavy (part = 4) as it is
        auto fallback = [&]() {
            if (t == 0) {
                const int q = P.heavy_big_off[E] + atomicAdd(&P.fb_count[E], 1);
                atomicAdd(&P.fb_total[E], 1);
                P.fb_unit[q] = unit; P.fb_idx[q] = i;
                P.fb_part[q] = from_big ? part_info : (1 | (kWide ? 1 << 24 : 0) | (replay_first ? kReplayFlag : 0));
                P.fb_slot[q] = part_slot;
            }
        };
        if (nparts > 1 && !list_part) {   // a part cut by the first endpoint's candidate (the tile kernel does not cut that way in this kernel's classes):
            if (t == 0) { P.part_n[part_slot] = 256; P.part_leaves[part_slot] = 0; if (part_log) P.part_logn[part_slot] = 0; }   // reported as failed, the span is listed again
            continue;
        }
        if (U.skip) { fallback(); continue; }   // (skip-mode batches are not launched here at all)
        TW_ITEM_BEGIN();
        const int64_t in_start = P.in_start[U.in_off + i], in_end = P.in_end[U.in_off + i];
        Scorer S;
        S.pass = pass;
        S.gp = P.gparam + (U.gp_off + (int64_t)(i / P.batch_size) * U.nslot) * 4;
        S.mix_n = P.mix_n + U.slot_off;
        S.mix_c = P.mix_c + (int64_t)U.slot_off * kMaxComp * 4;
        // ---- cut-offs (computed by k_enumerate_tile in pass 1), the DAG's predecessor masks
        for (int k = t; k < kMaxEp * kCandWords; k += nt) (&L.sbits[0][0])[k] = 0;
        for (int e = t; e < E; e += nt) {
            const int l0 = P.c_lo[ie_index(U, e, i)], h0 = P.c_hi[ie_index(U, e, i)];
            L.lo[e] = l0; L.wd[e] = h0 - l0 + 1; L.pm[e] = U.pred_mask[e];
        }
        wave_sync();
        // ---- stage the candidates that can occur in a tuple, order kept: all windows' loads in flight together (raw into LDS),
        // then compacted in place, endpoint after endpoint, by ballot prefix sums
        for (int s0 = 0; s0 < E * W; s0 += 4 * nt) {
            int64_t st[4], en[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int s = s0 + k * nt + t, e = s / W, r = s % W;
                st[k] = 0; en[k] = 0;
                if (s < E * W && r < L.wd[e]) { st[k] = P.out_start[U.ep_off[e] + L.lo[e] + r]; en[k] = P.out_end[U.ep_off[e] + L.lo[e] + r]; }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int s = s0 + k * nt + t, e = s / W, r = s % W;
                if (s < E * W && r < L.wd[e]) { L.ls[e][r] = st[k]; L.le[e][r] = en[k]; }
            }
        }
        wave_sync();
        bool none = false;
        for (int e = 0; e < E; e++) {
            const int w = L.wd[e];
            const uint64_t* gone = mode == 1 ? P.gone + ie_index(U, e, i) * kCandWords : nullptr;
            int c = 0;
            for (int r0 = 0; r0 < w; r0 += nt) {
                const int r = r0 + t;
                int64_t st = 0, e2 = 0;
                bool inside = false;
                if (r < w) {
                    st = L.ls[e][r]; e2 = L.le[e][r];
                    inside = !(in_start > st || e2 > in_end);
                    if (inside && gone != nullptr) inside = !((gone[r >> 6] >> (r & 63)) & 1ull);
                }
                const unsigned long long m = __ballot(inside);
                wave_sync();   // every lane has read its raw entry before the compacted ones are written over them
                if (inside) {
                    const int q = c + __popcll(m & ((1ull << lane) - 1ull));
                    L.ls[e][q] = st; L.le[e][q] = e2; L.lr[e][q] = (uint8_t)r;
                }
                c += __popcll(m);
                wave_sync();
            }
            if (t == 0) { L.cn[e] = c; L.magic[e] = c > 0 ? (uint32_t)((0x100000000ull + (unsigned)c - 1ull) / (unsigned)c) : 0u; }
            none |= c == 0;
        }
        wave_sync();
        TW_PHASE(0);
        long long leaves = 0;
        int nout = 0, nlog = 0;
        bool part_failed = false, deferred = false, part_ambiguous = false, fell_back = false;
        bool exact_replay = false;
        double ts[kTopK];
        int tslot[kTopK], nk = 0;
        LeanHeap<W> hp;
        hp.hs = L.hs; hp.hx = L.hx; hp.n = 0; hp.E = E; hp.ls = L.ls;
        if (!none) {
            // ---- root / closing terms: one (candidate span, root | closing) term per lane, all endpoints at once
            int wsum = 0;
            uint32_t any_order = 0, roots = 0;
            long long grid = 1;
            for (int e = 0; e < E; e++) { wsum += L.cn[e]; any_order |= L.pm[e]; roots |= U.npred[e] == 0 ? 1u << e : 0u; grid = grid < (1ll << 40) ? grid * L.cn[e] : grid; }
            // ---- the plan: per primary in-edge, in scoring order, where its pair table goes (the LDS pool while it lasts, then the
            // wavefront's slot of the global pool), and the scoring program
            int used = 0, gused = 0, n_ops = 0, n_edge = 0, pairs = 0;
            bool tables = true;
            double* gpair = pair_slot >= 0 ? P.pair_pool + (size_t)pair_slot * kPairSpill : nullptr;
            for (int e = 0; e < E; e++) {
                const int np = U.npred[e], ce = L.cn[e];
                for (int j = 0; j < np; j++) {
                    if (!U.pred_prim[e][j]) continue;
                    const int p = U.pred_list[e][j];
                    const int need = L.cn[p] * ce;
                    const bool in_lds = used + need <= pool && kTabPairs + used + need <= 0xffff;
                    if (!in_lds) {   // beyond the LDS pool: the wavefront's slot of the global pool (claimed at the first such table)
                        if (pair_slot == -1) {
                            if (t == 0) pair_slot = pool_acquire(P.pair_busy, kPairSlots, (unsigned)blockIdx.x * 48271u + (unsigned)(E * 2 + (kWide ? 1 : 0)) * 7919u + (unsigned)part * 104729u + 11u);
                            pair_slot = __shfl(pair_slot, 0);
                            if (pair_slot < 0) pair_slot = -2;
                            gpair = pair_slot >= 0 ? P.pair_pool + (size_t)pair_slot * kPairSpill : nullptr;
                        }
                        if (gpair == nullptr || gused + need > kPairSpill) { tables = false; continue; }
                    }
                    if (t == 0) {
                        L.ops[n_ops] = in_lds ? lean_op(0, e, p, ce, kTabPairs + used) : lean_op(3, e, p, ce, gused);
                        L.edge_first[n_edge] = pairs; L.edge_op[n_edge] = (uint8_t)n_ops;
                    }
                    n_ops++; n_edge++;
                    pairs += need;
                    if (in_lds) used += need; else gused += need;
                }
                if (np == 0) { if (t == 0) L.ops[n_ops] = lean_op(0, e, e, 0, e * W); n_ops++; }   // (the root term: a table of one row)
                if (t == 0) L.ops[n_ops] = lean_op(1, e, e, 0, kTabClose + e * W);
                n_ops++;
            }
            if (t == 0) L.edge_first[n_edge] = pairs;
            wave_sync();
            // ---- every term of the item, one per lane: root / closing terms per staged candidate, pair terms per pair of staged
            // candidates of a primary in-edge -- ONE index space, so that an item of a few dozen terms is one batch of the (in pass 2:
            // ~1.5 k instructions) term evaluation, not one batch per table
            for (int q = t; q < 2 * wsum + pairs; q += nt) {
                if (q < 2 * wsum) {
                    int es = 0, r = q >> 1;
                    for (int e = 0; e < E; e++) { if (r < L.cn[e]) { es = e; break; } r -= L.cn[e]; }
                    const int64_t st = L.ls[es][r], e2 = L.le[es][r];
                    if (q & 1) tabs[kTabClose + es * W + r] = score_term(S, slot_close(E, es), e2, in_end);
                    else tabs[es * W + r] = ((roots >> es) & 1u) ? score_term(S, slot_root(E, es), in_start, st) : 0.0;
                    continue;
                }
                const int x = q - 2 * wsum;
                int lo2 = 0, hi2 = n_edge - 1;   // the edge with edge_first[k] <= x < edge_first[k + 1]
                while (lo2 < hi2) { const int mid = (lo2 + hi2 + 1) >> 1; if (L.edge_first[mid] <= x) lo2 = mid; else hi2 = mid - 1; }
                const uint32_t op = L.ops[L.edge_op[lo2]];   // its word of the program: endpoints, row length, where the table lies
                const int e = (int)((op >> 2) & 7u), p = (int)((op >> 5) & 7u), ce = (int)((op >> 8) & 255u), at = (int)(op >> 16);
                const int y = x - L.edge_first[lo2], a2 = y / ce, b2 = y % ce;
                const int64_t pend = L.le[p][a2], st = L.ls[e][b2];
                if (pend <= st) {   // other pairs never occur in a tuple
                    const double v = score_term(S, slot_prim(E, p, e), pend, st);
                    if ((op & 3u) == 0u) tabs[at + y] = v; else gpair[at + y] = v;
                }
            }
            for (int k = n_ops; k < ((n_ops + 3) & ~3); k++) if (t == 0) L.ops[k] = lean_op(2, 0, 0, 0, 0);   // (the program is run four words at a time)
            n_ops = (n_ops + 3) & ~3;
            if (t == 0) L.n_ops = n_ops;
            if (gused > 0) __threadfence_block();   // (tables in global memory: written by all lanes, read by all lanes)
            wave_sync();
            TW_PHASE(1);
            // ---- which route
            const bool can_defer = E >= 3 && E >= P.defer_min_e && part == 0 && mode == 0 && nparts == 1 && any_order != 0 && (!replay_first || P.split_twins == 2);
            const bool as_grid = !list_part && (grid <= P.lean_grid || (any_order == 0 && grid <= 4096));   // (without call-order constraints every grid point is a tuple)
            bool list_all = false;    // pass 2 knows the tuple count of pass 1: few tuples are listed whole and scored
            if (!list_part && !as_grid && pass == 2 && mode == 0) list_all = P.leaves0[U.in_off + i] <= (can_defer ? kDeferTuples : kListScoreMax);
            // not for this kernel: a pair table beyond the pool; a long enumeration without call-order constraints (every grid point a tuple:
            // the pruned walk counts in closed form) or one of pass 2 known to hold more tuples than a list is scored from (pruned walk)
            if (!tables || (!list_part && !as_grid && any_order == 0) ||
                (!list_part && !as_grid && pass == 2 && mode == 0 && !can_defer && !list_all)) {
                if (list_part) part_failed = true; else { fallback(); fell_back = true; }   // (a part never goes there: it reports failure and its span is listed again, whole)
            }
            bool use_front = false;
            int n_front = 0;
            const unsigned long long* front = nullptr;
            if (!fell_back && !part_failed && !as_grid) {
                // ---- the feasible tuples as a list, level by level, depth-first order kept (k_enumerate_heavy, "tuple list")
                if (front_slot == -1) {
                    if (t == 0) front_slot = pool_acquire(P.frontier_busy, kFrontierSlots, ((unsigned)blockIdx.x * 40503u + (unsigned)(E * 2 + (kWide ? 1 : 0)) * 7919u + (unsigned)part * 104729u + 3u));
                    front_slot = __shfl(front_slot, 0);
                    if (front_slot < 0) front_slot = -2;
                }
                if (front_slot < 0) { if (list_part) part_failed = true; else { fallback(); fell_back = true; } }   // (no buffer pair left: a part reports failure -- its span is listed again --, a span goes to k_enumerate_heavy)
                else {
                    unsigned long long* fa = P.frontier + (size_t)front_slot * 2 * kFrontierCap;
                    const int cap = kFrontierCap;
                    unsigned long long* fb = fa + cap;
                    const bool count_only = !list_all && !list_part;
                    int nprev = L.cn[0], part_lvl = 0, minlo = 0x7fffffff;
                    long long leaves_counted = 0;
                    bool list_scored = false, overflow = false, defer_by_tuples = false;
                    int defer_np = 0, defer_slot = 0, defer_at = 0, defer_level = 0;
                    auto defer_reserve = [&](int np, int n) -> bool {
                        int got = -1, at = 0;
                        if (t == 0 && np >= 2) {
                            // (the arena first -- what it hands out is never given back --, the class' budget by bump_reserve, which takes nothing back either)
                            bool ok = atomicAdd(&P.part_used[E], 0) + np <= (P.part_off[E + 1] - P.part_off[E]) / 2 && atomicAdd(P.defer_used, 0) < P.defer_cap;
                            if (ok) {
                                at = atomicAdd(P.defer_used, n);
                                if (at >= 0 && (long long)at + n <= (long long)P.defer_cap) {
                                    const int old = bump_reserve(&P.part_used[E], np, (P.part_off[E + 1] - P.part_off[E]) / 2);   // (the span's own entry stays on the list: every part is an extra one)
                                    if (old >= 0) got = P.part_off[E] + 2 * old;
                                }
                            }
                            if (got < 0) atomicAdd(&P.defer_refused[E], 1);   // (refused -- a smaller request may still fit --, and counted)
                        }
                        defer_slot = __shfl(got, 0); defer_at = __shfl(at, 0); defer_np = np;
                        return defer_slot >= 0;
                    };
                    if (list_part) {   // the stretch of the span's listed prefixes of level part_lvl: the levels below are listed into the own buffers
                        const int p_lo = P.part_lo[part_slot], p_hi = P.part_hi[part_slot];
                        part_lvl = P.part_lvl[part_slot];
                        fb = fa; fa = P.defer_list + p_lo; nprev = p_hi - p_lo;
                    } else
                        for (int c = t; c < nprev; c += nt) fa[c] = (unsigned long long)c;
                    use_front = true;
                    __threadfence_block();
                    wave_sync();
                    for (int d = 1; d < E; d++) {
                        if (!use_front || deferred) break;
                        if (list_part && d <= part_lvl) continue;
                        const int cd = L.cn[d];
                        const uint32_t pmd = L.pm[d] & ((1u << d) - 1u);
                        const bool count_level = count_only && d == E - 1;
                        int nnext = 0;
                        // the last level of a counted enumeration: counted first (rep 0) and -- only where the count is small -- written
                        // after all (rep 1); every other level is written
                        for (int rep = count_level ? 0 : 1; rep < 2; rep++) {
                            const bool last_count = rep == 0;
                            if (rep == 1 && count_level) {
                                if (can_defer && leaves_counted > kDeferTuples) {
                                    const long long want = (leaves_counted + kSplitTuples - 1) / kSplitTuples;
                                    if (defer_reserve((int)(want > kMaxParts ? kMaxParts : want), nprev)) { deferred = true; defer_by_tuples = true; defer_level = E - 2; break; }
                                }
                                if (leaves_counted > kListScoreMax || leaves_counted > cap) { overflow = true; break; }
                                list_scored = true;
                            }
                            nnext = 0;
                            unsigned long long ent_ahead = t < nprev ? fa[t] : 0ull;   // (the list lives in global memory: one batch ahead of its use)
                            for (int base = 0; base < nprev; base += nt) {
                                const int f = base + t;
                                const bool valid = f < nprev;
                                const unsigned long long ent = ent_ahead;
                                ent_ahead = f + nt < nprev ? fa[f + nt] : 0ull;
                                int64_t tmin = INT64_MIN;   // the latest end among the predecessors' spans
                                for (uint32_t rest = pmd; rest != 0u; rest &= rest - 1u) {
                                    const int q = __ffs((int)rest) - 1;
                                    const int64_t en = L.le[q][lean_pos(ent, q)];
                                    tmin = en > tmin ? en : tmin;
                                }
                                int lo2 = 0, hi2 = cd;
                                while (lo2 < hi2) { const int mid = (lo2 + hi2) >> 1; if (L.ls[d][mid] < tmin) lo2 = mid + 1; else hi2 = mid; }
                                const int cnt = valid ? cd - lo2 : 0;
                                int incl = cnt;
                                for (int off = 1; off < 64; off <<= 1) { const int v = __shfl(incl, lane >= off ? lane - off : lane); if (lane >= off) incl += v; }
                                const int total = __shfl(incl, nt - 1 < 63 ? nt - 1 : 63);
                                if (last_count) {   // the tuples below every prefix: counted, their spans marked (not written, not scored)
                                    leaves_counted += total;
                                    if (cnt > 0 && pass == 1 && mode == 0) {
                                        for (int q = 0; q < E - 1; q++) {
                                            const int r = L.lr[q][lean_pos(ent, q)];
                                            const unsigned long long bit = 1ull << (r & 63);
                                            if (!(L.sbits[q][r >> 6] & bit)) atomicOr(&L.sbits[q][r >> 6], bit);
                                        }
                                        minlo = lo2 < minlo ? lo2 : minlo;
                                    }
                                    continue;
                                }
                                const int start = nnext + incl - cnt;
                                if (nnext + total <= cap)
                                    for (int c = lo2; c < lo2 + cnt; c++) fb[start + (c - lo2)] = ent | ((unsigned long long)c << (8 * d));
                                nnext += total;
                                if (nnext > cap) break;   // uniform
                            }
                        }
                        if (deferred || overflow) break;
                        if (nnext > cap) { use_front = false; overflow = true; break; }
                        if (!(count_level && !list_scored)) {   // (a level that was only counted leaves the list of its prefixes where it is)
                            unsigned long long* sw = fa; fa = fb; fb = sw;
                            if (list_part && d == part_lvl + 1) fb = fa + cap;   // (a list part's first level came from the arena: on between its own two buffers)
                            nprev = nnext;
                            // a long list before the last level: the span is cut here, its parts list the rest (kDeferPrefixes)
                            if (can_defer && d < E - 1 && nprev >= kDeferPrefixes) {
                                const int want = nprev / kDeferPrefixGrain;
                                if (defer_reserve(want > kMaxParts ? kMaxParts : want, nprev)) { deferred = true; defer_level = d; }
                            }
                        }
                        __threadfence_block();
                        wave_sync();
                    }
                    front = fa;
                    n_front = nprev;
                    if (deferred) {
                        // The list parts (k_enumerate_heavy): prefix f begins at tuple s_f (running sum of the children counts) and belongs to
                        // part floor(s_f np / total); the list is copied to the arena on the way
                        const int cd = L.cn[E - 1];
                        const uint32_t pml = L.pm[E - 1] & ((1u << (E - 1)) - 1u);
                        for (int q = t; q <= kMaxParts; q += nt) L.defer_lo[q] = defer_by_tuples ? nprev : (int)((long long)q * nprev / defer_np);
                        wave_sync();
                        long long running = 0;
                        int carry_pid = -1;
                        if (!defer_by_tuples)
                            for (int f = t; f < nprev; f += nt) P.defer_list[(int64_t)defer_at + f] = fa[f];
                        for (int base = 0; defer_by_tuples && base < nprev; base += nt) {
                            const int f = base + t;
                            const bool valid = f < nprev;
                            const unsigned long long ent = valid ? fa[f] : 0ull;
                            int64_t tmin = INT64_MIN;
                            for (uint32_t rest = pml; rest != 0u; rest &= rest - 1u) {
                                const int q = __ffs((int)rest) - 1;
                                const int64_t en = L.le[q][lean_pos(ent, q)];
                                tmin = en > tmin ? en : tmin;
                            }
                            int lo2 = 0, hi2 = cd;
                            while (lo2 < hi2) { const int mid = (lo2 + hi2) >> 1; if (L.ls[E - 1][mid] < tmin) lo2 = mid + 1; else hi2 = mid; }
                            const int cnt = valid ? cd - lo2 : 0;
                            int incl = cnt;
                            for (int off = 1; off < 64; off <<= 1) { const int v = __shfl(incl, lane >= off ? lane - off : lane); if (lane >= off) incl += v; }
                            const long long s0 = running + incl - cnt;
                            int pid = (int)(s0 * defer_np / leaves_counted);
                            pid = pid > defer_np - 1 ? defer_np - 1 : pid;
                            const int last_valid = nprev - base - 1 < 63 ? nprev - base - 1 : 63;
                            if (!valid) pid = __shfl(pid, last_valid);
                            int prev_pid = __shfl(pid, lane > 0 ? lane - 1 : 0);
                            if (lane == 0) prev_pid = carry_pid;
                            if (valid) {
                                for (int q = prev_pid + 1; q <= pid; q++) L.defer_lo[q] = f;
                                P.defer_list[(int64_t)defer_at + f] = ent;
                            }
                            carry_pid = __shfl(pid, 63);
                            running += __shfl(incl, 63);
                        }
                        __threadfence();
                        wave_sync();
                        const int log_flag = (replay_first && P.split_twins == 2) ? kPartLogFlag : 0;
                        const int flags = (kWide ? 1 << 24 : 0) | log_flag | kListSplitFlag;
                        int ebase = 0;
                        if (t == 0) {
                            ebase = P.heavy_big_off[E] + P.heavy_big_count[E] + atomicAdd(&P.defer_count[E], defer_np);
                            const int k = P.part_off[E] + atomicAdd(&P.split_count[E], 1);
                            P.split_unit[k] = unit; P.split_idx[k] = i; P.split_slot[k] = defer_slot; P.split_parts[k] = defer_np | flags;
                        }
                        ebase = __shfl(ebase, 0);
                        for (int q = t; q < defer_np; q += nt) {
                            P.part_lo[defer_slot + q] = defer_at + L.defer_lo[q];
                            P.part_hi[defer_slot + q] = defer_at + (q + 1 < defer_np ? L.defer_lo[q + 1] : nprev);
                            P.part_lvl[defer_slot + q] = defer_level;
                            P.heavy_big_unit[ebase + q] = unit; P.heavy_big_idx[ebase + q] = i;
                            P.heavy_big_part[ebase + q] = defer_np | (q << 8) | flags; P.heavy_big_slot[ebase + q] = defer_slot + q;
                        }
                        wave_sync();
                        use_front = false;
                    } else if (overflow || (count_only && !list_scored)) {
                        // too many tuples to score from a list of this wavefront (or a level outgrew its buffers): the pruned walk of
                        // k_enumerate_heavy; a list part without room is enumerated again with its span
                        use_front = false;
                        if (list_part) part_failed = true; else { fallback(); fell_back = true; }
                    }
                    (void)minlo;
                }
            }
#ifdef TW_ENUM_TRACE
            if (t == 0) printf("enumerate_lean: E %d pass %d mode %d span %d part %d/%d: %s, %d listed\n", E, pass, mode, i, part_no, nparts,
                               fell_back ? "handed to k_enumerate_heavy" : deferred ? "counted, deferred to list parts" : part_failed ? "list part without a buffer" : use_front ? "scored from its list" : "grid", n_front);
#endif
            TW_PHASE(2);
            if (!fell_back) {
            // ---- the tuples: a grid point or a listed tuple per lane; feasibility (grid), score by the program, top five
            const int Gtot = use_front ? n_front : ((deferred || part_failed) ? 0 : (int)grid);
            const int n_ops2 = L.n_ops;
            const double* gpair2 = P.pair_pool + (size_t)(pair_slot >= 0 ? pair_slot : 0) * kPairSpill;
            // What the batch loop needs per endpoint and is the same for every lane, in (scalar) registers -- a rolled loop that reads it
            // from LDS pays a round trip per endpoint and batch (measured: 6.7 us a batch, one wavefront in three waiting on LDS at any
            // time); the loops over endpoints below are unrolled to kMaxEp with a uniform guard.  The program: word k in lane k of one
            // register (LaneArr), fetched by v_readlane.
            static_assert(kMaxEp == 8, "the endpoint loops of the batch body are unrolled to eight");
            int cnS[kMaxEp];
            uint32_t pmS[kMaxEp], magicS[kMaxEp];
#pragma unroll
            for (int e = 0; e < kMaxEp; e++) {
                cnS[e] = e < E ? __builtin_amdgcn_readfirstlane(L.cn[e]) : 1;
                pmS[e] = e < E ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(L.pm[e] & ((1u << e) - 1u))) : 0u;
                magicS[e] = e < E ? (uint32_t)__builtin_amdgcn_readfirstlane((int)L.magic[e]) : 0u;
            }
            LaneArr<uint32_t> prog;
            prog.load(n_ops2, [&](int q) { return L.ops[q]; });
            // candidate spans that occur in a feasible tuple (pass 1): every lane collects its own by staged position in registers
            // (narrow windows: 32 positions an endpoint), OR-ed over the wavefront once per item
            uint32_t seen[kMaxEp];
#pragma unroll
            for (int e = 0; e < kMaxEp; e++) seen[e] = 0u;
            const bool mark = pass == 1 && mode == 0;
            // the value one word of the program adds
            auto fetch = [&](uint32_t op, unsigned long long pk) -> double {
                const int at = (int)(op >> 16) + lean_pos(pk, (int)((op >> 5) & 7u)) * (int)((op >> 8) & 255u) + lean_pos(pk, (int)((op >> 2) & 7u));
                if ((op & 3u) == 3u) return gpair2[at];   // (uniform)
                return tabs[at];
            };
            RegHeap RH;   // CPython's heap of the replay, in registers (tw_kernels.h)
            RH.clear(E);
            auto start_of = [&](int e, int pos2) -> int64_t { return L.ls[e][pos2]; };
            for (int attempt = (deferred || part_failed) ? 2 : (part_log || replay_first) ? 1 : 0; attempt < 2; attempt++) {
                exact_replay = attempt == 1;
                bool ambiguous = false;
                RH.n = 0; nk = 0; leaves = 0;
#pragma unroll
                for (int k = 0; k < kTopK; k++) { ts[k] = -dinf(); tslot[k] = k; }
                unsigned long long ent_next = use_front && t < Gtot ? front[t] : 0ull;   // listed tuples: one batch ahead of their use
                for (int base = 0; base < Gtot; base += nt) {
                    const int g = base + t;
                    bool ok = g < Gtot;
                    unsigned long long pk = 0ull;
                    int64_t en8[kMaxEp];
                    if (use_front) {
                        pk = ent_next;
                        ent_next = g + nt < Gtot ? front[g + nt] : 0ull;
#pragma unroll
                        for (int e = 0; e < kMaxEp; e++) en8[e] = e < E ? L.le[e][lean_pos(pk, e)] : INT64_MIN;
                    } else {
                        uint32_t rest = ok ? (uint32_t)g : 0u;   // grid point -> staged positions, last endpoint fastest
#pragma unroll
                        for (int e = kMaxEp - 1; e >= 0; e--) {
                            if (e >= E) continue;
                            const uint32_t c = (uint32_t)cnS[e];
                            const uint32_t q = c == 1 ? rest : __umulhi(rest, magicS[e]);
                            pk |= (unsigned long long)(rest - q * c) << (8 * e);
                            rest = q;
                        }
                        int64_t st8[kMaxEp];
#pragma unroll
                        for (int e = 0; e < kMaxEp; e++) {
                            const int x = lean_pos(pk, e);
                            st8[e] = e < E ? L.ls[e][x] : 0; en8[e] = e < E ? L.le[e][x] : INT64_MIN;
                        }
                        // call order: no predecessor's span ends after this one starts (traceweaver_v3.py:343-347)
#pragma unroll
                        for (int e = 1; e < kMaxEp; e++)
#pragma unroll
                            for (int p = 0; p < e; p++)
                                if ((pmS[e] >> p) & 1u) ok = ok && !(en8[p] > st8[e]);
                    }
                    // ScoreAssignmentAsPerInvocationGraph (traceweaver_v1.py:305-361) from the term tables, the additions in its order.
                    // (Every lane runs the program -- a lane without a tuple on positions that exist, its sum is not used -- so that the
                    // words of the program are uniform; the value of word k + 1 is on its way while word k is added.)
                    double score = 0.0;
                    {
                        int last = 0;
                        int64_t last_end = en8[0];
#pragma unroll
                        for (int e = 1; e < kMaxEp; e++) if (e < E && en8[e] > last_end) { last_end = en8[e]; last = e; }
                        for (int k = 0; k < n_ops2; k += 4) {   // four words at a time: their terms are read together, then added in order
                            uint32_t op[4];
                            double v[4];
#pragma unroll
                            for (int q = 0; q < 4; q++) { op[q] = prog.get(k + q); v[q] = fetch(op[q], pk); }
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                const uint32_t kind = op[q] & 3u;
                                const bool add = kind == 0u || kind == 3u || (kind == 1u && (int)((op[q] >> 2) & 7u) == last);
                                const double s2 = score + v[q];
                                score = add ? s2 : score;
                            }
                        }
                    }
                    if (ok && mark) {
                        if constexpr (!kWide) {
#pragma unroll
                            for (int e = 0; e < kMaxEp; e++) if (e < E) seen[e] |= 1u << lean_pos(pk, e);
                        } else {
                            for (int e = 0; e < E; e++) {
                                const int r = L.lr[e][lean_pos(pk, e)];
                                const unsigned long long bit = 1ull << (r & 63);
                                if (!(L.sbits[e][r >> 6] & bit)) atomicOr(&L.sbits[e][r >> 6], bit);
                            }
                        }
                    }
                    const unsigned long long feasible = __ballot(ok);
                    leaves += __popcll(feasible);
                    if (exact_replay) {
                        // (a tuple strictly below the root of a full heap of five is a no-op: see k_enumerate_heavy)
                        unsigned long long todo = __ballot(ok && !RH.below_root(score));
                        while (todo) {   // in enumeration order, CPython's heappush / heappop replayed by every lane alike (RegHeap)
                            const int j = __ffsll((long long)todo) - 1;
                            todo &= todo - 1;
                            const double sj = lane_value(score, j);
                            const unsigned long long pj = lane_value(pk, j);
                            RH.push(sj, pj, start_of);
                            if (part_log) {   // (its score was not below the root's when it came: it may enter the heap of the whole enumeration)
                                if (t == 0 && nlog < kPartLogCap) {   // (staged positions for now: turned into window positions when the item is done)
                                    P.part_log_sc[(int64_t)part_slot * kPartLogCap + nlog] = sj;
                                    P.part_log_ix[(int64_t)part_slot * kPartLogCap + nlog] = pj;
                                }
                                nlog++;
                            }
                        }
                    } else {
                        // beats the current fifth entry?  score ties are resolved exactly below, so let them through
                        const bool beats = ok && (nk < kTopK || score >= ts[kTopK - 1]);
                        unsigned long long todo = __ballot(beats);
                        if (__popcll(todo) > 2 * kTopK) {
                            // many candidates at once: only the five best of this batch, and whatever ties with the fifth, can be among the final five
                            double rest = beats ? score : -dinf(), thr = -dinf();
                            for (int r = 0; r < kTopK; r++) {
                                double m = rest;
                                for (int off = 32; off >= 1; off >>= 1) { const double o = __shfl_xor(m, off); m = o > m ? o : m; }
                                thr = m;
                                const unsigned long long at = __ballot(rest == m);
                                if (t == __ffsll((long long)at) - 1) rest = -dinf();
                            }
                            todo = __ballot(beats && score >= thr);
                        }
                        while (todo) {
                            const int j = __ffsll((long long)todo) - 1;
                            todo &= todo - 1;
                            const double sj = __shfl(score, j);
                            const unsigned long long pj = __shfl(pk, j);
                            // order of the candidate against a kept tuple of equal score: +1 greater, -1 smaller, 0 equivalent
                            auto tie_order = [&](unsigned long long a, unsigned long long b) -> int {
                                for (int e = 0; e < E; e++) {
                                    const int ca = lean_pos(a, e), cb = lean_pos(b, e);
                                    if (ca != cb) {
                                        const int64_t sa = L.ls[e][ca], sb = L.ls[e][cb];
                                        return sa > sb ? 1 : (sa < sb ? -1 : 0);
                                    }
                                }
                                return 0;
                            };
                            int tie[kTopK];
#pragma unroll
                            for (int k = 0; k < kTopK; k++) {
                                tie[k] = 0;
                                if (k < nk && sj == ts[k]) {
                                    int sl = 0;
#pragma unroll
                                    for (int q = 0; q < kTopK; q++) if (q == k) sl = tslot[q];
                                    tie[k] = tie_order(pj, L.keep[sl]);
                                    if (tie[k] == 0) ambiguous = true;
                                }
                            }
                            if (nk == kTopK) {
                                if (!(sj > ts[kTopK - 1] || (sj == ts[kTopK - 1] && tie[kTopK - 1] > 0))) continue;   // stays out
                                if (ts[kTopK - 2] == ts[kTopK - 1]) {   // the entry that drops out must be the unique minimum of the kept five
                                    int sa = 0, sb = 0;
#pragma unroll
                                    for (int q = 0; q < kTopK; q++) { if (q == kTopK - 2) sa = tslot[q]; if (q == kTopK - 1) sb = tslot[q]; }
                                    if (tie_order(L.keep[sa], L.keep[sb]) == 0) ambiguous = true;
                                }
                            }
                            const int lastpos = nk < kTopK ? nk : kTopK - 1;
                            int slot = 0;   // free slot, or the slot of the entry that drops out
#pragma unroll
                            for (int k = 0; k < kTopK; k++) if (k == lastpos) slot = tslot[k];
                            int pos2 = lastpos;
#pragma unroll
                            for (int k = kTopK - 1; k >= 1; k--) {
                                if (k == pos2 && (sj > ts[k - 1] || (sj == ts[k - 1] && tie[k - 1] > 0))) {
                                    ts[k] = ts[k - 1]; tslot[k] = tslot[k - 1];
                                    pos2 = k - 1;
                                }
                            }
#pragma unroll
                            for (int k = 0; k < kTopK; k++)
                                if (k == pos2) { ts[k] = sj; tslot[k] = slot; }
                            if (nk < kTopK) nk++;
                            wave_sync();   // tie_order of this round has read keep[slot] before it is overwritten
                            if (t == 0) L.keep[slot] = pj;
                            wave_sync();
                        }
                    }
                    wave_sync();
                    if (nparts > 1 && ambiguous) break;   // (uniform) the span will be enumerated as a whole anyway
                }
                if (attempt == 0) {   // the kept tuples themselves must be pairwise ordered
                    if (!(nparts > 1 && ambiguous)) {
#pragma unroll
                        for (int a = 0; a < kTopK; a++)
#pragma unroll
                            for (int b2 = a + 1; b2 < kTopK; b2++)
                                if (b2 < nk && ts[a] == ts[b2]) {
                                    int sa = 0, sb = 0;
#pragma unroll
                                    for (int q = 0; q < kTopK; q++) { if (q == a) sa = tslot[q]; if (q == b2) sb = tslot[q]; }
                                    bool same = true;   // (tie_order inline: the lambda of the batch loop is out of scope here)
                                    int ord = 0;
                                    const unsigned long long ka = L.keep[sa], kb = L.keep[sb];
                                    for (int e = 0; e < E && same; e++) {
                                        const int ca = lean_pos(ka, e), cb = lean_pos(kb, e);
                                        if (ca != cb) { const int64_t xa = L.ls[e][ca], xb = L.ls[e][cb]; ord = xa > xb ? 1 : (xa < xb ? -1 : 0); same = false; }
                                    }
                                    if (ord == 0) ambiguous = true;
                                }
                    }
                    if (nparts > 1) { part_ambiguous = ambiguous; break; }   // a part cannot replay CPython's heap here: the whole span is redone (k_merge_parts)
                    if (!ambiguous) break;
                    wave_sync();
                }
            }
            if (part_log && nlog > 0) {   // the log's tuples as positions in the cut-off windows (the same in every part), an entry per lane
                __threadfence_block();
                wave_sync();
                const int m = nlog < kPartLogCap ? nlog : kPartLogCap;
                for (int k = t; k < m; k += nt) {
                    const unsigned long long pj = P.part_log_ix[(int64_t)part_slot * kPartLogCap + k];
                    unsigned long long wpos = 0ull;
                    for (int e = 0; e < E; e++) wpos |= (unsigned long long)L.lr[e][lean_pos(pj, e)] << (8 * e);
                    P.part_log_ix[(int64_t)part_slot * kPartLogCap + k] = wpos;
                }
            }
            if constexpr (!kWide) {
                if (mark) {   // staged positions seen by any lane -> positions in the cut-off windows
#pragma unroll
                    for (int e = 0; e < kMaxEp; e++) {
                        if (e >= E) continue;
                        uint32_t m = seen[e];
                        for (int off = 32; off >= 1; off >>= 1) m |= __shfl_xor(m, off);
                        for (int c = t; c < cnS[e]; c += nt)
                            if ((m >> c) & 1u) { const int r = L.lr[e][c]; atomicOr(&L.sbits[e][r >> 6], 1ull << (r & 63)); }
                    }
                }
            }
            wave_sync();
            TW_PHASE(3);
            if (exact_replay) {   // the heap as CPython's list, sorted like the reference's final list.sort(reverse=True) (lane 0, in LDS)
                if (t == 0) {
#pragma unroll
                    for (int k = 0; k < kTopK; k++) { L.hs[k] = RH.s[k]; L.hx[k] = RH.x[k]; }
                    hp.n = RH.n;
                    hp.sort_desc();
                }
            }
            wave_sync();
            nout = exact_replay ? RH.n : nk;
            if (!exact_replay) {   // the kept five into the heap arrays (the result stores below read those)
                if (t == 0) {
#pragma unroll
                    for (int k = 0; k < kTopK; k++) {
                        if (k >= nk) continue;
                        int sl = 0;
#pragma unroll
                        for (int q = 0; q < kTopK; q++) if (q == k) sl = tslot[q];
                        L.hs[k] = ts[k]; L.hx[k] = L.keep[sl];
                    }
                }
                wave_sync();
            }
            }
        }   // !none
        if (fell_back) { TW_ITEM_END(0); continue; }
        if (nparts > 1) {   // a part: its top-5 (span indices), tuple count and candidate bitmap go to the scratch slot
            if (t == 0) {
                P.part_n[part_slot] = nout | ((part_ambiguous || part_failed) ? 256 : 0) | (nlog > kPartLogCap ? 512 : 0);
                P.part_leaves[part_slot] = leaves;
                if (part_log) P.part_logn[part_slot] = nlog;
            }
            for (int q = t; q < kTopK * (E + 1); q += nt) {
                const int k = q / (E + 1), f = q % (E + 1);
                if (k >= nout) continue;
                if (f == E) P.part_score[(int64_t)part_slot * kTopK + k] = L.hs[k];
                else P.part_idx[((int64_t)part_slot * kTopK + k) * kMaxEp + f] = L.lo[f] + (int)L.lr[f][lean_pos(L.hx[k], f)];
            }
            for (int q = t; q < E * kCandWords; q += nt) P.part_bits[((int64_t)part_slot * kMaxEp + q / kCandWords) * kCandWords + q % kCandWords] = L.sbits[q / kCandWords][q % kCandWords];
        } else if (!deferred) {   // results leave through all lanes: one (entry, field) per lane; staged positions back to span indices
            const int64_t g = U.in_off + i;
            int32_t* out_n = mode == 1 ? P.tkr_n : P.tk_n;
            int32_t* out_idx = mode == 1 ? P.tkr_idx : P.tk_idx;
            double* out_score = mode == 1 ? P.tkr_score : P.tk_score;
            if (t == 0) {
                out_n[g] = nout; (mode == 1 ? P.leaves_r : P.leaves)[g] = leaves; P.rep[g] = (uint8_t)mode;
                if (mode == 0 && pass == 1) P.leaves0[g] = leaves;
            }
            for (int q = t; q < kTopK * (E + 1); q += nt) {
                const int k = q / (E + 1), f = q % (E + 1);
                if (mode == 0 && k >= nout) continue;   // unused entries keep the -1 / NaN pattern they were given at load time
                if (f == E) out_score[tks_index(U, k, i)] = k < nout ? L.hs[k] : dnan();
                else out_idx[tk_index(U, k, f, i)] = k < nout ? L.lo[f] + (int)L.lr[f][lean_pos(L.hx[k], f)] : -1;
            }
            if (pass == 1 && mode == 0)
                for (int q = t; q < E * kCandWords; q += nt) P.c_bits[ie_index(U, q / kCandWords, i) * kCandWords + q % kCandWords] = L.sbits[q / kCandWords][q % kCandWords];
        }
        wave_sync();
        TW_PHASE(4);
        TW_ITEM_END(leaves);
    }
    if (front_slot >= 0 && t == 0) pool_release(P.frontier_busy, front_slot);
    if (pair_slot >= 0 && t == 0) pool_release(P.pair_busy, pair_slot);
}

}  // namespace tw
