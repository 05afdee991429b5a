// tw_tile.h -- k_enumerate_tile<E>: the candidate-window builder and the short enumerations of a tile of incoming spans,
// by the whole workgroup (included by tw_kernels.h).
//
// Reference functions: FindCutoffs traceweaver_v3.py:182-217, DfsTraverseX / DfsTraverse3 :236-351 (feasibility :328-347),
// ScoreAssignmentAsPerInvocationGraph traceweaver_v1.py:305-361, the size-5 heap of traceweaver_v3.py:304-307.
//
// One workgroup per tile of P.tile_spans consecutive incoming spans of one service.  A thread per span only does what is
// per span (its cut-offs); everything else is spread over the lanes by *item*, so that a span with 60 tuples beside spans
// with two does not hold a wavefront (the per-thread kernel this replaces ran with a quarter of its lanes active and 744
// dependent global loads per wavefront):
//   1. cut-offs per span (pass 1: galloping searches in global memory, kept in c_lo / c_hi; pass 2 reads them back);
//   2. the union of the tile's candidate windows per endpoint -- neighbouring spans' windows overlap and the lists are in
//      start order, so the union is one contiguous slice -- is loaded ONCE, coalesced, into LDS as 32-bit offsets from the
//      tile's first start time; containment (traceweaver_v3.py:328-333) becomes a bit mask per (span, endpoint);
//   3. items = (span, endpoint, contained candidate): the root / closing terms (traceweaver_v1.py:349-357) are evaluated one
//      item per lane into LDS tables -- in pass 2 a term is a mixture log-likelihood of ~1.5 k instructions, and every lane
//      has one to evaluate;
//   4. grid points = (span, tuple of contained candidates), one per lane: call-order feasibility (traceweaver_v3.py:343-347),
//      the tuple's score from the tables in the reference's order of additions (bit-identical), tuple count and the
//      bitmap of candidates that occur in a feasible tuple;
//   5. the five largest tuples of a span by counting: rank = number of the span's tuples that are greater under Python's
//      order of (score, [spans]) -- score, then start_mus of the first differing span.  That order is a strict partial
//      order (two tuples whose first differing spans start together are incomparable), and CPython's heap returns exactly
//      the tuples of rank < 5, in rank order, whenever none of those is incomparable with another tuple of equal score:
//      such a tuple is comparable with everything, so whenever it is the root of the six-entry heap it is its true minimum,
//      has five greater tuples and rank >= 5 -- a tuple of rank < 5 is never popped.  A span where a tuple of rank < 5 has
//      an incomparable partner goes to the wavefront kernel, which replays CPython's heap push by push.
// Spans whose enumeration is longer than kTileMax tuples, whose windows are wider than 32 spans or reach beyond the staged
// slice, or that are too long for 32-bit offsets go to the work lists of k_enumerate_heavy, as before.
#pragma once

namespace tw {

#ifndef TW_TILE_MAX
#define TW_TILE_MAX 256
#endif
constexpr int kTileMax = TW_TILE_MAX;   // largest enumeration (product of the contained candidates) the tile kernel takes itself
#ifndef TW_TILE_REPLAY_MAX
#define TW_TILE_REPLAY_MAX 256
#endif
constexpr int kTileReplayMax = TW_TILE_REPLAY_MAX;   // most feasible tuples of a span whose heap its own thread replays (phase 5b); the others go to the wavefront kernel
// items of a span before those of endpoint e (contained candidates per endpoint, 8 bits each, in pk)
__device__ __forceinline__ int pos_base(unsigned long long pk, int e) {
    int acc = 0;
    for (int f = 0; f < e; f++) acc += (int)((pk >> (8 * f)) & 255ull);
    return acc;
}
#ifndef TW_TILE_ITEMS
#define TW_TILE_ITEMS 768
#define TW_TILE_GRID 1024
#endif
// SPANS = incoming spans a workgroup serves: kTile, or kTile / 8 for the sub-tile launches of a class of few tiles (the deep call
// graphs of a small batch) -- with tables sized for those 16 spans: 15 KB of LDS instead of 48 KB.  (The sub-tile workgroups of one
// such class used to ask for all the LDS of the GPU, 912 x 43 KB; workgroups that need a third of a CU's LDS at once are not placed
// while smaller ones -- the wavefront kernels of the other classes -- keep taking what becomes free: the classes' tile kernels ran one
// after the other.)
constexpr int kSubTileSpans = kTile / 8;
template <int E, int SPANS = kTile>
struct TileCfg {
#ifdef TW_TILE_SMALL   // host-emulation build of the tests: tiny tables, so that segments and slice overflows occur
    static constexpr int kSlice = 24, kItems = kMaxEp * 32, kGrid = kTileMax > 64 ? kTileMax : 64;
#else
    static constexpr int kSlice = SPANS < kTile ? 64 : (E <= 4 ? 192 : 160);   // outgoing spans staged per endpoint
    static constexpr int kItems = SPANS < kTile ? kMaxEp * 32 : TW_TILE_ITEMS;   // (span, endpoint, candidate) items per segment of the tile
    static constexpr int kGrid = SPANS < kTile ? (kTileMax > 512 ? kTileMax : 512) : TW_TILE_GRID;   // tuple slots per segment (a span's slots are padded to a multiple of four)
#endif
    static_assert(kItems >= kMaxEp * 32 && kGrid >= ((kTileMax + 3) & ~3) && kGrid % 4 == 0, "a single span must fit a segment");
};

// rest / c for 0 <= rest <= 4096, 1 <= c <= 64: (rest + 0.5) / c is at least 0.5 / c away from an integer, far more than the
// rounding of the reciprocal and the product -- exact on the GPU (v_rcp_f32) and in the host emulation alike
__device__ __forceinline__ uint32_t div_small(uint32_t rest, uint32_t c) {
#ifdef TW_HOST_EMULATION
    const float rc = 1.0f / (float)c;
#else
    const float rc = __builtin_amdgcn_rcpf((float)c);
#endif
    return (uint32_t)(((float)rest + 0.5f) * rc);
}

// a term from the gap itself (score_term computes x = (double)(t2 - t1) and then exactly this)
__device__ __forceinline__ double score_term_gap(const Scorer& S, int slot, long long gap) {
    const double x = (double)gap;
    if (S.pass == 1) {
        const double* g = S.gp + slot * 4;
        return term_gauss(g[0], g[3], g[2], x);
    }
    return score_term_mix_x(S, slot, x);
}

// Optional phase timers (build with -DTW_PROFILE_TILE; read back through tw_debug_profile): 10 ns ticks of thread 0 of every
// workgroup of the class TW_PROFILE_TILE_E (4) per phase -- [0] cut-offs, [1] slices, [2] masks / lists / prefix sums, [3] terms, [4] tuples,
// [5] ranks, [6] results, [7] workgroup lifetime, [8] workgroups, [9] segments, [10] items, [11] tuples.  Compiled out by default.
#ifdef TW_PROFILE_TILE
#ifndef TW_PROFILE_TILE_E
#define TW_PROFILE_TILE_E 4   // the endpoint-count class that is timed
#endif
#define TW_TILE_T0() long long _tp = wall_clock64(); const long long _tb = _tp; long long _ta[7] = {0, 0, 0, 0, 0, 0, 0}
#define TW_TILE_TICK(k) do { const long long _n = wall_clock64(); _ta[k] += _n - _tp; _tp = _n; } while (0)
#define TW_TILE_COUNT(k, v) do { if (E == TW_PROFILE_TILE_E && threadIdx.x == 0) atomicAdd((unsigned long long*)&P.prof[k], (unsigned long long)(v)); } while (0)
#define TW_TILE_FLUSH() do { if (E == TW_PROFILE_TILE_E && threadIdx.x == 0) { for (int _k = 0; _k < 7; _k++) atomicAdd((unsigned long long*)&P.prof[_k], (unsigned long long)_ta[_k]); \
    atomicAdd((unsigned long long*)&P.prof[7], (unsigned long long)(wall_clock64() - _tb)); atomicAdd((unsigned long long*)&P.prof[8], 1ull); } } while (0)
#else
#define TW_TILE_T0() do {} while (0)
#define TW_TILE_TICK(k) do {} while (0)
#define TW_TILE_COUNT(k, v) do {} while (0)
#define TW_TILE_FLUSH() do {} while (0)
#endif

// (phase 5b's heap in registers took the kernel from 64 to 135 registers for four endpoints: three workgroups per CU where the LDS lets four
// in, 20.0 -> 20.3 ms on the headline.  Up to four endpoints the compiler is held to four wavefronts per SIMD -- 128 registers)
#ifdef TW_HOST_EMULATION
#define TW_TILE_ATTR
#else
#define TW_TILE_ATTR __attribute__((amdgpu_waves_per_eu(E <= 4 ? 4 : 2)))
#endif
template <int E, int SPANS = kTile>
__global__ void __launch_bounds__(4 * kTile) TW_TILE_ATTR k_enumerate_tile(Dev P, int pass, const int32_t* tile_ids, int n_tiles_e, int sub_tiles) {
    if (*P.err != 0) return;  // e.g. NaN parameters (hazard H3): nothing downstream is meaningful
    typedef TileCfg<E, SPANS> C;
    constexpr int SL = C::kSlice;
    // sub_tiles > 1: a class of few tiles (the deep call graphs of a small batch: 57 tiles on 256 CUs, each a millisecond of
    // segments one after the other) is launched with sub_tiles workgroups per tile, each serving a stretch of the tile's spans
    const int vb = xcd_tile(blockIdx.x, n_tiles_e * sub_tiles);
    const int tile = tile_ids[vb / sub_tiles];
    const TileDev T = P.tiles[tile];
    const UnitDev& U = P.units[T.unit];
    const int t = threadIdx.x, nt = blockDim.x;
    const int sub_spans = P.tile_spans / sub_tiles, first = T.first + (vb % sub_tiles) * sub_spans;
    const int ns = min(sub_spans, min(P.tile_spans, U.n_in - T.first) - (vb % sub_tiles) * sub_spans);   // spans of this workgroup; thread t < ns owns span t
    if (ns <= 0) return;
    __shared__ int32_t s_is[SPANS], s_ie[SPANS];          // start / end of the incoming spans as offsets from `base`
    __shared__ int32_t sl_st[E][SL], sl_en[E][SL];        // the staged slice of every endpoint list, same offsets (clamped)
    __shared__ int32_t s_A[E], s_B[E];                    // the slice = positions [A, B) of the endpoint's list
    __shared__ uint16_t s_lo[E][SPANS];                   // first position of the span's window, relative to A
    __shared__ uint32_t s_cm[E][SPANS];                   // bit r: position lo + r holds a contained candidate
    __shared__ uint32_t s_bits[E][SPANS];                 // ... that occurs in a feasible tuple
    __shared__ int32_t s_item0[SPANS + 1], s_grid0[SPANS + 1], s_leaves[SPANS];
    __shared__ uint8_t s_amb[SPANS];
    __shared__ unsigned long long s_cn[SPANS];            // contained candidates per endpoint, 8 bits each
    __shared__ int32_t s_wtot[4 * kTile / 64][2];
    __shared__ double t_root[C::kItems], t_close[C::kItems];
    __shared__ uint16_t it_idx[C::kItems];                // slice position of the item's candidate
    __shared__ double g_sc[C::kGrid];
    __shared__ uint8_t g_span[C::kGrid], g_rank[C::kGrid];   // the tuple's span (tile-local) / its rank; an infeasible grid point has score NaN
    __shared__ int32_t s_segend, s_anyamb;                 // ... any span of the segment whose top five the ranks do not decide (phase 5b)
    const bool live = t < ns;
    const int i = first + (live ? t : 0);
    const int64_t base = P.in_start[U.in_off + first];
    const int64_t in_start = P.in_start[U.in_off + i], in_end = P.in_end[U.in_off + i];
    const int64_t* os[E];
    const int64_t* oe[E];
#pragma unroll
    for (int e = 0; e < E; e++) { os[e] = P.out_start + U.ep_off[e]; oe[e] = P.out_end + U.ep_off[e]; }
    for (int x = t; x < E; x += nt) { s_A[x] = 0x7fffffff; s_B[x] = 0; }
    // the unit's call-order DAG in (scalar) registers: predecessor masks, predecessor counts, and per endpoint the predecessor
    // list in in_edges() order packed 4 bits each (index | primary << 3)
    uint32_t dag_pm[E], dag_np[E], dag_pl[E];
    uint32_t ordered = 0;   // any call-order constraint at all?  (without: every grid point is a tuple)
#pragma unroll
    for (int e = 0; e < E; e++) {
        dag_pm[e] = U.pred_mask[e];
        dag_np[e] = U.npred[e];
        uint32_t pk = 0;
#pragma unroll
        for (int j = 0; j < E; j++) pk |= (j < (int)dag_np[e] ? ((uint32_t)U.pred_list[e][j] | ((uint32_t)U.pred_prim[e][j] << 3)) : 0u) << (4 * j);
        dag_pl[e] = pk;
        ordered |= dag_pm[e];
    }
    TW_TILE_T0();
    // ---- 1. cut-offs ------------------------------------------------------------------------------------------------
    int32_t lo[E], hi[E];
    if (pass == 1) {   // FindCutoffs (traceweaver_v3.py:182-217), reverse topological order
#pragma unroll
        for (int e = E - 1; e >= 0; e--) {
            const int n = (int)(U.ep_off[e + 1] - U.ep_off[e]);
            int64_t tmax = in_end;
#pragma unroll
            for (int f = e + 1; f < E; f++) {
                if (!((U.succ_mask[e] >> f) & 1)) continue;
                const int nf = (int)(U.ep_off[f + 1] - U.ep_off[f]);
                const int anchor = hi[f] >= 0 ? hi[f] : nf - 1;  // Python's [-1] wrap (hazard H10)
                const int64_t st = os[f][anchor];
                if (st < tmax) tmax = st;
            }
            if (U.skip) {   // lists as handed over, possibly out of order: exactly Python's bisect_left / bisect_right
                lo[e] = lower_bound_i64(os[e], n, in_start);
                hi[e] = upper_bound_i64(os[e], n, tmax) - 1;
            } else {
                lo[e] = bound_near<false>(os[e], n, in_start, i);
                hi[e] = bound_near<true>(os[e], n, tmax, lo[e]) - 1;
            }
        }
        if (live) {
#pragma unroll
            for (int e = 0; e < E; e++) { P.c_lo[ie_index(U, e, i)] = lo[e]; P.c_hi[ie_index(U, e, i)] = hi[e]; }
        }
    } else {
#pragma unroll
        for (int e = 0; e < E; e++) { lo[e] = P.c_lo[ie_index(U, e, i)]; hi[e] = P.c_hi[ie_index(U, e, i)]; }
    }
    bool wide = false, empty = false, narrow = true;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int w = hi[e] - lo[e] + 1;
        wide |= (w > 64 * kCandWords);
        narrow &= (w <= kNarrow);
        empty |= (w <= 0);
    }
    if (live && wide) { raise_err(P, TW_ERR_WINDOW_WIDTH); }
    group_sync();
    TW_TILE_TICK(0);
    // ---- 2. the tile's slice of every endpoint list ------------------------------------------------------------------
    {   // (one pair of LDS atomics per wavefront and endpoint, not per lane: same-address atomics serialise)
        const bool in = live && !empty && !wide;
#pragma unroll
        for (int e = 0; e < E; e++) {
            int a = in ? lo[e] : 0x7fffffff, b = in ? hi[e] + 1 : 0;
            for (int off = 32; off >= 1; off >>= 1) { const int a2 = __shfl_xor(a, off), b2 = __shfl_xor(b, off); a = a2 < a ? a2 : a; b = b2 > b ? b2 : b; }
            if ((t & 63) == 0) { atomicMin(&s_A[e], a); atomicMax(&s_B[e], b); }
        }
    }
    group_sync();
    int32_t cnt[E];
#pragma unroll
    for (int e = 0; e < E; e++) { const int c = s_B[e] - s_A[e]; cnt[e] = c < 0 ? 0 : (c > SL ? SL : c); }
    // (a class whose every span goes to the wavefront kernel -- TW_TILE_MAX_DEEP = 0 -- only classifies here: no slice)
    if (E >= P.lean_min_e && P.tile_max_deep == 0) {
#pragma unroll
        for (int e = 0; e < E; e++) cnt[e] = 0;
    }
    auto clamp32 = [](long long v) -> int32_t { return (int32_t)(v > 0x7fffffffll ? 0x7fffffffll : (v < -0x7fffffffll ? -0x7fffffffll : v)); };
    for (int x0 = 0; x0 < SL; x0 += nt) {   // (one round with the production sizes: the loads of all endpoints are in flight together)
        const int x = x0 + t;
        int64_t a[E], b[E];
#pragma unroll
        for (int e = 0; e < E; e++) { a[e] = 0; b[e] = 0; if (x < cnt[e]) { a[e] = os[e][s_A[e] + x]; b[e] = oe[e][s_A[e] + x]; } }
#pragma unroll
        for (int e = 0; e < E; e++) if (x < cnt[e]) { sl_st[e][x] = clamp32(a[e] - base); sl_en[e][x] = clamp32(b[e] - base); }
    }
    group_sync();
    TW_TILE_TICK(1);
    // ---- containment masks, the size of the span's enumeration, who enumerates it ------------------------------------
    const long long iso = in_start - base, ieo = in_end - base;
    // (offsets beyond 32 bits: not for this kernel; the clamped ends of the slice compare correctly against anything inside)
    const bool far = iso < 0 || ieo >= 0x7fffffffll || in_end < in_start || in_end - in_start >= (1ll << 31);
    uint32_t cm[E];
    bool fits = narrow && !far && !empty;
    long long prod = empty ? 0 : 1;
#pragma unroll
    for (int e = 0; e < E; e++) {
        cm[e] = 0;
        const int rel = lo[e] - s_A[e], w = hi[e] - lo[e] + 1;
        if (fits && (rel < 0 || rel + w > cnt[e])) fits = false;   // the window reaches beyond the staged slice
        if (fits) {
            for (int r = 0; r < w; r++)
                if (sl_st[e][rel + r] >= (int32_t)iso && sl_en[e][rel + r] <= (int32_t)ieo) cm[e] |= 1u << r;
            prod = prod <= kTileMax ? prod * __popcll((unsigned long long)cm[e]) : prod;
        }
    }
    // (the deep call graphs: every tuple costs this kernel a term per primary call-order edge -- in pass 2 a mixture term of ~1.5 k
    // instructions --, the wavefront kernel tabulates those per pair of candidates: its classes hand over from P.tile_max_deep tuples on)
    const long long tile_max = E >= P.lean_min_e ? (long long)P.tile_max_deep : (long long)kTileMax;
    const bool mine = live && !wide && (empty || (fits && prod <= tile_max));   // this kernel enumerates the span
    {   // the others: listed for the wavefront kernel with what it needs to plan (k_enumerate_heavy)
        long long hprod = 0;
        int first_cands = 0;
        bool twins_any = false;
        if (live && !mine && !wide) {
            // the windows also hold spans that start inside the incoming span but end after it; what the enumeration
            // costs is the product of the *contained* candidates (a third of the raw product on the bench workload)
            hprod = 1;
            bool twins = false;   // two candidates of one endpoint that start together: Python's order of tuples may not decide
#pragma unroll
            for (int e = 0; e < E; e++) {
                int v = 0;
                const int rel = lo[e] - s_A[e], w = hi[e] - lo[e] + 1;
                if (!far && rel >= 0 && rel + w <= cnt[e]) {   // the window lies in the staged slice
                    int32_t prev = -0x7fffffff - 1;
                    for (int r = 0; r < w; r++) {
                        const int32_t st = sl_st[e][rel + r];
                        if (st >= (int32_t)iso && sl_en[e][rel + r] <= (int32_t)ieo) { v++; twins |= st == prev; prev = st; }
                    }
                } else {
                    int64_t prev = INT64_MIN;
                    for (int cx = lo[e]; cx <= hi[e]; cx++) {
                        const int64_t st = os[e][cx];
                        if (st >= in_start && oe[e][cx] <= in_end) { v++; twins |= st == prev; prev = st; }
                    }
                }
                if (hprod <= (1ll << 40)) hprod *= v;
                if (e == 0) first_cands = v;
            }
            // with twins the parts cannot settle the top five among themselves (millisecond-granular traces: nearly always): they
            // replay CPython's heap on their shares and log what entered it (log mode, see heavy_append)
            if (twins && P.split_twins == 0) first_cands = 0;
            twins_any = twins;
            if (E >= P.lean_min_e) first_cands = 0;   // (k_enumerate_lean's classes are never cut by the first endpoint's candidate)
            // a class that defers its long spans cuts them by their listed prefixes once their tuples are counted (kListSplitFlag)
            if (E >= 3 && E >= P.defer_min_e) {
                uint32_t any_order = 0;
#pragma unroll
                for (int e = 0; e < E; e++) any_order |= U.pred_mask[e];
                if (any_order != 0) first_cands = 0;
            }
        }
        heavy_append<E>(P, live && !mine && !wide, narrow, hprod > kBigProduct, T.unit, i, hprod, U.skip ? 0 : first_cands, twins_any);
    }
    const bool some = mine && !empty && prod > 0;   // has tuples to enumerate here
    int nitem = 0;
    unsigned long long cnpack = 0;   // contained candidates per endpoint, 8 bits each
    if (live) {
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int rel = lo[e] - s_A[e], c = some ? __popcll((unsigned long long)cm[e]) : 0;
            s_lo[e][t] = (uint16_t)(some ? rel : 0);
            s_cm[e][t] = some ? cm[e] : 0u;
            s_bits[e][t] = 0u;
            nitem += c;
            cnpack |= (unsigned long long)c << (8 * e);
        }
        s_cn[t] = cnpack;
        s_is[t] = (int32_t)iso; s_ie[t] = (int32_t)ieo;
        s_leaves[t] = (some && !ordered) ? (int)prod : 0; s_amb[t] = 0;
    }
    {   // inclusive prefix sums of the items and of the tuple slots (a span's slots padded to a multiple of four: the rank
        // loop reads four scores a step) over the spans: wavefront scans by shuffles, then the wavefronts' totals through LDS
        int a = live ? nitem : 0, b = (live && some) ? (((int)prod + 3) & ~3) : 0;
        const int lane = t & 63, wv = t >> 6;
        for (int off = 1; off < 64; off <<= 1) {
            const int a2 = __shfl_up(a, off), b2 = __shfl_up(b, off);
            if (lane >= off) { a += a2; b += b2; }
        }
        if (lane == 63 || t == nt - 1) { s_wtot[wv][0] = a; s_wtot[wv][1] = b; }
        group_sync();
        for (int w = 0; w < wv; w++) { a += s_wtot[w][0]; b += s_wtot[w][1]; }
        if (live) { s_item0[t + 1] = a; s_grid0[t + 1] = b; }
        if (t == 0) { s_item0[0] = 0; s_grid0[0] = 0; }
        group_sync();
    }
    TW_TILE_TICK(2);
    Scorer S;
    S.pass = pass;
    S.mix_n = P.mix_n + U.slot_off;
    S.mix_c = P.mix_c + (int64_t)U.slot_off * kMaxComp * 4;
    auto span_of = [&](const int32_t* pre, int first, int last, int x) -> int {   // largest s in [first, last) with pre[s] <= x
        int a = first, b = last - 1;
        while (a < b) { const int mid = (a + b + 1) >> 1; if (pre[mid] <= x) a = mid; else b = mid - 1; }
        return a;
    };
    // grid point -> the tuple's candidates: digits (last endpoint fastest, the reference's order), item numbers, slice positions
    auto decode = [&](int s, int gi, int it_first, int32_t (&kx)[E], int32_t (&ix)[E]) {
        uint32_t rest = (uint32_t)gi;
        const unsigned long long pk = s_cn[s];
        int below[E], acc = s_item0[s] - it_first;
#pragma unroll
        for (int e = 0; e < E; e++) { below[e] = acc; acc += (int)((pk >> (8 * e)) & 255ull); }
#pragma unroll
        for (int e = E - 1; e >= 0; e--) {
            const uint32_t c = (uint32_t)((pk >> (8 * e)) & 255ull);
            const uint32_t q = c == 1 ? rest : div_small(rest, c);
            kx[e] = below[e] + (int32_t)(rest - q * c);
            rest = q;
            ix[e] = it_idx[kx[e]];
        }
    };
    // ---- segments of the tile: as many spans as the tables hold ------------------------------------------------------
    for (int seg = 0; seg < ns;) {
        if (t == 0) { s_segend = ns; s_anyamb = 0; }
        group_sync();
        if (live && t >= seg && (s_item0[t + 1] - s_item0[seg] > C::kItems || s_grid0[t + 1] - s_grid0[seg] > C::kGrid)) atomicMin(&s_segend, t);
        group_sync();
        const int send = s_segend;
        const int it0 = s_item0[seg], nIt = s_item0[send] - it0, g0 = s_grid0[seg], nG = s_grid0[send] - g0;
        TW_TILE_TICK(2);
        TW_TILE_COUNT(9, 1); TW_TILE_COUNT(10, nIt); TW_TILE_COUNT(11, nG);
        // ---- 3. root / closing terms, one (span, endpoint, candidate) per lane ----------------------------------------
        for (int k = t; k < nIt; k += nt) {
            const int s = span_of(s_item0, seg, send, it0 + k);
            int j = it0 + k - s_item0[s], es = 0;
            uint32_t m = 0;
            bool found = false;
#pragma unroll
            for (int e = 0; e < E; e++)
                if (!found) { m = s_cm[e][s]; const int c = __popcll((unsigned long long)m); if (j < c) { es = e; found = true; } else j -= c; }
            const int idx = (int)s_lo[es][s] + nth_set_bit((unsigned long long)m, j);
            int32_t st = 0, en = 0;
            bool is_root = false;
#pragma unroll
            for (int e = 0; e < E; e++) if (e == es) { st = sl_st[e][idx]; en = sl_en[e][idx]; is_root = dag_np[e] == 0; }
            S.gp = P.gparam + (U.gp_off + (int64_t)((first + s) / P.batch_size) * U.nslot) * 4;
            t_root[k] = is_root ? score_term_gap(S, slot_root(E, es), (long long)st - s_is[s]) : 0.0;
            t_close[k] = score_term_gap(S, slot_close(E, es), (long long)s_ie[s] - en);
            it_idx[k] = (uint16_t)idx;
        }
        group_sync();
        TW_TILE_TICK(3);
        // ---- 4. tuples, one per lane: feasibility and score ----------------------------------------------------------
        for (int g = t; g < nG; g += nt) {
            const int s = span_of(s_grid0, seg, send, g0 + g);
            int32_t kx[E], ix[E], st[E], en[E];
            const int gi = g0 + g - s_grid0[s];
            uint32_t real = 1;   // tuples of the span (its slots beyond are padding)
            {
                const unsigned long long pk = s_cn[s];
#pragma unroll
                for (int e = 0; e < E; e++) real *= (uint32_t)((pk >> (8 * e)) & 255ull);
            }
            bool ok = (uint32_t)gi < real;
            decode(s, ok ? gi : 0, it0, kx, ix);
#pragma unroll
            for (int e = 0; e < E; e++) {
                st[e] = sl_st[e][ix[e]]; en[e] = sl_en[e][ix[e]];
#pragma unroll
                for (int p = 0; p < e; p++)
                    if (((dag_pm[e] >> p) & 1) && en[p] > st[e]) ok = false;   // traceweaver_v3.py:343-347
            }
            double sj = 0.0;
            if (ok) {
                // ScoreAssignmentAsPerInvocationGraph, no-skip branch (traceweaver_v1.py:305-361)
                int last = 0;
                int32_t last_end = en[0];
#pragma unroll
                for (int e = 1; e < E; e++) if (en[e] > last_end) { last_end = en[e]; last = e; }
                S.gp = P.gparam + (U.gp_off + (int64_t)((first + s) / P.batch_size) * U.nslot) * 4;
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const int np = (int)dag_np[e];
#pragma unroll
                    for (int j = 0; j < E; j++) {
                        if (j >= np) continue;
                        const uint32_t pj = (dag_pl[e] >> (4 * j)) & 15u;
                        if (!(pj & 8u)) continue;
                        const int p = (int)(pj & 7u);
                        int32_t pend = 0;
#pragma unroll
                        for (int q = 0; q < E; q++) if (q == p) pend = en[q];
                        sj += score_term_gap(S, slot_prim(E, p, e), (long long)st[e] - pend);
                    }
                    if (np == 0) sj += t_root[kx[e]];
                    if (e == last) sj += t_close[kx[e]];
                }
                if (ordered) {
                    atomicAdd(&s_leaves[s], 1);
                    if (pass == 1) {
#pragma unroll
                        for (int e = 0; e < E; e++) atomicOr(&s_bits[e][s], 1u << (ix[e] - (int)s_lo[e][s]));
                    }
                }
            }
            g_sc[g] = ok ? sj : dnan();
            g_span[g] = (uint8_t)s;
        }
        group_sync();
        TW_TILE_TICK(4);
        // ---- 5. rank of every tuple among the tuples of its span ------------------------------------------------------
        for (int g = t; g < nG; g += nt) {
            const double my = g_sc[g];
            int rank = kTopK;
            if (my == my) {
                const int s = g_span[g];
                const int gs = s_grid0[s] - g0, ge = s_grid0[s + 1] - g0;
                rank = 0;
                bool partner = false;
                // four scores a step (the span's slots are padded to a multiple of four with NaN, which compares as neither
                // greater nor equal): how many are greater, how many are equal -- the tuple itself is one of the equal ones
                int eq = 0;
                for (int h0 = gs; h0 < ge && rank < kTopK; h0 += 4) {
                    const double o0 = g_sc[h0], o1 = g_sc[h0 + 1], o2 = g_sc[h0 + 2], o3 = g_sc[h0 + 3];
                    rank += (o0 > my) + (o1 > my) + (o2 > my) + (o3 > my);
                    eq += (o0 == my) + (o1 == my) + (o2 == my) + (o3 == my);
                }
                if (eq > 1 && rank < kTopK) {
                    // equal scores (millisecond-granular data): Python compares the span lists -- start_mus of the first differing
                    // span (spans.py:51-52); count the equal ones that are greater, note one that is incomparable
                    int32_t ka[E], ia[E];
                    decode(s, g - gs, it0, ka, ia);
                    for (int h = gs; h < ge; h++) {
                        if (h == g || !(g_sc[h] == my)) continue;
                        int32_t kb[E], ib[E];
                        decode(s, h - gs, it0, kb, ib);
                        int ord = 0;   // +1: h is greater
                        bool done = false;
#pragma unroll
                        for (int e = 0; e < E; e++)
                            if (!done && ia[e] != ib[e]) {
                                const int32_t a = sl_st[e][ia[e]], b = sl_st[e][ib[e]];
                                ord = b > a ? 1 : (b < a ? -1 : 0);
                                done = true;
                            }
                        if (ord > 0) rank++;
                        else if (ord == 0) partner = true;
                    }
                }
                if (partner && rank < kTopK) { s_amb[s] = 1; s_anyamb = 1; }
            }
            g_rank[g] = (uint8_t)(rank < kTopK ? rank : kTopK);
        }
        group_sync();
        TW_TILE_TICK(5);
        // ---- results: the tuples of rank < 5 write themselves; then one thread per span ---------------------------------
        for (int g = t; g < nG; g += nt) {
            if (g_rank[g] >= kTopK) continue;
            const int s = g_span[g];
            if (s_amb[s]) continue;
            int32_t kx[E], ix[E];
            decode(s, g0 + g - s_grid0[s], it0, kx, ix);
            const int k = g_rank[g], si = first + s;
            P.tk_score[tks_index(U, k, si)] = g_sc[g];
#pragma unroll
            for (int e = 0; e < E; e++) P.tk_idx[tk_index(U, k, e, si)] = s_A[e] + ix[e];
        }
        // ---- 5b. a span whose top five the ranks do not decide (s_amb: equal scores, tuples Python's order does not compare --
        // millisecond-granular traces have them in a third of their spans): the reference's procedure itself, by the span's own
        // thread -- CPython's heap over the span's tuples in enumeration order (traceweaver_v3.py:304-307), then its list.sort --
        // on the scores of phase 4.  (Such a span used to go to the wavefront kernel, which staged its windows and evaluated its terms
        // again for a wavefront's ~90 us: 10.7 of the 29 ms of a pass on the nodejs shape at 14.4 M spans.)  The term tables of the
        // segment are free by now: kTopK + 1 slots of each per span hold the heap for the sort.
        if (s_anyamb != 0) group_sync();   // (the lanes above skip the tuples of undecided spans: s_amb is cleared only when all have passed; the flag is the same for all since the barrier after the ranks)
        if (s_anyamb != 0 && live && t >= seg && t < send && mine && s_amb[t] && (t + 1) * (kTopK + 1) <= C::kItems && s_leaves[t] <= kTileReplayMax) {
            RegHeap RH;
            RH.clear(E);
            const unsigned long long pk = s_cn[t];
            auto start_of = [&](int e, int pos) -> int32_t { return sl_st[e][it_idx[s_item0[t] - it0 + pos_base(pk, e) + pos]]; };
            uint32_t real = 1;
#pragma unroll
            for (int e = 0; e < E; e++) real *= (uint32_t)((pk >> (8 * e)) & 255ull);
            const int gs = s_grid0[t] - g0;
            for (uint32_t gi = 0; gi < real; gi++) {
                const double sc = g_sc[gs + gi];
                if (!(sc == sc) || RH.below_root(sc)) continue;   // infeasible / strictly below the root of a full heap: the push leaves the array as it is
                unsigned long long x = 0;
                uint32_t rest = gi;
#pragma unroll
                for (int e = E - 1; e >= 0; e--) {
                    const uint32_t c = (uint32_t)((pk >> (8 * e)) & 255ull);
                    const uint32_t q = c == 1 ? rest : div_small(rest, c);
                    x |= (unsigned long long)(rest - q * c) << (8 * e);
                    rest = q;
                }
                RH.push(sc, x, start_of);
            }
            double* hs = t_root + t * (kTopK + 1);
            unsigned long long* hx = reinterpret_cast<unsigned long long*>(t_close + t * (kTopK + 1));
#pragma unroll
            for (int k = 0; k < kTopK; k++) { hs[k] = RH.s[k]; hx[k] = RH.x[k]; }
            py_sort_desc(hs, hx, RH.n, [&](double sa, unsigned long long xa, double sb, unsigned long long xb) { return RH.lt(sa, xa, sb, xb, start_of); });
            const int si = first + t;
            for (int k = 0; k < RH.n; k++) {
                P.tk_score[tks_index(U, k, si)] = hs[k];
#pragma unroll
                for (int e = 0; e < E; e++)
                    P.tk_idx[tk_index(U, k, e, si)] = s_A[e] + (int)it_idx[s_item0[t] - it0 + pos_base(pk, e) + (int)((hx[k] >> (8 * e)) & 255ull)];
            }
            s_amb[t] = 0;   // (read again only by this thread: below, and when the tile hands its undecided spans on)
        }
        if (live && t >= seg && t < send && mine && !s_amb[t]) {
            // a span's list has min(5, feasible tuples) entries in every pass; the unused entries keep the -1 / NaN
            // pattern of tw_load_batch and are never written (most lists are short: this halves the store traffic)
            const int64_t g = U.in_off + i;
            const int leaves = s_leaves[t];
            P.tk_n[g] = leaves < kTopK ? leaves : kTopK;
            P.leaves[g] = leaves;
            if (pass == 1) P.leaves0[g] = leaves;
            P.rep[g] = 0;
            if (pass == 1) {
#pragma unroll
                for (int e = 0; e < E; e++) {
                    P.c_bits[ie_index(U, e, i) * kCandWords] = (uint64_t)(ordered ? s_bits[e][t] : s_cm[e][t]);
                    for (int w = 1; w < kCandWords; w++) P.c_bits[ie_index(U, e, i) * kCandWords + w] = 0;
                }
            }
        }
        group_sync();
        TW_TILE_TICK(6);
        seg = send;
    }
    TW_TILE_FLUSH();
    // rare (millisecond-granular data): CPython's heapq / list.sort must be replayed push by push -- the wavefront kernel does
    heavy_append<E>(P, mine && s_amb[live ? t : 0] != 0, true, false, T.unit, i);
}

}  // namespace tw
