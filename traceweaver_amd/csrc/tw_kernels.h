// tw_kernels.h -- HIP kernels of the span->parent assignment engine (gfx950).
//
// Kernel map (reference function -> kernel), paths under
// /root/reference/src/trace_reconstructor/ports/python/algorithms/:
//   k_block_params      ComputeEpPairDistParams3            traceweaver_v3.py:580-646
//   k_enumerate_tile (tw_tile.h) / k_enumerate_heavy
//                       FindCutoffs + DfsTraverseX/3 + ScoreAssignmentAsPerInvocationGraph + heap top-5
//                                                           traceweaver_v3.py:182-351, traceweaver_v1.py:259-361
//   k_scan_* / k_perfect_cut / k_window_flags / k_window_index
//                       CreateWindows2 + PerfectCut         traceweaver_v3.py:1020-1078
//   k_select            BuildMISInstance + Gurobi_MIS       traceweaver_v3.py:1252-1281,1395-1419
//   k_claim / k_detect / k_repair
//                       AddAssignment(delete_out_spans)     traceweaver_v1.py:457-463 (span consumption)
//   k_finalize          AddAssignment bookkeeping, counters traceweaver_v1.py:433-455, traceweaver_v3.py:1201-1217
//   k_gaps              ComputeEpPairDistParams5 durations  traceweaver_v3.py:717-762
#pragma once
#include "tw_device.h"

namespace tw {

// ---------------------------------------------------------------------------------------------
// small helpers
__device__ __forceinline__ void raise_err(const Dev& P, int code) { atomicCAS(P.err, 0, code); }

// Appends to a work list with one atomic per wavefront instead of one per lane (same-address atomics serialise
// at ~20 ns each): the first active lane reserves slots for every lane that has an entry.  Returns the slot or -1.
__device__ __forceinline__ int wave_append(int32_t* counter, bool pred) {
    const unsigned long long mask = __ballot(pred);
    if (mask == 0) return -1;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(mask));
    base = __shfl(base, leader);
    return pred ? base + __popcll(mask & ((1ull << lane) - 1ull)) : -1;
}
// `n` consecutive units of a budget that is handed out by a bump counter: the first of them, or -1 when they do not fit.  What is
// added is never taken back.  (Subtracting on refusal hands the same units out twice -- A adds, B adds behind A and is served, A is
// refused and subtracts, C is served from where A began: inside B's units.  Two split spans then shared scratch slots, and one of them
// came back with another span's tuples, about once in sixty batches of the stress units.  A compare-and-swap loop is exact but takes
// turns: 2 400 lanes asking at once held the tile kernel of the media shape at eight requests in flight for 13 ms.)  A request that
// finds the budget spent does not add -- the counter stays near the budget whatever the batch --; one that overshoots leaves a hole.
__device__ __forceinline__ int bump_reserve(int32_t* counter, int n, int budget) {
    if (atomicAdd(counter, 0) + n > budget) return -1;
    const int old = atomicAdd(counter, n);
    return old + n <= budget ? old : -1;
}

// Barrier of a workgroup that is a single wavefront: LDS operations of one wavefront execute in program order, so
// only the compiler has to be kept from reordering.  (__syncthreads() would also wait for every outstanding global
// load/store -- the release fence at workgroup scope -- which costs microseconds per use in the work-list kernels.)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void group_sync() {  // workgroup barrier of kernels that run with one wavefront or several
    if (blockDim.x <= 64) wave_sync(); else __syncthreads();
}
constexpr int kWorkChunk = 4;  // work-list entries a wavefront claims per atomic

__device__ __forceinline__ int lower_bound_i64(const int64_t* a, int n, int64_t t) {  // first a[x] >= t
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < t) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ int upper_bound_i64(const int64_t* a, int n, int64_t t) {  // first a[x] > t
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= t) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Galloping variants: spans are rank-aligned in no-skip mode (n_out == n_in, both sorted by start), so the
// answer lies within a few positions of the incoming span's own rank.  kUpper: first a[x] > t, else >= t.
template <bool kUpper>
__device__ __forceinline__ int bound_near(const int64_t* a, int n, int64_t t, int g) {
    auto before = [&](int x) { return kUpper ? a[x] <= t : a[x] < t; };  // x lies strictly before the answer
    g = g < 0 ? 0 : (g > n ? n : g);
    int lo, hi;
    if (g >= n || !before(g)) {  // answer <= g
        hi = g;
        int step = 1;
        while (true) {
            lo = hi - step;
            if (lo < 0) { lo = 0; break; }
            if (before(lo)) { lo = lo + 1; break; }
            hi = lo;
            step <<= 1;
        }
    } else {  // answer > g
        lo = g + 1;
        int step = 1;
        while (true) {
            hi = lo + step - 1;
            if (hi >= n) { hi = n; break; }
            if (!before(hi)) break;
            lo = hi + 1;
            step <<= 1;
        }
    }
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (before(mid)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------------------------------------
// Which key bits differ at all?  acc[0] |= key ^ first key, acc[1] |= key, over `nseg` index ranges.  The radix
// sorts then skip the bit positions that are the same in every key (timestamps of one corpus share their
// upper ~34 bits; gap samples are integers stored as doubles, their low mantissa bits are zero).
__global__ void k_key_bits(const unsigned long long* keys, const uint32_t* seg_begin, const uint32_t* seg_end, int nseg,
                           unsigned long long* acc) {
    const unsigned long long k0 = keys[seg_begin[0]];
    unsigned long long diff = 0, any = 0;
    for (int s = blockIdx.y; s < nseg; s += gridDim.y) {
        const uint32_t a = seg_begin[s], b = seg_end[s];
        for (uint32_t i = a + blockIdx.x * blockDim.x + threadIdx.x; i < b; i += gridDim.x * blockDim.x) {
            const unsigned long long k = keys[i];
            diff |= k ^ k0;
            any |= k;
        }
    }
    // one atomic pair per workgroup: wavefronts by shuffles, then through LDS
    __shared__ unsigned long long sh_diff, sh_any;
    if (threadIdx.x == 0) { sh_diff = 0; sh_any = 0; }
    __syncthreads();
    for (int off = 32; off >= 1; off >>= 1) {
        if (off < (int)blockDim.x) { diff |= __shfl_down(diff, off); any |= __shfl_down(any, off); }
    }
    if ((threadIdx.x & 63) == 0) { atomicOr(&sh_diff, diff); atomicOr(&sh_any, any); }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (sh_diff & ~acc[0]) atomicOr(&acc[0], sh_diff);
        if (sh_any & ~acc[1]) atomicOr(&acc[1], sh_any);
    }
}

// Rows sorted as one array.  A segmented sort hands every long row to a single workgroup; instead the rows are
// packed into composite keys (row number above the kb varying key bits), sorted by one device-wide radix sort over
// kb + log2(rows) bits, and unpacked again.  Bits of a key outside [begin_bit, begin_bit + kb) are the same in
// every key (k_key_bits) and are restored from `fixed`.
__global__ void k_rows_pack(const unsigned long long* keys, const uint32_t* seg_begin, const uint32_t* seg_end, const uint32_t* dst_off,
                            int nseg, int begin_bit, int kb, unsigned long long* comp) {
    const unsigned long long mask = kb >= 64 ? ~0ull : ((1ull << kb) - 1);
    for (int s = blockIdx.y; s < nseg; s += gridDim.y) {
        const uint32_t a = seg_begin[s], b = seg_end[s], d = dst_off[s];
        for (uint32_t i = a + blockIdx.x * blockDim.x + threadIdx.x; i < b; i += gridDim.x * blockDim.x)
            comp[d + (i - a)] = (kb >= 64 ? 0ull : ((unsigned long long)s << kb)) | ((keys[i] >> begin_bit) & mask);
    }
}
__global__ void k_rows_unpack(const unsigned long long* comp, const uint32_t* seg_begin, const uint32_t* seg_end, const uint32_t* dst_off,
                              int nseg, int begin_bit, int kb, unsigned long long fixed, unsigned long long* out) {
    const unsigned long long mask = kb >= 64 ? ~0ull : ((1ull << kb) - 1);
    for (int s = blockIdx.y; s < nseg; s += gridDim.y) {
        const uint32_t a = seg_begin[s], b = seg_end[s], d = dst_off[s];
        for (uint32_t i = a + blockIdx.x * blockDim.x + threadIdx.x; i < b; i += gridDim.x * blockDim.x)
            out[i] = fixed | ((comp[d + (i - a)] & mask) << begin_bit);
    }
}

// ---------------------------------------------------------------------------------------------
// End times of every list in ascending order (the rank-aligned arrays of ComputeEpPairDistParams3, traceweaver_v3.py:590-646,
// `sorted(...)` of the ends).  The lists are sorted by (start, end), so their ends are nearly sorted: span j > i ends before
// span i only if it started while i was in flight.  Rank = index + (later spans that end earlier) - (earlier spans that end
// later): one thread per span walks forward over the spans that start before it ends -- as many as are in flight with it --
// and counts; 16 B per span read once, 8 B written, 4 B of counter: a device-wide radix sort of the same keys took five
// read + write passes (43 % of the HBM traffic of a step in round 2).  Ties keep their order (equal values: same array).
__device__ __forceinline__ int set_tile(const TileSet& S) { return S.ids != nullptr ? S.ids[blockIdx.x] : (int)blockIdx.x; }   // the tile of this workgroup (TileSet: tw_device.h)
__device__ __forceinline__ TileSet all_tiles(const Dev& P) { return TileSet{nullptr, 0, P.n_tiles, 0}; }

__device__ __forceinline__ void rank_list(const int64_t* st, const int64_t* en, int n, int i, int32_t* delta) {
    const int64_t a = en[i];
    int cnt = 0;
    for (int j = i + 1; j < n && st[j] < a; j++)
        if (en[j] < a) { cnt++; atomicAdd(&delta[j], -1); }
    if (cnt) atomicAdd(&delta[i], cnt);
}
__global__ void k_rank_ends(Dev P, TileSet S, int32_t* d_in, int32_t* d_out) {   // both zeroed
    const TileDev Tl = P.tiles[set_tile(S)];
    const UnitDev& U = P.units[Tl.unit];
    const int i = Tl.first + threadIdx.x;
    if (i >= U.n_in) return;
    rank_list(P.in_start + U.in_off, P.in_end + U.in_off, U.n_in, i, d_in + U.in_off);
    for (int e = 0; e < U.E; e++)
        rank_list(P.out_start + U.ep_off[e], P.out_end + U.ep_off[e], (int)(U.ep_off[e + 1] - U.ep_off[e]), i, d_out + U.ep_off[e]);
}
__global__ void k_place_ends(Dev P, TileSet S, const int32_t* d_in, const int32_t* d_out) {
    const TileDev Tl = P.tiles[set_tile(S)];
    const UnitDev& U = P.units[Tl.unit];
    const int i = Tl.first + threadIdx.x;
    if (i >= U.n_in) return;
    P.in_end_sorted[U.in_off + i + d_in[U.in_off + i]] = P.in_end[U.in_off + i];
    for (int e = 0; e < U.E; e++)
        P.out_end_sorted[U.ep_off[e] + i + d_out[U.ep_off[e] + i]] = P.out_end[U.ep_off[e] + i];
}

// ---------------------------------------------------------------------------------------------
// Pass-1 Gaussian parameters: one thread per (unit, 100-span block, slot).
// mean = (sum t2 - sum t1)/n over rank-aligned sorted arrays, std = sqrt(ceil(n/10)) * tstd(batch means).
__global__ void k_block_params(Dev P, int64_t total, int cls) {   // cls > 0: only the units of that endpoint count (a class' launch on its own stream)
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    // locate unit by gp_off (units are few: linear/binary search on the descriptor table)
    int lo = 0, hi = P.n_units - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (P.units[mid].gp_off <= g) lo = mid; else hi = mid - 1;
    }
    const UnitDev& U = P.units[lo];
    if (cls > 0 && U.E != cls) return;
    const int64_t r = g - U.gp_off;
    const int b = (int)(r / U.nslot), q = (int)(r % U.nslot), E = U.E;
    double* out = P.gparam + g * 4;
    const int64_t *t1 = nullptr, *t2 = nullptr;
    if (q < E) {  // root(in -> e): only when e has no DAG predecessor
        if (U.npred[q] == 0) { t1 = P.in_start + U.in_off; t2 = P.out_start + U.ep_off[q]; }
    } else if (q < E + E * E) {
        const int p = (q - E) / E, e = (q - E) % E;
        bool prim = false;
        for (int j = 0; j < U.npred[e]; j++) prim |= (U.pred_list[e][j] == p && U.pred_prim[e][j]);
        if (prim) { t1 = P.out_end_sorted + U.ep_off[p]; t2 = P.out_start + U.ep_off[e]; }
    } else {
        const int e = q - E - E * E;
        t1 = P.out_end_sorted + U.ep_off[e];
        t2 = P.in_end_sorted + U.in_off;
    }
    if (t1 == nullptr) { out[0] = out[1] = out[2] = out[3] = dnan(); return; }
    const int a = b * P.batch_size, z = min(a + P.batch_size, U.n_in), len = z - a;
    double bm[10], mean;
    const int bs = (len + 9) / 10;
    int m = 0;
    if (!U.float_time) {  // integer microseconds: Python sums ints exactly, one correctly rounded division
        int64_t s1 = 0, s2 = 0;
        for (int kb = 0; kb < 10; kb++) {
            const int st = kb * bs, en = min((kb + 1) * bs, len);
            if (en - st <= 0) continue;
            int64_t u1 = 0, u2 = 0;
            for (int i = a + st; i < a + en; i++) { u1 += t1[i]; u2 += t2[i]; }
            s1 += u1; s2 += u2;
            bm[m++] = (double)(u2 - u1) / (double)(en - st);
        }
        mean = (double)(s2 - s1) / (double)len;
    } else {  // load-scaled unit: Python's sum() over floats adds left to right in binary64 (helpers/transforms.py:21,30
              // make every timestamp a float); the timestamps are exact images of those floats, and scaling by a power
              // of two commutes with every rounding below, so the scale is applied once at the end
        double s1 = 0.0, s2 = 0.0;
        for (int kb = 0; kb < 10; kb++) {
            const int st = kb * bs, en = min((kb + 1) * bs, len);
            if (en - st <= 0) continue;
            double u1 = 0.0, u2 = 0.0;
            for (int i = a + st; i < a + en; i++) {
                const double v1 = (double)t1[i], v2 = (double)t2[i];
                u1 += v1; u2 += v2; s1 += v1; s2 += v2;
            }
            bm[m++] = (u2 - u1) / (double)(en - st);
        }
        mean = (s2 - s1) / (double)len;
    }
    const double mu = np_sum(bm, m) / (double)m;
    double d2[10];
    for (int kb = 0; kb < m; kb++) { const double d = bm[kb] - mu; d2[kb] = d * d; }
    const double var = np_sum(d2, m) / (double)(m - 1);
    const double sd = (sqrt((double)bs) * sqrt(var)) * U.tscale;  // in microseconds
    if (sd != sd) raise_err(P, TW_ERR_NAN_PARAMS);
    const double used = sd < 1.0e-12 ? 0.001 : sd;  // traceweaver_v1.py:130-131
    // The scorer works on raw timestamp differences: y = (x*tscale - mean*tscale) / used == (x - mean) / (used / tscale)
    // bit for bit when tscale is a power of two, so the mean stays in timestamp units and the divisor is rescaled;
    // tw_get_gauss_params reports the mean in microseconds.
    out[0] = mean; out[1] = sd; out[2] = tw_log(used); out[3] = used / U.tscale;
}

// Mixture constants for pass 2: [slot][comp] = mean * prec_chol, prec_chol * tscale, log(prec_chol), log(weight).
// sklearn evaluates y = x * prec_chol - mean * prec_chol with x in microseconds; on raw timestamp differences
// (x = d * tscale, tscale a power of two) d * (prec_chol * tscale) is the same binary64 number as (d * tscale) * prec_chol.
// Slots without a mixture (n <= 0: the "(0,0)" fallback of traceweaver_v3.py:765-766, scored as a Gaussian with
// mean 0 and std 0.001, traceweaver_v1.py:130-131) get those Gaussian constants in component 0.
__global__ void k_mix_consts(const double* mix_p, const int32_t* mix_n, const int32_t* slot_unit, const UnitDev* units, double* mix_c, int64_t total) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int64_t slot = g / kMaxComp;
    const double tscale = units[slot_unit[slot]].tscale;
    if (mix_n[slot] <= 0) {
        if (g % kMaxComp == 0) { mix_c[g * 4 + 0] = 0.0; mix_c[g * 4 + 1] = 0.001 / tscale; mix_c[g * 4 + 2] = tw_log(0.001); mix_c[g * 4 + 3] = 0.0; }
        return;
    }
    const double w = mix_p[g * 3 + 0], mu = mix_p[g * 3 + 1], pc = mix_p[g * 3 + 2];
    mix_c[g * 4 + 0] = mu * pc; mix_c[g * 4 + 1] = pc * tscale; mix_c[g * 4 + 2] = tw_log(pc); mix_c[g * 4 + 3] = tw_log(w);
}

// ---------------------------------------------------------------------------------------------
// Scoring terms (traceweaver_v1.py:117-139)
struct Scorer {
    int pass;            // 1 Gaussian, 2 mixtures
    const double* gp;    // this span's block of Gaussian parameters [nslot][4]
    const int32_t* mix_n;
    const double* mix_c; // [nslot][kMaxComp][4]
};

// x = t2 - t1 in timestamp units; mean in the same units, `used` = std / tscale (k_block_params)
__device__ __forceinline__ double term_gauss(double mean, double used, double logstd, double x) {
    const double y = (x - mean) / used;
    return (-(y * y) / 2.0 - kLogSqrt2Pi) - logstd;  // scipy.stats.norm.logpdf
}

// not inlined: the mixture term is ~1.5k instructions and is called from several places per kernel; keeping one
// copy keeps the enumeration kernels inside the instruction cache
__device__ __noinline__ double term_mix(int n, const double* c, double x) {
    // the per-component terms stay in registers (an array indexed by the loop counter would live in scratch memory:
    // one store and two loads per component and call); k < kMaxComp = 5
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0, amax = -dinf();
    for (int k = 0; k < n; k++) {
        const double y = x * c[k * 4 + 1] - c[k * 4 + 0];  // x_us * prec_chol - mean * prec_chol (k_mix_consts)
        const double lp = y * y;
        const double ak = (-0.5 * (kLog2Pi + lp) + c[k * 4 + 2]) + c[k * 4 + 3];  // sklearn _estimate_weighted_log_prob
        a0 = k == 0 ? ak : a0; a1 = k == 1 ? ak : a1; a2 = k == 2 ? ak : a2; a3 = k == 3 ? ak : a3; a4 = k == 4 ? ak : a4;
        if (ak > amax) amax = ak;
    }
    double m = 0.0, s = 0.0;  // scipy.special.logsumexp: max split out, log1p of the rest
    for (int k = 0; k < n; k++) {   // (left to the compiler's unroller: rolled, the five values cost 4 more VGPRs and one wave of occupancy in three kernels)
        const double ak = k == 0 ? a0 : (k == 1 ? a1 : (k == 2 ? a2 : (k == 3 ? a3 : a4)));
        if (ak == amax) m += 1.0; else s += tw_exp_nonpos(ak - amax);   // adding the 0.0 of a maximal component changes nothing: s >= +0.0
    }
    if (s != 0.0) s = s / m;
    return (tw_log1p(s) + (m == 1.0 ? 0.0 : tw_log(m))) + amax;  // log(1.0) is exactly 0.0
}

__device__ __forceinline__ double score_term(const Scorer& S, int slot, int64_t t1, int64_t t2) {
    const double x = (double)(t2 - t1);
    if (S.pass == 1) {
        const double* g = S.gp + slot * 4;
        return term_gauss(g[0], g[3], g[2], x);
    }
    const int n = S.mix_n[slot];
    const double* c = S.mix_c + (int64_t)slot * kMaxComp * 4;
    if (n <= 0) return term_gauss(c[0], c[1], c[2], x);  // "(0,0)" fallback, traceweaver_v3.py:765-766 (constants from k_mix_consts)
    return term_mix(n, c, x);
}

// ---------------------------------------------------------------------------------------------
// Candidate enumeration for one incoming span.  E is a compile-time constant so that the index
// tuple, the kept lists and the cutoffs live in registers.
template <int E>
struct Cand {
    double score;
    int32_t idx[E];
};

// Speculative enumeration on all spans: this *is* top_k_2 (traceweaver_v3.py:1185) and equals top_k
// (traceweaver_v3.py:1182) for every span none of whose candidates was consumed by an earlier window.
//
// Two tiers.  k_enumerate_tile (tw_tile.h): one workgroup per tile of incoming spans computes the cut-offs, stages the
// tile's candidate windows in LDS and enumerates the spans with up to kTileMax tuples, items and tuples spread over its
// lanes.  Spans with a longer enumeration (or windows that do not fit its tables) go to a work list and are enumerated by
// k_enumerate_heavy, one wavefront per span: the wavefront walks the leading endpoints together and
// spreads the tuples of the trailing endpoints over its lanes -- feasibility and the log-likelihood (the
// costly part) run in parallel, the heap is then fed in enumeration order (wave-uniform pushes) so that
// ties resolve exactly as in the sequential reference.
constexpr int kHeavyThreads = 64;

// Optional phase timers (build with -DTW_PROFILE; read back through tw_debug_profile): cycles spent by
// lane 0 of every heavy-enumeration wavefront per phase.  Compiled out by default.
#ifdef TW_PROFILE
#ifndef TW_PROFILE_E
#define TW_PROFILE_E 4   // the endpoint-count class whose narrow instantiation is timed
#endif
// prof[12] longest item (ticks << 24 | tuples, capped), [13] sum of item ticks, [14] items, [8] sum of wavefront lifetimes,
// [9] wavefronts that drew work, [6] longest wavefront lifetime, [15] items >= 100 us -- of the narrow E = 4 instantiation
#define TW_PROF_DECL() unsigned long long _tw_info = 0ull; const long long _tw_born = wall_clock64(); long long _tw_t = _tw_born, _tw_p = _tw_born, _tw_ph[5] = {0, 0, 0, 0, 0}, _tw_it[5] = {0, 0, 0, 0, 0}; const bool _tw_on = (E == TW_PROFILE_E && !kWide && mode == 0)
#define TW_ITEM_BEGIN() do { _tw_t = wall_clock64(); _tw_p = _tw_t; for (int _k = 0; _k < 5; _k++) _tw_it[_k] = 0; } while (0)
// phases of an item: [0] stage candidates, [1] term and pair tables, [2] tuple list, [3] walk + top-5, [4] results
#define TW_PHASE(k) do { const long long _n = wall_clock64(); _tw_ph[k] += _n - _tw_p; _tw_it[k] += _n - _tw_p; _tw_p = _n; } while (0)
#define TW_ITEM_END(tuples) do { if (_tw_on && threadIdx.x == 0) { const unsigned long long _d = (unsigned long long)(wall_clock64() - _tw_t); \
    const unsigned long long _l = (unsigned long long)(tuples) > 0xffffffull ? 0xffffffull : (unsigned long long)(tuples); \
    if (atomicMax((unsigned long long*)&P.prof[12], (_d << 24) | _l) < ((_d << 24) | _l)) { P.prof[11] = ((unsigned long long)_tw_it[0] << 48) | ((unsigned long long)_tw_it[1] << 32) | ((unsigned long long)_tw_it[2] << 16) | (unsigned long long)_tw_it[3]; P.prof[10] = (unsigned long long)_tw_it[3]; P.prof[7] = (unsigned long long)_tw_it[2]; P.prof[5] = _tw_info; } atomicAdd((unsigned long long*)&P.prof[13], _d); atomicAdd((unsigned long long*)&P.prof[14], 1ull); \
    if (_d >= 10000ull) atomicAdd((unsigned long long*)&P.prof[15], 1ull); \
    { int _b = 63 - __clzll((long long)(_d | 1ull)) - 9; _b = _b < 0 ? 0 : (_b > 15 ? 15 : _b); atomicAdd((unsigned long long*)&P.prof[16 + _b], 1ull); } } } while (0)   /* [16 + b]: items of 5.12 us x 2^b .. 2^(b+1) */
#define TW_ITEM_INFO(v) do { _tw_info = (v); } while (0)
#define TW_PROF_FLUSH() do { if (_tw_on && threadIdx.x == 0) { const unsigned long long _d = (unsigned long long)(wall_clock64() - _tw_born); \
    atomicAdd((unsigned long long*)&P.prof[8], _d); atomicAdd((unsigned long long*)&P.prof[9], 1ull); atomicMax((unsigned long long*)&P.prof[6], _d); \
    for (int _k = 0; _k < 5; _k++) atomicAdd((unsigned long long*)&P.prof[_k], (unsigned long long)_tw_ph[_k]); } } while (0)
#else
#define TW_PROF_DECL() do {} while (0)
#define TW_ITEM_BEGIN() do {} while (0)
#define TW_PHASE(k) do {} while (0)
#define TW_ITEM_END(tuples) do {} while (0)
#define TW_ITEM_INFO(v) do {} while (0)
#define TW_PROF_FLUSH() do {} while (0)
#endif

// Work list of k_enumerate_heavy<E, W>: the class' slice [heavy_in_off[E], heavy_in_off[E+1]) of heavy_in_unit /
// heavy_in_idx is filled from the front with the spans whose candidate windows are all <= kNarrow wide (served by
// the small-LDS instantiation, more wavefronts per CU) and from the back with the others.
constexpr int kNarrow = 32;
// Narrow spans whose enumeration is long (product of the staged candidates > kBigProduct) go to a list of their own that
// the narrow instantiation serves first, one to a wavefront: a persistent kernel ends with its longest item, so the long
// ones must not start last.
#ifndef TW_BIG_PRODUCT
#define TW_BIG_PRODUCT 768
#define TW_SPLIT_MIN 4096
#define TW_SPLIT_GRAIN 2048
#define TW_GRID_TARGET 1024
#endif
constexpr int kBigProduct = TW_BIG_PRODUCT;
// One wavefront scores ~0.5 tuples per ns when it has a SIMD to itself: a span with 4e4 tuples (one in 1e5 on the bench
// workload) kept a kernel alive for 1.5 ms that was otherwise done after 0.6 ms.  From kSplitMin contained grid points on, the
// span is listed kSplitGrain points a part (at most kMaxParts, at most one part per candidate of the first endpoint): part p
// takes the tuples whose first span is among the p-th share of that endpoint's staged candidates.  (The host-emulation
// build of the tests sets tiny thresholds so that the route is exercised.)
#ifndef TW_MAX_PARTS
#define TW_MAX_PARTS 48
#endif
// kMaxParts: parts of a span cut by its listed prefixes (deferred spans, see kListSplitFlag); kMaxCandParts: by the first endpoint's candidate
constexpr int kSplitMin = TW_SPLIT_MIN, kSplitGrain = TW_SPLIT_GRAIN, kMaxParts = TW_MAX_PARTS, kMaxCandParts = 16;
static_assert(kMaxParts >= kMaxCandParts && kMaxParts <= 255, "part numbers are eight bits");
// A span with twin candidates (Python's order of tuples may not decide between two of its tuples) is split in LOG MODE: CPython's
// size-5 heap (traceweaver_v3.py:304-307) is an exact function of the sequence of pushes, and a push whose score is strictly below
// the score of the root of a full heap leaves the array as it was.  The root's score is the fifth largest score pushed so far, and
// the fifth largest of a part's own tuples so far is never above that of all tuples so far: so the tuples of a part whose score is
// not below the running root score of the part's OWN heap are a superset of the part's tuples that change the real heap.  Every
// part replays the heap on its share and logs those tuples in order (kPartLogCap of them); k_merge_parts replays the concatenation
// of the logs, part after part -- the reference's heap, push by push, without one wavefront walking the whole enumeration.
#ifndef TW_PART_LOG_CAP
#define TW_PART_LOG_CAP 256
#endif
constexpr int kPartLogCap = TW_PART_LOG_CAP;
constexpr int kPartLogFlag = 1 << 25;   // in heavy_big_part / split_parts
// A span with twin candidates that is enumerated whole goes straight to the replay of CPython's heap (the second attempt of
// k_enumerate_heavy): on millisecond-granular traces the first attempt -- the five largest under the strict part of Python's order --
// nearly always ends undecided, and the span was walked twice.  The replay is the reference's procedure itself, so starting with it
// never changes a result.  In heavy_big_part (long enumerations) / in heavy_in_idx (the others; span indices stay below 2^26).
constexpr int kReplayFlag = 1 << 26;
constexpr int kIdxReplayFlag = 1 << 30;
// Deep call graphs (P.defer_min_e endpoints and more, call-order constraints): a span's cost is its tuple count, which neither the grid
// of its candidates (a thousand times larger) nor ranges of the first endpoint's candidates (most feasible tuples sit under one or two
// of them) predict.  The wavefront that draws such a span lists its prefixes of E - 1 endpoints and counts the last level anyway
// (count_only); from kDeferTuples tuples on it does not go on alone: it copies the list to an arena (P.defer_list), cuts it into
// stretches of about kSplitTuples tuples -- contiguous stretches of the enumeration, balanced by construction -- and appends one LIST
// PART per stretch behind the class' listed entries.  A second launch of the kernel (part = 3) serves them: a list part stages the
// span's candidates like any other item, expands its stretch's last level into its own buffer and scores the tuples from there;
// k_merge_parts combines the parts as it does those cut by the first endpoint's candidate (top five or logs, in part order).
constexpr int kListSplitFlag = 1 << 27;   // in heavy_big_part / split_parts
#ifndef TW_DEFER_TUPLES
#define TW_DEFER_TUPLES 4096
#define TW_SPLIT_TUPLES 2048
#endif
constexpr long long kDeferTuples = TW_DEFER_TUPLES;
constexpr int kSplitTuples = TW_SPLIT_TUPLES;
// The listing itself is one wavefront's work, a batch of 64 prefixes at a time (2.5 us a batch on a SIMD of its own: 1.6 ms for the
// 20 000 prefixes of a 51 000-tuple span): a span whose list reaches kDeferPrefixes entries at a level before the last is cut THERE,
// into stretches of equal numbers of prefixes -- its list parts list the remaining levels themselves (P.part_lvl = the level handed over).
#ifndef TW_DEFER_PREFIXES
#define TW_DEFER_PREFIXES 1024
#define TW_DEFER_PREFIX_GRAIN 64
#endif
constexpr int kDeferPrefixes = TW_DEFER_PREFIXES, kDeferPrefixGrain = TW_DEFER_PREFIX_GRAIN;
constexpr long long kDeferListPerSpan = 64, kDeferListMin = 1ll << 22;   // entries of the arena per incoming span of the deferring classes / at least
template <int E>
__device__ __forceinline__ bool heavy_append(const Dev& P, bool pred, bool narrow, bool big, int unit, int i, long long prod = 0, int first_cands = 0,
                                             bool twins = false) {
    const bool isbig = pred && big;   // narrow or wide windows: the list entry says which instantiation takes it
    const int wide_flag = (narrow ? 0 : 1 << 24) | ((twins && P.split_twins == 2) ? kPartLogFlag : 0);
    int nparts = 1, slot_base = 0;
    if (E >= 2 && isbig && prod >= kSplitMin && first_cands >= 2) {
        long long want = prod / kSplitGrain;
        want = want > kMaxCandParts ? kMaxCandParts : want;
        nparts = (int)(want > first_cands ? first_cands : want);
        if (nparts >= 2) {   // the class' budget of extra list entries (two scratch slots go with each)
            const int old = bump_reserve(&P.part_used[E], nparts - 1, (P.part_off[E + 1] - P.part_off[E]) / 2);
            if (old < 0) { atomicAdd(&P.defer_refused[E], 1); nparts = 1; }   // (refused -- a smaller request may still fit --, and counted)
            else slot_base = P.part_off[E] + 2 * old;
        } else nparts = 1;
    }
    const bool split = nparts > 1;
    if (pred && !narrow) P.any_wide[E] = 1;   // (every writer stores the same value)
    const int sb = wave_append(&P.heavy_big_count[E], isbig && !split);
    const int sn = wave_append(&P.heavy_in_count[E], pred && narrow && !big);
    const int sw = wave_append(&P.heavy_in_count[kMaxEp + 1 + E], pred && !narrow && !big);
    if (!pred) return false;
    if (split) {
        const int base = P.heavy_big_off[E] + atomicAdd(&P.heavy_big_count[E], nparts);
        for (int p = 0; p < nparts; p++) {
            P.heavy_big_unit[base + p] = unit; P.heavy_big_idx[base + p] = i;
            P.heavy_big_part[base + p] = nparts | (p << 8) | wide_flag; P.heavy_big_slot[base + p] = slot_base + p;   // (wide_flag: wide windows, log mode)
        }
        const int k = P.part_off[E] + atomicAdd(&P.split_count[E], 1);
        P.split_unit[k] = unit; P.split_idx[k] = i; P.split_slot[k] = slot_base; P.split_parts[k] = nparts | wide_flag;
        return true;
    }
    if (isbig) {
        const int q = P.heavy_big_off[E] + sb;
        P.heavy_big_unit[q] = unit; P.heavy_big_idx[q] = i; P.heavy_big_part[q] = 1 | (wide_flag & (1 << 24)) | (twins ? kReplayFlag : 0); P.heavy_big_slot[q] = 0;
        return true;
    }
    const int pos = narrow ? P.heavy_in_off[E] + sn : P.heavy_in_off[E + 1] - 1 - sw;
    P.heavy_in_unit[pos] = unit;
    P.heavy_in_idx[pos] = i | (twins ? kIdxReplayFlag : 0);
    return true;
}

__device__ __forceinline__ bool heavy_append_rt(const Dev& P, int E, bool pred, bool narrow, int unit, int i) {  // E is the same for the whole wavefront
    const int sn = wave_append(&P.heavy_in_count[E], pred && narrow);
    const int sw = wave_append(&P.heavy_in_count[kMaxEp + 1 + E], pred && !narrow);
    if (!pred) return false;
    const int pos = narrow ? P.heavy_in_off[E] + sn : P.heavy_in_off[E + 1] - 1 - sw;
    P.heavy_in_unit[pos] = unit;
    P.heavy_in_idx[pos] = i;
    return true;
}

// position of the n-th set bit of m (n < popcount(m)): six halvings
__device__ __forceinline__ int nth_set_bit(unsigned long long m, int n) {
    int pos = 0;
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
        const unsigned long long low = m & ((1ull << w) - 1ull);
        const int c = __popcll(low);
        if (n >= c) { n -= c; m >>= w; pos += w; } else m = low;
    }
    return pos;
}

// a term of pass 2 from the gap itself (score_term computes x = (double)(t2 - t1) and then exactly this)
__device__ __forceinline__ double score_term_mix_x(const Scorer& S, int slot, double x) {
    const int n = S.mix_n[slot];
    const double* c = S.mix_c + (int64_t)slot * kMaxComp * 4;
    if (n <= 0) return term_gauss(c[0], c[1], c[2], x);
    return term_mix(n, c, x);
}

// CPython heap / sort replay on an LDS-resident heap (k_enumerate_heavy, degenerate-tie spans only).  Entries hold
// positions in the staged candidate lists; Python's order on equal scores is start_mus of the first differing span.
template <int E, int W>
struct LdsHeap {
    Cand<E>* heap;
    int nheap;
    const int64_t (*ls)[W];  // staged start times [E][W]
    __device__ bool lt(const Cand<E>& a, const Cand<E>& b) const {
        if (a.score != b.score) return a.score < b.score;
#pragma unroll
        for (int e = 0; e < E; e++)
            if (a.idx[e] != b.idx[e]) return ls[e][a.idx[e]] < ls[e][b.idx[e]];
        return false;
    }
    __device__ void siftdown(int startpos, int pos) {
        const Cand<E> item = heap[pos];
        while (pos > startpos) {
            const int parent = (pos - 1) >> 1;
            if (lt(item, heap[parent])) { heap[pos] = heap[parent]; pos = parent; continue; }
            break;
        }
        heap[pos] = item;
    }
    __device__ void siftup(int pos) {
        const int startpos = pos;
        const Cand<E> item = heap[pos];
        int child = 2 * pos + 1;
        while (child < nheap) {
            const int right = child + 1;
            if (right < nheap && !lt(heap[child], heap[right])) child = right;
            heap[pos] = heap[child];
            pos = child;
            child = 2 * pos + 1;
        }
        heap[pos] = item;
        siftdown(startpos, pos);
    }
    __device__ void push(const Cand<E>& c) {
        heap[nheap++] = c;
        siftdown(0, nheap - 1);
        if (nheap > kTopK) {
            const Cand<E> last = heap[--nheap];
            if (nheap > 0) { heap[0] = last; siftup(0); }
        }
    }
    __device__ void reverse(int n) {
        for (int i = 0, j = n - 1; i < j; i++, j--) { const Cand<E> t = heap[i]; heap[i] = heap[j]; heap[j] = t; }
    }
    __device__ void sort_desc() {
        const int n = nheap;
        if (n < 2) return;
        reverse(n);
        int run = 2;
        if (lt(heap[1], heap[0])) {
            for (int i = 2; i < n; i++, run++) if (!lt(heap[i], heap[i - 1])) break;
            reverse(run);
        } else {
            for (int i = 2; i < n; i++, run++) if (lt(heap[i], heap[i - 1])) break;
        }
        for (int start = run; start < n; start++) {
            int l = 0, r = start;
            const Cand<E> pivot = heap[start];
            do {
                const int p = l + ((r - l) >> 1);
                if (lt(pivot, heap[p])) r = p; else l = p + 1;
            } while (l < r);
            for (int p = start; p > l; p--) heap[p] = heap[p - 1];
            heap[l] = pivot;
        }
        reverse(n);
    }
};

// One wavefront per span.  Nothing lives in scratch memory: what is the same for every lane (cut-offs, the prefix
// being walked, the staged candidates, the term tables, the replay heap) sits in LDS, what differs per lane (its
// grid point) sits in registers addressed by compile-time indices.
//
//   * Only the candidates that can occur in a tuple at all are staged: spans of the cut-off window that lie inside
//     the incoming span (traceweaver_v3.py:328-333) and -- when a window is re-solved after earlier windows took
//     spans away (mode 1, traceweaver_v1.py:457-463) -- are still there.  Order is kept; lr[][] maps a staged
//     position back to the position in the cut-off window.  On the bench workloads two thirds of the raw grid points
//     fail containment; the enumeration only visits the grid of the staged candidates.
//   * Every score term is a table look-up: root(in.start -> s.start) and closing(s.end -> in.end) per staged
//     candidate, and the term of a primary call-order edge (p.end -> s.start, traceweaver_v1.py:345-347) per pair of
//     staged candidates (as long as the pair tables fit their LDS pool; edges that do not fit are evaluated per tuple).
//     A term is evaluated once per candidate (pair) instead of once per tuple -- with mixtures (pass 2) a term is
//     ~1.5k instructions and an 8-endpoint tuple has up to 9 of them.  The tuple score adds the same doubles in the
//     same order as the reference, so it is bit-identical.
// Registers of the wavefront kernel.  Measured and not adopted (profiles/r02d_variants.jsonl, profiles/r02d_spill_traffic.md):
// -DTW_HEAVY_ATTR='__attribute__((amdgpu_waves_per_eu(3)))' holds it to 168 VGPRs -- three wavefronts per SIMD instead of the two
// its 232 registers allow (E = 4) -- and the enumeration group does get 7 % faster (2.57 -> 2.39 ms per launch at 6.4 M spans,
// 8.70 -> 8.08 ms at 25.6 M), but the 256 B per lane it then spills are written back to HBM: 2.4 GB per launch at 25.6 M spans,
// five times everything else the group writes.  The registers have to go by restructuring (cold paths out of line), not by a cap.
// Round 5: the log-mode bookkeeping took the E = 4 instantiation from 256 to 257 registers -- one wavefront per SIMD instead of two,
// 1.09 -> 2.09 ms for the media shape's class (profiles/HISTORY.md).  Up to four endpoints the kernel is therefore held to two
// wavefronts per SIMD (256 registers: what it used before; the few bytes it spills stay in the scratch's L2 lines).
#ifndef TW_HEAVY_WAVES
#define TW_HEAVY_WAVES(E) ((E) <= 4 ? 2 : 1)
#endif
#ifndef TW_HEAVY_ATTR
#ifdef TW_HOST_EMULATION
#define TW_HEAVY_ATTR
#else
#define TW_HEAVY_ATTR __attribute__((amdgpu_waves_per_eu(TW_HEAVY_WAVES(E))))
#endif
#endif
constexpr int kPairPoolPerEp = 160;   // doubles of pair-term tables per endpoint (E <= 4); 2048 doubles (16 KB) for the deep call graphs
constexpr int kGridTarget = TW_GRID_TARGET;     // grid points per prefix the split of the endpoints aims at (a prefix step costs about as much as five grid batches)
// Deep call graphs with many candidates per endpoint (Alibaba shape: 7-8 endpoints x 10-14 candidates = 10^7-10^8 grid
// points for 10^4-10^5 feasible tuples): walking the prefixes one after the other and testing a dense grid below each
// leaves most lanes idle (0.1 % of the grid is feasible).  Such a span first lists its feasible tuples, level by level
// (see the kernel), in the reference's depth-first order, and then scores them a wavefront at a time.
#ifndef TW_FRONTIER_GRID_DEEP
#define TW_FRONTIER_GRID_DEEP 1024
#endif
// from this many grid points (staged candidates) on, and E >= 3.  The deep classes list much earlier: a batch of 64 grid points costs an
// eight-endpoint wavefront 3 us (it runs alone on its SIMD), and a grid of 32 400 points with 598 tuples held the class' kernel for
// 1.7 ms of walking (round 5) where the list of its tuples takes ten batches to score.
__host__ __device__ constexpr long long frontier_grid(int E) { return E >= 5 ? TW_FRONTIER_GRID_DEEP : 1 << 15; }
#define kFrontierGrid frontier_grid(E)
// Long enumerations are walked with an upper bound (see "bounds of the pruned walk" in the kernel): from this many grid points on
#ifndef TW_PRUNE_MIN
#define TW_PRUNE_MIN 1024
#define TW_PRUNE_GRID 128
#endif
constexpr long long kPruneMin = TW_PRUNE_MIN;
constexpr int kPruneGrid = TW_PRUNE_GRID;    // grid points per prefix the split of the endpoints aims at when the walk is pruned (finer prefixes: more to cut)
#ifndef TW_FRONTIER_CAP
#define TW_FRONTIER_CAP (1 << 15)
#define TW_FRONTIER_BIG_CAP (1 << 21)
#define TW_FRONTIER_BIG_SLOTS 48
#endif
// A listed enumeration with no more tuples than this is scored from its list (every tuple, a wavefront at a time); one with more only
// counts its last level and finds the five best by the pruned walk.  (Round 5: the walk's threshold is a score that five tuples reach --
// an enumeration of fewer than five tuples, or one whose bound is loose, walked every feasible prefix one after the other: single
// spans of the deep call graphs held their class' kernel for 3 ms with a list of a few thousand tuples at hand.)
#ifndef TW_LIST_SCORE_MAX
#define TW_LIST_SCORE_MAX 16384
#endif
constexpr long long kListScoreMax = TW_LIST_SCORE_MAX;
constexpr int kFrontierCap = TW_FRONTIER_CAP;          // prefixes per level and wavefront (two buffers of 8 B entries); beyond: a slot of the pool
constexpr int kFrontierBigCap = TW_FRONTIER_BIG_CAP;       // ... of kFrontierBigSlots lists this long (32 MB a slot); beyond, or none left: the walk
constexpr int kFrontierSlots = 4096;            // buffer pairs of kFrontierCap entries, claimed by the wavefronts that need one
constexpr int kFrontierBigSlots = TW_FRONTIER_BIG_SLOTS;   // (the host-emulation build of the tests uses tiny sizes so that all three routes are exercised)
// The tuple-list buffers are pools with a flag per buffer: claimed by a compare-and-swap (a few probes from a start that differs
// from wavefront to wavefront), given back when the wavefront (its own pair) or the span (a long list) is done -- so what a batch
// needs is bounded by the wavefronts resident at a time, not by how many spans of the batch ask (a bump counter that was only reset
// between the passes ran dry on large batches, and those spans fell back to the plain walk: 16 M resident spans of the deep
// call graphs were slower per span than 1 M).  One lane calls.
__device__ __forceinline__ int pool_acquire(int32_t* busy, int n, unsigned start) {
    for (int k = 0; k < 64 && k < n; k++) {
        const int s = (int)((start + (unsigned)k) % (unsigned)n);
        if (atomicCAS(&busy[s], 0, 1) == 0) return s;
    }
    return -1;
}
__device__ __forceinline__ void pool_release(int32_t* busy, int slot) {
    __threadfence();
    atomicExch(&busy[slot], 0);
}

// the work-list cursors of the wavefront kernels: one per (kind of launch, instantiation, class), all reset with the counter block of
// the pass / the repair round -- no memset between the launches of a class' chain
__device__ __forceinline__ int32_t* enum_cursor(const Dev& P, int part, bool wide, int E) {
    const int kind = part == 0 || part == 2 ? 0 : (part == 3 ? 1 : (part == 1 ? 2 : 3));
    return P.heavy_in_next + ((kind * 2 + (wide ? 1 : 0)) * (kMaxEp + 1) + E);
}

// What the tile kernel of class E has listed when the launch before this one has ended: the end of stretch `j - 1` of the class' lists
// (stretch 0 starts at 0: the snapshots are zeroed with the counter block of the pass)
__global__ void k_enum_snapshot(Dev P, int E, int j) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int32_t* snap = P.enum_snap + (E * (kEnumStretches + 1) + j) * 4;
    snap[0] = P.heavy_big_count[E]; snap[1] = P.heavy_in_count[E]; snap[2] = P.heavy_in_count[kMaxEp + 1 + E];
}

template <int E, int W>
__global__ void __launch_bounds__(kHeavyThreads) TW_HEAVY_ATTR k_enumerate_heavy(Dev P, int pass, int mode, int part_arg, int pool) {
    // part: 0 = the class' lists, the long enumerations first (one launch serves both); 1 = only the long ones (the split spans
    // that k_merge_parts lists again); 2 = only the others.  `pool` = doubles of dynamic LDS for the pair-term tables.
    // part_arg = part | (stretch + 1) << 8: a launch of part 0 that serves one stretch of the class' lists -- the entries the tile
    // kernel appended between two snapshots of the list counters (k_enum_snapshot; launch_enumerate runs such launches beside the
    // tile kernel of the class' following tiles instead of behind the last one)
    if (*P.err != 0) return;  // an earlier kernel of this pass reported an error: its outputs are not usable
    const int part = part_arg & 255, stretch = (part_arg >> 8) - 1;
    static_assert(W == kNarrow || W == 64 * kCandWords, "one instantiation per half of the work list");
    constexpr bool kWide = W != kNarrow;
    constexpr int kList = kWide ? kMaxEp + 1 + E : E;
    const int kPool = pool;
    __shared__ unsigned long long sbits[kMaxEp][kCandWords];
    __shared__ int64_t ls[E][W], le[E][W];      // staged candidates: start / end
    __shared__ double troot[E][W], tclose[E][W];
    HIP_DYNAMIC_SHARED(double, tpair)
    __shared__ int16_t tp_off[E][E];            // pair table of the j-th in-edge of endpoint e, -1: evaluated per tuple
    __shared__ uint8_t lr[E][W];                // position of the staged candidate in the cut-off window
    __shared__ Cand<E> sheap[kTopK + 1];        // CPython heap replay (degenerate ties only), lane 0
    __shared__ int32_t keep_idx[kTopK][E];      // staged positions of the kept entries otherwise
    __shared__ int32_t px[E];                   // the prefix the wavefront is walking (staged positions, same for every lane)
    __shared__ int64_t pxs[E], pxe[E];
    __shared__ double pxp[E];                   // ... and the sum of the score terms its levels 0..d determine (pruned walk)
    __shared__ double sbound[E + 1];            // upper bound of the terms the levels d.. can add (pruned walk)
    __shared__ int32_t defer_lo[kMaxParts + 1]; // first listed prefix of every list part (deferred spans)
    const int t = threadIdx.x, nt = blockDim.x;
    // part 3: the list parts of the deferred spans (see kListSplitFlag), appended behind the class' listed entries by the launch before
    // part 4: what k_enumerate_lean left to this kernel (P.fb_*: entries in the format of the list of long enumerations)
    const int32_t* snap = stretch >= 0 ? P.enum_snap + (E * (kEnumStretches + 1) + stretch) * 4 : nullptr;   // [0] long enumerations, [1] narrow, [2] wide entries listed so far
    const int big_lo = stretch >= 0 ? snap[0] : 0, in_lo = stretch >= 0 ? snap[kWide ? 2 : 1] : 0;
    const int defer_base = part == 3 ? P.heavy_big_count[E] : big_lo;
    const int n_big = stretch >= 0 ? snap[4] - big_lo
                                   : (part == 2 ? 0 : (part == 3 ? P.defer_count[E] : (part == 1 ? P.redo_count[E] : (part == 4 ? P.fb_count[E] : P.heavy_big_count[E]))));   // (both instantiations walk the list of long enumerations; each takes its own)
    const int count = n_big + (stretch >= 0 ? snap[4 + (kWide ? 2 : 1)] - in_lo : ((part == 0 || part == 2) ? P.heavy_in_count[kList] : 0));
    int32_t* next_counter = stretch >= 0 ? P.enum_stretch_next + (E * kEnumStretches + stretch) * 2 + (kWide ? 1 : 0)
                                         : enum_cursor(P, part, kWide, E);   // (a cursor per kind of launch: nothing to reset between the launches of a class)
    const int32_t *big_unit = part == 4 ? P.fb_unit : P.heavy_big_unit, *big_idx = part == 4 ? P.fb_idx : P.heavy_big_idx;
    const int32_t *big_part = part == 4 ? P.fb_part : P.heavy_big_part, *big_slot_of = part == 4 ? P.fb_slot : P.heavy_big_slot;
    const int nstatic = (int)gridDim.x * kWorkChunk;
    if ((int)blockIdx.x >= count) return;   // nothing for this wavefront (its strided static items start at its block index)
    int chunk_pos = 0, chunk_end = 0;
    int front_slot = -1;   // this wavefront's pair of tuple-list buffers: -1 not claimed yet, -2 none left
    TW_PROF_DECL();
    bool first_chunk = true;
    while (true) {
        // dynamic work distribution: candidate products span four orders of magnitude, a static split leaves
        // most wavefronts idle behind the few that drew the large spans.  The first chunk of a wavefront is its own
        // (no atomic: thousands of same-address atomics at kernel start serialise at ~20 ns each); later chunks come
        // from the counter, which starts behind the static ones.
        if (chunk_pos == chunk_end) {
            if (first_chunk) { chunk_pos = (int)blockIdx.x * kWorkChunk; first_chunk = false; }
            else {
                if (t == 0) chunk_pos = nstatic + atomicAdd(next_counter, kWorkChunk);
                chunk_pos = __shfl(chunk_pos, 0);
            }
            const int limit = chunk_pos < nstatic ? nstatic : count;   // static chunks are strided over the wavefronts below
            chunk_end = chunk_pos + kWorkChunk < limit ? chunk_pos + kWorkChunk : limit;
            if (chunk_pos >= count && chunk_pos >= nstatic) { TW_PROF_FLUSH(); break; }
        }
        int item = chunk_pos++;
        if (item < nstatic) item = (item % kWorkChunk) * (int)gridDim.x + item / kWorkChunk;   // the long spans at the front: one to a wavefront
        if (item >= count) continue;
        // wave-uniform by construction; telling the compiler so turns every access to the unit descriptor below
        // into a scalar load (SGPRs, constant cache) instead of 64 lanes loading the same address
        const bool from_big = item < n_big;
        const int pos = from_big ? P.heavy_big_off[E] + defer_base + item : (kWide ? P.heavy_in_off[E + 1] - 1 - (in_lo + item - n_big) : P.heavy_in_off[E] + in_lo + (item - n_big));
        const int unit = __builtin_amdgcn_readfirstlane((from_big ? big_unit : P.heavy_in_unit)[pos]);
        const int i_raw = __builtin_amdgcn_readfirstlane((from_big ? big_idx : P.heavy_in_idx)[pos]);
        const int i = i_raw & ~kIdxReplayFlag;
        // a part of a split enumeration?  (number of parts, which one, where its result goes)
        const int part_info = from_big ? __builtin_amdgcn_readfirstlane(big_part[pos]) : 1;
        if (from_big && (((part_info >> 24) & 1) != 0) != kWide) continue;   // the other instantiation's
        const int nparts = part_info & 255, part_no = (part_info >> 8) & 0xffff;
        const bool part_log = nparts > 1 && (part_info & kPartLogFlag) != 0;   // log mode: this part replays CPython's heap and logs what entered it
        const bool replay_first = nparts == 1 && (from_big ? (part_info & kReplayFlag) != 0 : (i_raw & kIdxReplayFlag) != 0);   // twin candidates: see kReplayFlag
        int nlog = 0;                                                         // (lane 0's count)
        const bool list_part = nparts > 1 && (part_info & kListSplitFlag) != 0;   // its tuples: a stretch of the span's listed prefixes (P.part_lo / part_hi)
        bool part_failed = false, deferred = false;
        const int part_slot = from_big ? __builtin_amdgcn_readfirstlane(big_slot_of[pos]) : 0;
        const UnitDev& U = P.units[unit];
        TW_ITEM_BEGIN();
        const int64_t in_start = P.in_start[U.in_off + i], in_end = P.in_end[U.in_off + i];
        Scorer S;
        S.pass = pass;
        S.gp = P.gparam + (U.gp_off + (int64_t)(i / P.batch_size) * U.nslot) * 4;
        S.mix_n = P.mix_n + U.slot_off;
        S.mix_c = P.mix_c + (int64_t)U.slot_off * kMaxComp * 4;
        // cut-offs (FindCutoffs, traceweaver_v3.py:182-217) were computed by k_enumerate_tile in pass 1
        int32_t lo[E], cn[E];   // first span of the cut-off window; number of staged candidates
        // the unit's call-order DAG in registers: predecessor masks, predecessor counts, and per endpoint the
        // predecessor list in in_edges() order packed 4 bits each (index | primary << 3)
        uint32_t dag_pm[E], dag_np[E], dag_pl[E];
#pragma unroll
        for (int e = 0; e < E; e++) {
            dag_pm[e] = U.pred_mask[e];
            dag_np[e] = U.npred[e];
            uint32_t pk = 0;
#pragma unroll
            for (int j = 0; j < E; j++) pk |= (j < (int)dag_np[e] ? ((uint32_t)U.pred_list[e][j] | ((uint32_t)U.pred_prim[e][j] << 3)) : 0u) << (4 * j);
            dag_pl[e] = pk;
        }
        for (int k = t; k < kMaxEp * kCandWords; k += nt) (&sbits[0][0])[k] = 0;
        bool none = false;
        // stage the candidates that can occur in a tuple, order kept (ballot prefix sums).  The loads come first, all of
        // them -- cut-offs of every endpoint, then the first chunk of every window: one memory round trip per step for the
        // span instead of one per endpoint (the phase is latency-bound: profiles/tools/profile_enumerate.py)
        int32_t wd[E];
#pragma unroll
        for (int e = 0; e < E; e++) { lo[e] = P.c_lo[ie_index(U, e, i)]; wd[e] = P.c_hi[ie_index(U, e, i)]; }
        int64_t st0[E], en0[E];
#pragma unroll
        for (int e = 0; e < E; e++) {
            wd[e] = wd[e] - lo[e] + 1;
            st0[e] = 0; en0[e] = 0;
            if (t < wd[e]) { st0[e] = P.out_start[U.ep_off[e] + lo[e] + t]; en0[e] = P.out_end[U.ep_off[e] + lo[e] + t]; }
        }
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int w = wd[e];
            const int64_t* os = P.out_start + U.ep_off[e];
            const int64_t* oe = P.out_end + U.ep_off[e];
            const uint64_t* gone = mode == 1 ? P.gone + ie_index(U, e, i) * kCandWords : nullptr;
            int c = 0;
            for (int r0 = 0; r0 < w; r0 += nt) {
                const int r = r0 + t;
                int64_t st = st0[e], e2 = en0[e];
                bool inside = false;
                if (r < w) {
                    if (r0 > 0) { st = os[lo[e] + r]; e2 = oe[lo[e] + r]; }
                    inside = !(in_start > st || e2 > in_end);
                    if (inside && gone != nullptr) inside = !((gone[r >> 6] >> (r & 63)) & 1ull);
                }
                const unsigned long long m = __ballot(inside);
                if (inside) {
                    const int q = c + __popcll(m & ((1ull << t) - 1ull));
                    ls[e][q] = st; le[e][q] = e2; lr[e][q] = (uint8_t)r;
                }
                c += __popcll(m);
            }
            cn[e] = c;
            none |= c == 0;
        }
        wave_sync();
        TW_PHASE(0);
        int64_t leaves = 0;
        int nout = 0, big_slot = -1;   // (big_slot: a long tuple list of the pool, given back when the span is done)
        int c0_begin = 0, c0_end = cn[0];   // the staged candidates of the first endpoint this wavefront enumerates
        if (nparts > 1 && !list_part) { c0_begin = (int)((long long)part_no * cn[0] / nparts); c0_end = (int)((long long)(part_no + 1) * cn[0] / nparts); none |= c0_begin == c0_end; }
        bool part_ambiguous = false;
        if (!none) {
        {   // one (candidate span, root | closing) term per lane, all endpoints at once
            int wsum = 0;
#pragma unroll
            for (int e = 0; e < E; e++) wsum += cn[e];
            for (int q = t; q < 2 * wsum; q += nt) {
                int es = 0, r = q >> 1;
                bool found = false;
#pragma unroll
                for (int e = 0; e < E; e++)
                    if (!found) { if (r < cn[e]) { es = e; found = true; } else r -= cn[e]; }
                const int64_t st = ls[es][r], e2 = le[es][r];
                if (q & 1) tclose[es][r] = score_term(S, slot_close(E, es), e2, in_end);
                else {
                    bool is_root = false;
#pragma unroll
                    for (int e = 0; e < E; e++) if (e == es) is_root = dag_np[e] == 0;
                    troot[es][r] = is_root ? score_term(S, slot_root(E, es), in_start, st) : 0.0;
                }
            }
        }
        if constexpr (E > 1) {   // pair tables of the primary in-edges, in scoring order, while the pool lasts
            int used = 0;
#pragma unroll
            for (int e = 0; e < E; e++) {
#pragma unroll
                for (int j = 0; j < E; j++) {
                    if (j >= (int)dag_np[e]) continue;
                    const uint32_t pj = (dag_pl[e] >> (4 * j)) & 15u;
                    if (!(pj & 8u)) continue;
                    const int p = (int)(pj & 7u);
                    int cp = 0;
#pragma unroll
                    for (int q = 0; q < E; q++) if (q == p) cp = cn[q];
                    const int need = cp * cn[e];
                    const bool fits = used + need <= kPool;
                    if (t == 0) tp_off[e][j] = (int16_t)(fits ? used : -1);
                    if (fits) {
                        for (int q = t; q < need; q += nt) {
                            const int a = q / cn[e], b2 = q % cn[e];
                            const int64_t pend = le[p][a], st = ls[e][b2];
                            if (pend <= st) tpair[used + q] = score_term(S, slot_prim(E, p, e), pend, st);  // other pairs never occur in a tuple
                        }
                        used += need;
                    }
                }
            }
        }
        // ---- bounds of the pruned walk.  A tuple's score is a sum of tabulated terms: per endpoint the root term or the terms of
        // its primary in-edges (fixed once the endpoint and its predecessors are chosen), plus ONE closing term (of whichever
        // span ends last).  For a prefix x_0..x_d the real number  P_d + sbound[d+1] + max closing term,  P_d = the terms the
        // prefix determines, sbound[d+1] = per remaining endpoint the largest value its terms can take, bounds the exact
        // score of every tuple below the prefix.  The scores themselves are binary64 sums in the reference's order; they differ
        // from the exact sums by at most (2E + 1) ulp-sized errors of the partial sums -- `margin` (1e-12 of the sum of the
        // largest |term| per position) covers both that and the rounding of the bound, with three orders of magnitude to spare.
        // A prefix is cut when bound + margin < threshold STRICTLY, threshold = a score that five distinct tuples are known to
        // reach: the fifth kept score, or thr0 -- the fifth largest score of up to 64 distinct tuples built greedily from the
        // best two candidates per endpoint before the walk starts.  Every tuple that reaches the final fifth score is still
        // visited (thresholds only rise towards it), so the kept five and the check for undecided ties see what the full walk
        // sees.  In the replay of CPython's heap (second attempt) the threshold is the heap's current minimum: a tuple strictly
        // below it is a no-op there (see the batch loop), so skipping it changes nothing either.
        // What the cut tuples would have added to the tuple count and the candidate bitmap must be known without visiting them:
        // without call-order constraints every grid point is a tuple (closed form); in pass 2 the first solve has the same
        // tuples as in pass 1 (leaves0); the long enumerations of deep call graphs (those that list their tuples, below) list only
        // the prefixes of E - 1 endpoints and count the last level: tuple count and candidate bitmap without a single score.
        // Otherwise the walk is not pruned.
        bool prune = false, tables_ok = false, counted = false, list_all = false, can_defer = false;
        long long leaves_counted = 0;
        double mclose = -dinf(), margin = 0.0, thr0 = -dinf();
        long long tail[E];   // tuples below a prefix that ends at level d when every grid point is one: the product of the later counts
        uint32_t any_order = 0;
        {
            long long grid = 1;
#pragma unroll
            for (int e = E - 1; e >= 0; e--) { tail[e] = grid; grid = grid < (1ll << 40) ? grid * cn[e] : grid; any_order |= dag_pm[e]; }
            // pass 2 knows the tuple count of pass 1: a long enumeration of few tuples is listed and scored whole (list_all), not walked
            // (a class that defers its long spans lists and counts them in pass 2 as well: the count decides who scores them)
            can_defer = E >= 3 && E >= P.defer_min_e && part == 0 && mode == 0 && nparts == 1 && !U.skip && any_order != 0 && (!replay_first || P.split_twins == 2);
            bool count_again = false;
            if (E >= 3 && pass == 2 && mode == 0 && any_order != 0 && grid >= kFrontierGrid && !U.skip) {
                const long long n0 = P.leaves0[U.in_off + i];
                list_all = n0 <= (can_defer ? kDeferTuples : kListScoreMax);
                count_again = can_defer && !list_all;
            }
            const bool countable = any_order == 0 || (pass == 2 && mode == 0 && !list_all && !count_again);   // (otherwise, for the long enumerations of deep call graphs: counted by listing prefixes, below)
            if (E >= 2 && grid >= kPruneMin && (countable || (E >= 3 && grid >= kFrontierGrid)) && !U.skip && !list_part) {
                auto wave_max = [&](double v) -> double {
                    for (int off = 32; off >= 1; off >>= 1) { const double o = __shfl_xor(v, off); v = o > v ? o : v; }
                    return v;
                };
                bool tables = true;
                double aabs = 0.0, aclose = 0.0, ube[E];
#pragma unroll
                for (int e = 0; e < E; e++) {
                    double m = -dinf(), a = 0.0, mc = -dinf(), ac = 0.0;
                    for (int c = t; c < cn[e]; c += nt) {
                        const double vr = troot[e][c], vc = tclose[e][c];
                        m = vr > m ? vr : m; a = fabs(vr) > a ? fabs(vr) : a;
                        mc = vc > mc ? vc : mc; ac = fabs(vc) > ac ? fabs(vc) : ac;
                    }
                    m = wave_max(m); a = wave_max(a); mc = wave_max(mc); ac = wave_max(ac);
                    mclose = mc > mclose ? mc : mclose;
                    aclose = ac > aclose ? ac : aclose;   // (one closing term per tuple: its largest magnitude counts once)
                    ube[e] = dag_np[e] == 0 ? m : 0.0;
                    double ae = dag_np[e] == 0 ? a : 0.0;
#pragma unroll
                    for (int j = 0; j < E; j++) {
                        if (j >= (int)dag_np[e]) continue;
                        const uint32_t pj = (dag_pl[e] >> (4 * j)) & 15u;
                        if (!(pj & 8u)) continue;
                        const int p = (int)(pj & 7u);
                        const int off = (int)tp_off[e][j];
                        if (off < 0) { tables = false; continue; }
                        int cp = 0;
#pragma unroll
                        for (int q = 0; q < E; q++) if (q == p) cp = cn[q];
                        double mp = -dinf(), ap = 0.0;
                        for (int q = t; q < cp * cn[e]; q += nt) {
                            const int a2 = q / cn[e], b2 = q % cn[e];
                            if (le[p][a2] <= ls[e][b2]) { const double v = tpair[off + q]; mp = v > mp ? v : mp; ap = fabs(v) > ap ? fabs(v) : ap; }
                        }
                        ube[e] += wave_max(mp);
                        ae += wave_max(ap);
                    }
                    aabs += ae;
                }
                aabs += aclose;
                double sb = 0.0;
                if (t == 0) sbound[E] = 0.0;
#pragma unroll
                for (int e = E - 1; e >= 0; e--) { sb += ube[e]; if (t == 0) sbound[e] = sb; }
                margin = aabs * 1.0e-12 + 1.0e-300;
                tables_ok = tables && aabs < dinf() && aabs == aabs;
                prune = tables_ok && countable;
            }
        }
        wave_sync();
        // Python orders (score, [spans]) tuples by score, then by start_mus of the first differing span; two
        // tuples whose first differing spans start at the same time are "equivalent" (neither is less), and
        // only then does the outcome of heapq / list.sort depend on the order of the pushes.  First attempt:
        // keep the five largest tuples under the strict part of that order in registers (every lane the same
        // values) and watch for an equivalence that could matter (between the candidate, the evicted entry
        // or the kept entries).  If one shows up -- millisecond-granular data -- the span is redone with the
        // CPython heap replayed push by push in LDS (second attempt).
        // the feasible tuples as a list (see kFrontierGrid), 8 bits per staged position, built level by level: a prefix may
        // be followed by exactly the candidates that start no earlier than its latest predecessor ends -- the staged
        // candidates are in start order, so that is a tail of the list, found by bisection: one lane per prefix, the
        // children written behind one another (wavefront prefix sum of the counts), depth-first order kept
        TW_PHASE(1);
        bool use_front = false, list_scored = false;
        int n_front = 0;
        const unsigned long long* front = nullptr;
        if constexpr (E >= 3) {
            long long grid = 1;   // (without call-order constraints every grid point is a tuple: nothing to gain from listing them;
                                  // a pruned walk -- pass 2: the tuple count is known -- visits far fewer tuples than the list holds)
#pragma unroll
            for (int e = 0; e < E; e++) grid = grid < (1ll << 40) ? grid * cn[e] : grid;
            if (prune) grid = 0;
            // count_only: the prefixes of E - 1 endpoints are listed, their tuples counted and their candidates marked -- the pruned
            // walk then finds the five best without scoring the rest
            const bool count_only = tables_ok && !prune && !list_all && !list_part;
            int minlo = 0x7fffffff;
            const bool listed = (grid >= kFrontierGrid && any_order != 0 && !U.skip) || list_part;
            int defer_np = 0, defer_slot = 0, defer_at = 0, defer_level = 0;
            bool defer_by_tuples = false;
            // the class' budget of extra list entries and the arena: room for np parts and n list entries?  (lane 0 asks, all lanes hear)
            auto defer_reserve = [&](int np, int n) -> bool {
                int got = -1, at = 0;
                if (t == 0 && np >= 2) {
                    // (a full arena is not asked again: the counter stays below 2^31 however many spans of a very large batch would like
                    // to defer -- what can be added beyond the cap is bounded by the lists resident at a time.  The arena first: what it
                    // hands out is never given back, the class' budget is only taken when the list has its room)
                    if (atomicAdd(&P.part_used[E], 0) + np <= (P.part_off[E + 1] - P.part_off[E]) / 2 && atomicAdd(P.defer_used, 0) < P.defer_cap) {
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
            if (listed && front_slot < 0 && front_slot != -2) {
                // the first such span of this wavefront claims one of the kFrontierSlots buffer pairs (kernels of several
                // classes run side by side: the block index does not identify a wavefront across them)
                if (t == 0) front_slot = pool_acquire(P.frontier_busy, kFrontierSlots, ((unsigned)blockIdx.x * 40503u + (unsigned)(E * 2 + (kWide ? 1 : 0)) * 7919u + (unsigned)part * 104729u));
                front_slot = __shfl(front_slot, 0);
                if (front_slot < 0) front_slot = -2;   // none free: this wavefront walks
            }
            if (listed && front_slot >= 0) {
                unsigned long long* fa = P.frontier + (size_t)front_slot * 2 * kFrontierCap;
                int cap = kFrontierCap;
                const int lane = t & 63;
                for (int tries = 0; tries < 2; tries++) {
                    unsigned long long* fb = fa + cap;
                    int nprev = c0_end - c0_begin;
                    int part_lvl = 0;
                    if (list_part) {   // the stretch of the span's listed prefixes of level part_lvl: the levels below are left to list (into the own buffers)
                        const int p_lo = P.part_lo[part_slot], p_hi = P.part_hi[part_slot];
                        part_lvl = P.part_lvl[part_slot];
                        fb = fa; fa = P.defer_list + p_lo; nprev = p_hi - p_lo;
                    } else
                        for (int c = c0_begin + t; c < c0_end; c += nt) fa[c - c0_begin] = (unsigned long long)c;
                    use_front = true;
                    leaves_counted = 0;
                    __threadfence_block();
                    wave_sync();
#pragma unroll
                    for (int d = 1; d < E; d++) {
                        if (!use_front || deferred) continue;
                        if (list_part && d <= part_lvl) continue;
                        const int cd = cn[d];
                        const bool count_level = count_only && d == E - 1;
                        int nnext = 0;
                        // the last level of a counted enumeration: counted first (rep 0) and -- only where the count is small -- written
                        // after all (rep 1: the enumeration is scored from its list, see kListScoreMax); every other level is written
                        for (int rep = count_level ? 0 : 1; rep < 2; rep++) {
                        const bool last_count = rep == 0;
                        if (rep == 1 && count_level) {
                            if (can_defer && leaves_counted > kDeferTuples) {   // many tuples: cut into list parts, if the class' budget and the arena allow
                                const long long want = (leaves_counted + kSplitTuples - 1) / kSplitTuples;
                                if (defer_reserve((int)(want > kMaxParts ? kMaxParts : want), nprev)) { deferred = true; defer_by_tuples = true; defer_level = E - 2; break; }
                            }
                            if (leaves_counted > kListScoreMax || leaves_counted > cap) break;
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
#pragma unroll
                            for (int q = 0; q < E - 1; q++)
                                if (q < d && ((dag_pm[d] >> q) & 1)) { const int64_t en = le[q][(ent >> (8 * q)) & 255ull]; tmin = en > tmin ? en : tmin; }
                            int lo2 = 0, hi2 = cd;
                            while (lo2 < hi2) { const int mid = (lo2 + hi2) >> 1; if (ls[d][mid] < tmin) lo2 = mid + 1; else hi2 = mid; }
                            const int cnt = valid ? cd - lo2 : 0;
                            int incl = cnt;
                            for (int off = 1; off < 64; off <<= 1) { const int v = __shfl(incl, lane >= off ? lane - off : lane); if (lane >= off) incl += v; }
                            const int total = __shfl(incl, nt - 1 < 63 ? nt - 1 : 63);
                            if (last_count) {   // the tuples below every prefix: counted, their spans marked (not written, not scored)
                                leaves_counted += total;
                                if (cnt > 0 && pass == 1 && mode == 0) {
#pragma unroll
                                    for (int q = 0; q < E - 1; q++) {
                                        const int r = lr[q][(ent >> (8 * q)) & 255ull];
                                        const unsigned long long bit = 1ull << (r & 63);
                                        if (!(sbits[q][r >> 6] & bit)) atomicOr(&sbits[q][r >> 6], bit);
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
                        if (nnext > cap) use_front = false;
                        if (!(count_level && !list_scored)) {   // (a level that was only counted leaves the list of its prefixes where it is)
                            unsigned long long* sw = fa; fa = fb; fb = sw;
                            if (list_part && d == part_lvl + 1) fb = fa + cap;   // (a list part's first level came from the arena: on between its own two buffers)
                            nprev = nnext;
                            // a long list before the last level: the span is cut here, its parts list the rest (see kDeferPrefixes)
                            if (can_defer && use_front && d < E - 1 && nprev >= kDeferPrefixes) {
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
                        // The list parts.  Prefix f begins at tuple s_f (running sum of the children counts): it belongs to part
                        // floor(s_f np / total); the first prefix of every part is where that number changes (parts that no prefix
                        // begins in stay empty).  The list is copied to the arena on the way.
                        const int cd = cn[E - 1];
                        for (int q = t; q <= kMaxParts; q += nt) defer_lo[q] = defer_by_tuples ? nprev : (int)((long long)q * nprev / defer_np);
                        wave_sync();
                        long long running = 0;
                        int carry_pid = -1;
                        if (!defer_by_tuples)   // stretches of equal numbers of prefixes: only the copy
                            for (int f = t; f < nprev; f += nt) P.defer_list[(int64_t)defer_at + f] = fa[f];
                        for (int base = 0; defer_by_tuples && base < nprev; base += nt) {
                            const int f = base + t;
                            const bool valid = f < nprev;
                            const unsigned long long ent = valid ? fa[f] : 0ull;
                            int64_t tmin = INT64_MIN;
#pragma unroll
                            for (int q = 0; q < E - 1; q++)
                                if ((dag_pm[E - 1] >> q) & 1) { const int64_t en = le[q][(ent >> (8 * q)) & 255ull]; tmin = en > tmin ? en : tmin; }
                            int lo2 = 0, hi2 = cd;
                            while (lo2 < hi2) { const int mid = (lo2 + hi2) >> 1; if (ls[E - 1][mid] < tmin) lo2 = mid + 1; else hi2 = mid; }
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
                                for (int q = prev_pid + 1; q <= pid; q++) defer_lo[q] = f;
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
                            P.part_lo[defer_slot + q] = defer_at + defer_lo[q];
                            P.part_hi[defer_slot + q] = defer_at + (q + 1 < defer_np ? defer_lo[q + 1] : nprev);
                            P.part_lvl[defer_slot + q] = defer_level;
                            P.heavy_big_unit[ebase + q] = unit; P.heavy_big_idx[ebase + q] = i;
                            P.heavy_big_part[ebase + q] = defer_np | (q << 8) | flags; P.heavy_big_slot[ebase + q] = defer_slot + q;
                        }
                        wave_sync();
                        use_front = false;
                        break;
                    }
                    if (count_only && use_front && !list_scored) {   // counted: on to the pruned walk
                        for (int off = 32; off >= 1; off >>= 1) { const int o = __shfl_xor(minlo, off); minlo = o < minlo ? o : minlo; }
                        if (pass == 1 && mode == 0)   // (the admissible candidates of the last endpoint are tails of its list: their union is one)
                            for (int c = t; c < cn[E - 1]; c += nt) if (c >= minlo) { const int r = lr[E - 1][c]; atomicOr(&sbits[E - 1][r >> 6], 1ull << (r & 63)); }
                        counted = true; prune = true; use_front = false;
                        break;
                    }
                    if (use_front || tries == 1) break;
                    // the lists outgrew the wavefront's own buffers: once more in a slot of the pool, if one is left
                    int slot = 0;
                    if (t == 0) slot = pool_acquire(P.frontier_big_busy, P.frontier_big_slots, (unsigned)blockIdx.x * 40503u + (unsigned)i * 7919u + (unsigned)E);
                    slot = __shfl(slot, 0);
                    if (slot < 0) break;
                    big_slot = slot;
                    fa = P.frontier_big + (size_t)slot * 2 * kFrontierBigCap;
                    cap = kFrontierBigCap;
                }
            }
            if (list_part && !use_front) part_failed = true;   // (no buffer for its tuples: the span is enumerated again as a whole)
        }
#ifdef TW_ENUM_TRACE
        if (t == 0 && E >= 3 && (use_front || counted || deferred || part_failed))
            printf("enumerate_heavy<%d>: pass %d mode %d span %d part %d/%d: %s, %lld tuples\n", E, pass, mode, i, part_no, nparts,
                   deferred ? "counted, deferred to list parts" : part_failed ? "list part without a buffer" : list_part ? "list part, scored" : list_scored ? "counted, then scored from its list" : list_all ? "listed (count of pass 1) and scored" : counted ? "counted, pruned walk" : "listed and scored",
                   (counted || deferred) ? leaves_counted : (long long)n_front);
#endif
        LdsHeap<E, W> hp;
        hp.heap = sheap; hp.nheap = 0; hp.ls = ls;
        wave_sync();
        if (prune) {
            // thr0: up to 64 distinct tuples, lane t taking at endpoint e (in call order) the best -- or, where bit e of t is set,
            // the second best -- admissible candidate by the terms the endpoint adds given the lane's earlier choices; their scores in
            // the reference's order of additions; the fifth largest.  (Two lanes differ at the first endpoint where their bits
            // differ, so the tuples are distinct; a lane that finds no candidate drops out.)
            constexpr int NB = E < 6 ? E : 6;
            bool valid = t < (1 << NB);
            int32_t x[E];
            int64_t xs[E], xe[E];
#pragma unroll
            for (int e = 0; e < E; e++) {
                double best = -dinf(), second = -dinf();
                int bi = -1, si = -1;
                for (int c = 0; c < cn[e]; c++) {
                    const int64_t st = ls[e][c];
                    bool ok = valid;
                    double v = dag_np[e] == 0 ? troot[e][c] : 0.0;
#pragma unroll
                    for (int j = 0; j < E; j++) {
                        if (j >= (int)dag_np[e]) continue;
                        const uint32_t pj = (dag_pl[e] >> (4 * j)) & 15u;
                        const int p = (int)(pj & 7u);
                        int64_t pend = 0;
                        int xp = 0;
#pragma unroll
                        for (int q = 0; q < E; q++) if (q == p) { pend = xe[q]; xp = x[q]; }
                        if (pend > st) ok = false;
                        else if (pj & 8u) v += tpair[(int)tp_off[e][j] + xp * cn[e] + c];
                    }
                    if (ok) {
                        if (v > best) { second = best; si = bi; best = v; bi = c; }
                        else if (v > second) { second = v; si = c; }
                    }
                }
                const int pick = (e < NB && ((t >> e) & 1)) ? si : bi;
                if (pick < 0) valid = false;
                x[e] = pick < 0 ? 0 : pick;
                xs[e] = ls[e][x[e]]; xe[e] = le[e][x[e]];
            }
            double score = 0.0;
            if (valid) {   // ScoreAssignmentAsPerInvocationGraph (traceweaver_v1.py:305-361) from the term tables, as in the walk below
                int last = 0;
                int64_t last_end = xe[0];
#pragma unroll
                for (int e = 1; e < E; e++) if (xe[e] > last_end) { last_end = xe[e]; last = e; }
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const int np = (int)dag_np[e];
#pragma unroll
                    for (int j = 0; j < E; j++) {
                        if (j >= np) continue;
                        const uint32_t pj = (dag_pl[e] >> (4 * j)) & 15u;
                        if (!(pj & 8u)) continue;
                        const int p = (int)(pj & 7u);
                        int xp = 0;
#pragma unroll
                        for (int q = 0; q < E; q++) if (q == p) xp = x[q];
                        score += tpair[(int)tp_off[e][j] + xp * cn[e] + x[e]];
                    }
                    if (np == 0) score += troot[e][x[e]];
                    if (e == last) score += tclose[e][x[e]];
                }
            }
            double rest = valid ? score : -dinf();
            const int nvalid = __popcll(__ballot(valid));
            for (int r = 0; r < kTopK; r++) {
                double m = rest;
                for (int off = 32; off >= 1; off >>= 1) { const double o = __shfl_xor(m, off); m = o > m ? o : m; }
                thr0 = m;
                const unsigned long long at = __ballot(valid && rest == m);
                if (at != 0ull && t == __ffsll((long long)at) - 1) rest = -dinf();
            }
            if (nvalid < kTopK) thr0 = -dinf();
        }
        TW_PHASE(2);
        // candidate spans that occur in a feasible tuple (pass 1): every lane collects its own in registers and adds them
        // to the bitmap in LDS once per span -- one same-address LDS atomic per lane, endpoint and BATCH was a fifth of the walk
        constexpr int kBitWords = (W + 63) / 64;
        unsigned long long mybits[E][kBitWords];
#pragma unroll
        for (int e = 0; e < E; e++)
#pragma unroll
            for (int w = 0; w < kBitWords; w++) mybits[e][w] = 0ull;
        double ts[kTopK];
        int tslot[kTopK], nk = 0;
        bool exact_replay = false, ambiguous = false;
        for (int attempt = (deferred || part_failed) ? 2 : (part_log || replay_first) ? 1 : 0; attempt < 2; attempt++) {
        exact_replay = attempt == 1;
        hp.nheap = 0; nk = 0; leaves = 0;
#pragma unroll
        for (int k = 0; k < kTopK; k++) { ts[k] = -dinf(); tslot[k] = k; }
        // Split the endpoints at L: levels 0..L-1 are walked together (every lane the same prefix), the
        // tuples of levels L..E-1 -- a grid of G = prod cn_e points, in enumeration order -- are spread over
        // the lanes.  L is the deepest split that gives a prefix about kGridTarget grid points (< kGridTarget * 128).
        int L = E - 1;
        int G = cn[E - 1];
        const int gtarget = prune ? kPruneGrid : kGridTarget;
#pragma unroll
        for (int e = E - 2; e >= 0; e--)
            if (L == e + 1 && G * cn[e] <= gtarget * 2 && G < gtarget) { L = e; G *= cn[e]; }
        if (G < kHeavyThreads) {   // at least a wavefront of grid points whenever the product allows it
#pragma unroll
            for (int e = E - 2; e >= 0; e--)
                if (L == e + 1 && G < kHeavyThreads && (!prune || e >= 1)) { L = e; G *= cn[e]; }   // (a pruned walk keeps at least one level to cut at)
        }
        if (use_front) { L = E; G = 1; }   // the tuples are listed: no walk, no grid
        // a part walks its share of the first level; if that level is inside the grid (short enumerations only: never with the
        // production thresholds) the first part takes everything and the others nothing
        int w0_begin = c0_begin, w0_end = c0_end;
        if (nparts > 1 && L == 0) { w0_begin = 0; w0_end = part_no == 0 ? cn[0] : 0; }
        // grid point -> staged positions of the levels L..E-1, last endpoint fastest: divisions by the wave-uniform
        // counts as multiplications (exact for g * (c - 1) < 2^32)
        uint32_t magic[E];
#pragma unroll
        for (int e = 0; e < E; e++) magic[e] = (uint32_t)((0x100000000ull + (unsigned)cn[e] - 1ull) / (unsigned)cn[e]);
        auto grid_digits = [&](int g, int32_t (&x)[E]) {
            uint32_t rest = (uint32_t)g;   // (listed tuples are read from the list instead, see the batch loop)
#pragma unroll
            for (int e = E - 1; e >= 0; e--) {
                if (e >= L) {
                    const uint32_t q = cn[e] == 1 ? rest : __umulhi(rest, magic[e]);
                    x[e] = (int32_t)(rest - q * (uint32_t)cn[e]);
                    rest = q;
                } else x[e] = px[e];
            }
        };
        // order of the candidate tuple (prefix px[0..L) + grid point gj) against the kept tuple in LDS slot sl
        // when their scores are equal: +1 candidate greater, -1 smaller, 0 equivalent
        auto tie_order = [&](const int32_t (&ci)[E], int sl) -> int {
#pragma unroll
            for (int e = 0; e < E; e++) {
                const int32_t ki = keep_idx[sl][e];
                if (ci[e] != ki) {
                    const int64_t a = ls[e][ci[e]], b2 = ls[e][ki];
                    return a > b2 ? 1 : (a < b2 ? -1 : 0);
                }
            }
            return 0;
        };
        auto slot_order = [&](int sa, int sb) -> int {  // same for two kept tuples
#pragma unroll
            for (int e = 0; e < E; e++) {
                const int32_t ia = keep_idx[sa][e], ib = keep_idx[sb][e];
                if (ia != ib) {
                    const int64_t a = ls[e][ia], b2 = ls[e][ib];
                    return a > b2 ? 1 : (a < b2 ? -1 : 0);
                }
            }
            return 0;
        };
        auto cd_full = [&](int dd) -> int {
            int v = 0;
#pragma unroll
            for (int e = 0; e < E; e++) if (e == dd) v = cn[e];
            return v;
        };
        int d = 0;
        if (L > 0 && t == 0) px[0] = w0_begin - 1;
        wave_sync();
        const bool once = (L == 0) || use_front;
        const int Gtot = use_front ? n_front : (nparts > 1 && L == 0 && w0_end == 0 ? 0 : G);
        while (once || d >= 0) {
            if (L > 0 && !use_front) {  // next feasible prefix, every lane in lockstep (DfsTraverseX order)
                int cd = 0;
                uint32_t pmd = 0;
#pragma unroll
                for (int e = 0; e < E; e++) if (e == d) { cd = cn[e]; pmd = dag_pm[e]; }
                if (d == 0) cd = w0_end;
                int c = px[d] + 1;
                bool found = false;
                int64_t fst = 0, fen = 0;
                double fpp = 0.0;
                // the threshold a tuple must reach to matter (see "bounds of the pruned walk")
                double thr = -dinf();
                if (prune) {
                    if (exact_replay) { if (__shfl(hp.nheap, 0) == kTopK) thr = sheap[0].score; }
                    else { thr = nk == kTopK ? ts[kTopK - 1] : -dinf(); thr = thr0 > thr ? thr0 : thr; }
                }
                uint32_t npd = 0, pld = 0;
                long long taild = 0;
#pragma unroll
                for (int e = 0; e < E; e++) if (e == d) { npd = dag_np[e]; pld = dag_pl[e]; taild = tail[e]; }
                for (; c < cd; c++) {
                    const int64_t st = ls[d][c];
                    bool ok = true;
                    for (int p = 0; p < d; p++)
                        if (((pmd >> p) & 1) && pxe[p] > st) { ok = false; break; }
                    if (!ok) continue;
                    if (prune) {
                        double pp = d > 0 ? pxp[d - 1] : 0.0;
                        if (npd == 0) pp += troot[d][c];
                        for (int j = 0; j < (int)npd; j++) {
                            const uint32_t pj = (pld >> (4 * j)) & 15u;
                            if (pj & 8u) pp += tpair[(int)tp_off[d][j] + px[pj & 7u] * cd_full(d) + c];
                        }
                        if (pp + sbound[d + 1] + mclose + margin < thr) {   // nothing below this prefix reaches the threshold
                            if (any_order == 0) leaves += taild;            // ... every grid point below it is a tuple
                            continue;
                        }
                        fpp = pp;
                    }
                    fst = st; fen = le[d][c]; found = true; break;
                }
                wave_sync();   // every lane has read px[d] before it changes
                if (!found) { d--; continue; }
                if (t == 0) { px[d] = c; pxs[d] = fst; pxe[d] = fen; pxp[d] = fpp; }
                if (d < L - 1) {
                    d++;
                    if (t == 0) px[d] = -1;
                    wave_sync();
                    continue;
                }
                wave_sync();
            }
            bool any = false;
            unsigned long long ent_next = use_front && t < Gtot ? front[t] : 0ull;   // listed tuples: one batch ahead of their use
            for (int base = 0; base < Gtot; base += nt) {
                const int g = base + t;
                bool ok = g < Gtot;
                double score = 0.0;
                int32_t x[E];
                int64_t xs[E], xe[E];
                if (use_front) {
#pragma unroll
                    for (int e = 0; e < E; e++) x[e] = (int32_t)((ent_next >> (8 * e)) & 255ull);
                    ent_next = g + nt < Gtot ? front[g + nt] : 0ull;
                } else grid_digits(ok ? g : 0, x);
                // the lane's tuple as staged positions, 8 bits each: whoever inserts it gets it by shuffle (the listed tuples
                // live in global memory: reading them again, one lane at a time, cost a round trip per insertion)
                unsigned long long packed = 0ull;
#pragma unroll
                for (int e = 0; e < E; e++) packed |= (unsigned long long)(uint32_t)x[e] << (8 * e);
#pragma unroll
                for (int e = 0; e < E; e++) {
                    if (e < L) {
                        if (use_front) { xs[e] = ls[e][x[e]]; xe[e] = le[e][x[e]]; }
                        else { xs[e] = pxs[e]; xe[e] = pxe[e]; }
                    } else {
                        const int64_t st = ls[e][x[e]];
#pragma unroll
                        for (int p = 0; p < e; p++)
                            if (((dag_pm[e] >> p) & 1) && xe[p] > st) ok = false;
                        xs[e] = st; xe[e] = le[e][x[e]];
                    }
                }
                if (ok) {
                    // ScoreAssignmentAsPerInvocationGraph (traceweaver_v1.py:305-361) from the term tables
                    int last = 0;
                    int64_t last_end = xe[0];
#pragma unroll
                    for (int e = 1; e < E; e++) if (xe[e] > last_end) { last_end = xe[e]; last = e; }
#pragma unroll
                    for (int e = 0; e < E; e++) {
                        const int np = (int)dag_np[e];
#pragma unroll
                        for (int j = 0; j < E; j++) {
                            if (j >= np) continue;
                            const uint32_t pj = (dag_pl[e] >> (4 * j)) & 15u;
                            if (!(pj & 8u)) continue;
                            const int p = (int)(pj & 7u);
                            int64_t pend = 0;
                            int xp = 0;
#pragma unroll
                            for (int q = 0; q < E; q++) if (q == p) { pend = xe[q]; xp = x[q]; }
                            const int off = E > 1 ? (int)tp_off[e][j] : -1;
                            score += off >= 0 ? tpair[off + xp * cn[e] + x[e]] : score_term(S, slot_prim(E, p, e), pend, xs[e]);
                        }
                        if (np == 0) score += troot[e][x[e]];
                        if (e == last) score += tclose[e][x[e]];
                    }
                    if (pass == 1 && mode == 0) {
#pragma unroll
                        for (int e = 0; e < E; e++)
                            if (e >= L || use_front) {
                                const int r = lr[e][x[e]];
#pragma unroll
                                for (int w = 0; w < kBitWords; w++) if ((r >> 6) == w) mybits[e][w] |= 1ull << (r & 63);
                            }
                    }
                }
                const unsigned long long feasible = __ballot(ok);
                leaves += __popcll(feasible);
                any |= feasible != 0;
                if (exact_replay) {
                    // A tuple whose score is strictly below the score of the heap minimum is a no-op on a full heap
                    // of five: heappush sifts it to the root (slots 5 -> 2 -> 0), heappop removes it again and the
                    // displaced entries return to the slots they came from (the old root wins the comparison at
                    // slot 2 because no child is less than its parent) -- the array is bit-identical afterwards.
                    const int nh = __shfl(hp.nheap, 0);
                    const double hmin = nh == kTopK ? sheap[0].score : -dinf();
                    unsigned long long todo = __ballot(ok && !(score < hmin));
                    while (todo) {  // in enumeration order, CPython's heappush / heappop replayed by lane 0
                        const int j = __ffsll((long long)todo) - 1;
                        todo &= todo - 1;
                        const double sj = __shfl(score, j);
                        const unsigned long long pj = __shfl(packed, j);
                        if (t == 0) {
                            Cand<E> cand;
                            cand.score = sj;
#pragma unroll
                            for (int e = 0; e < E; e++) cand.idx[e] = (int32_t)((pj >> (8 * e)) & 255ull);
                            hp.push(cand);
                            if (part_log) {   // (its score was not below the root's when it came: it may enter the heap of the whole enumeration)
                                if (nlog < kPartLogCap) {
                                    unsigned long long wpos = 0ull;   // positions in the cut-off windows: the same in every part
#pragma unroll
                                    for (int e = 0; e < E; e++) wpos |= (unsigned long long)lr[e][cand.idx[e]] << (8 * e);
                                    P.part_log_sc[(int64_t)part_slot * kPartLogCap + nlog] = sj;
                                    P.part_log_ix[(int64_t)part_slot * kPartLogCap + nlog] = wpos;
                                }
                                nlog++;
                            }
                        }
                    }
                } else {
                    // beats the current fifth entry?  score ties are resolved exactly below, so let them through
                    const bool beats = ok && (nk < kTopK || score >= ts[kTopK - 1]);
                    unsigned long long todo = __ballot(beats);
                    if (__popcll(todo) > 2 * kTopK) {
                        // many candidates at once (the first batches of a span): only the five best of this batch, and
                        // whatever ties with the fifth, can be among the final five -- the others are strictly below five
                        // tuples of this very batch.  thr = fifth largest score of the batch (5 rounds of wave maximum).
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
                        const unsigned long long pj = __shfl(packed, j);
                        int32_t cj[E];
#pragma unroll
                        for (int e = 0; e < E; e++) cj[e] = (int32_t)((pj >> (8 * e)) & 255ull);
                        // exact order against every kept entry of equal score (rare): greater[k] / equivalence
                        int tie[kTopK];
#pragma unroll
                        for (int k = 0; k < kTopK; k++) {
                            tie[k] = 0;
                            if (k < nk && sj == ts[k]) {
                                int sl = 0;
#pragma unroll
                                for (int q = 0; q < kTopK; q++) if (q == k) sl = tslot[q];
                                tie[k] = tie_order(cj, sl);
                                if (tie[k] == 0) ambiguous = true;
                            }
                        }
                        if (nk == kTopK) {
                            if (!(sj > ts[kTopK - 1] || (sj == ts[kTopK - 1] && tie[kTopK - 1] > 0))) continue;  // stays out
                            // the entry that drops out must be the unique minimum of the kept five
                            if (ts[kTopK - 2] == ts[kTopK - 1]) {
                                int sa = 0, sb = 0;
#pragma unroll
                                for (int q = 0; q < kTopK; q++) { if (q == kTopK - 2) sa = tslot[q]; if (q == kTopK - 1) sb = tslot[q]; }
                                if (slot_order(sa, sb) == 0) ambiguous = true;
                            }
                        }
                        const int lastpos = nk < kTopK ? nk : kTopK - 1;
                        int slot = 0;  // free slot, or the slot of the entry that drops out
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
                        wave_sync();   // tie_order of this round has read keep_idx[slot] before it is overwritten
                        if (t == 0) {
#pragma unroll
                            for (int e = 0; e < E; e++) keep_idx[slot][e] = cj[e];
                        }
                        wave_sync();
                    }
                }
                wave_sync();
                if (nparts > 1 && ambiguous) break;   // (uniform) the span will be enumerated as a whole anyway
            }
            if (nparts > 1 && ambiguous) break;
            if (t == 0 && any && pass == 1 && mode == 0 && !use_front)
                for (int e = 0; e < L; e++) {
                    const int r = lr[e][px[e]];
                    sbits[e][r >> 6] |= 1ull << (r & 63);
                }
            if (once) break;
        }
        if (attempt == 0) {  // the kept tuples themselves must be pairwise ordered
#pragma unroll
            for (int a = 0; a < kTopK; a++)
#pragma unroll
                for (int b2 = a + 1; b2 < kTopK; b2++)
                    if (b2 < nk && ts[a] == ts[b2]) {
                        int sa = 0, sb = 0;
#pragma unroll
                        for (int q = 0; q < kTopK; q++) { if (q == a) sa = tslot[q]; if (q == b2) sb = tslot[q]; }
                        if (slot_order(sa, sb) == 0) ambiguous = true;
                    }
            if (nparts > 1) { part_ambiguous = ambiguous; break; }   // a part cannot replay CPython's heap: the whole span is redone (k_merge_parts)
            if (!ambiguous) break;
            wave_sync();
        }
        }  // attempt
        if (pass == 1 && mode == 0) {
#pragma unroll
            for (int e = 0; e < E; e++)
#pragma unroll
                for (int w = 0; w < kBitWords; w++) if (mybits[e][w] != 0ull) atomicOr(&sbits[e][w], mybits[e][w]);
            if (prune && any_order == 0) {   // (no call-order constraints: every staged candidate occurs in a tuple, visited or cut)
#pragma unroll
                for (int e = 0; e < E; e++)
                    for (int c = t; c < cn[e]; c += nt) { const int r = lr[e][c]; atomicOr(&sbits[e][r >> 6], 1ull << (r & 63)); }
            }
        }
        // the tuples of the first solve are the same in both passes: where the pruned walk of pass 2 did not count them (call-order
        // constraints: the cut prefixes' tuples are not a product), the count of pass 1 stands
        if (prune && any_order != 0) leaves = counted ? leaves_counted : (nparts > 1 ? 0 : P.leaves0[U.in_off + i]);
        wave_sync();
#ifdef TW_PROFILE
        {   // what the longest item was: flags | parts << 8 | part << 16 | staged candidates per endpoint, 6 bits each, from bit 24
            unsigned long long info = (prune ? 1ull : 0ull) | (exact_replay ? 2ull : 0ull) | (use_front ? 4ull : 0ull) | (counted ? 8ull : 0ull) | (part_log ? 16ull : 0ull) |
                                      (list_scored ? 32ull : 0ull) | (list_all ? 64ull : 0ull) | (tables_ok ? 128ull : 0ull) | ((unsigned long long)nparts << 8) | ((unsigned long long)(part_no & 255) << 16);
#pragma unroll
            for (int e = 0; e < E; e++) info |= (unsigned long long)(cn[e] > 31 ? 31 : cn[e]) << (24 + 5 * e);
            TW_ITEM_INFO(info);
        }
#endif
        TW_PHASE(3);
        if (t == 0) {
            if (exact_replay) hp.sort_desc();
            else {
#pragma unroll
                for (int k = 0; k < kTopK; k++) {
                    if (k >= nk) continue;
                    sheap[k].score = ts[k];
                    int sl = 0;
#pragma unroll
                    for (int q = 0; q < kTopK; q++) if (q == k) sl = tslot[q];
                    for (int e = 0; e < E; e++) sheap[k].idx[e] = keep_idx[sl][e];
                }
            }
        }
        wave_sync();
        nout = exact_replay ? __shfl(hp.nheap, 0) : nk;
        }  // !none
        if (nparts > 1) {   // a part: its top-5 (span indices), tuple count and candidate bitmap go to the scratch slot
            if (t == 0) {
                P.part_n[part_slot] = nout | ((part_ambiguous || part_failed) ? 256 : 0) | (nlog > kPartLogCap ? 512 : 0);   // (flags: the span is enumerated again as a whole)
                P.part_leaves[part_slot] = leaves;
                if (part_log) P.part_logn[part_slot] = nlog;
            }
            for (int q = t; q < kTopK * (E + 1); q += nt) {
                const int k = q / (E + 1), f = q % (E + 1);
                if (k >= nout) continue;
                if (f == E) P.part_score[(int64_t)part_slot * kTopK + k] = sheap[k].score;
                else {
                    int loe = 0;
#pragma unroll
                    for (int x = 0; x < E; x++) if (x == f) loe = lo[x];
                    P.part_idx[((int64_t)part_slot * kTopK + k) * kMaxEp + f] = loe + (int)lr[f][sheap[k].idx[f]];
                }
            }
            for (int q = t; q < E * kCandWords; q += nt) P.part_bits[((int64_t)part_slot * kMaxEp + q / kCandWords) * kCandWords + q % kCandWords] = sbits[q / kCandWords][q % kCandWords];
        } else if (!deferred)
        {   // results leave through all lanes: one (entry, field) per lane; staged positions back to span indices
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
                if (mode == 0 && k >= nout) continue;  // unused entries keep the -1 / NaN pattern they were given at load time
                if (f == E) out_score[tks_index(U, k, i)] = k < nout ? sheap[k].score : dnan();
                else {
                    int loe = 0;
#pragma unroll
                    for (int x = 0; x < E; x++) if (x == f) loe = lo[x];
                    out_idx[tk_index(U, k, f, i)] = k < nout ? loe + (int)lr[f][sheap[k].idx[f]] : -1;
                }
            }
            if (pass == 1 && mode == 0) {
                for (int q = t; q < E * kCandWords; q += nt) {
                    const int e = q / kCandWords, f = q % kCandWords;
                    P.c_bits[ie_index(U, e, i) * kCandWords + f] = sbits[e][f];
                }
            }
        }
        wave_sync();
        if (big_slot >= 0 && t == 0) pool_release(P.frontier_big_busy, big_slot);
        TW_PHASE(4);
        TW_ITEM_END(leaves);
    }
    if (front_slot >= 0 && t == 0) pool_release(P.frontier_busy, front_slot);
}

// Combines the parts of the split spans of the endpoint-count class E (see kSplitMin): the five largest tuples of the union
// under Python's order of (score, [spans]) -- score, then start_mus of the first differing span -- the sum of the tuple
// counts and the union of the candidate bitmaps.  The parts partition the tuples by their first span, in enumeration order,
// and each kept its own five largest, so the five largest of the union are among the kept ones.  Where that order does not
// decide (two kept tuples of equal score whose first differing spans start together, inside a part or across parts), the
// reference's result depends on the order of its heap operations: such a span is put back on the class' list (the host has
// emptied it) and enumerated once more by one wavefront, which replays CPython's heap.
// One wavefront per split span: the kept tuples (<= kMaxParts * 5) sit in LDS, every lane ranks one or two of them against
// all others, and the tuples ranked 0..4 are written by the lanes that hold them.
// A split span in log mode (twin candidates, see heavy_append): the concatenation of the parts' logs, part after part, is replayed
// through CPython's heapq (push; beyond five entries pop the smallest) and sorted like the reference's final list.sort -- LdsHeap's
// operations on (score, window positions) entries with the endpoint count at run time.  Lane 0 works; the logs and the start
// times of the spans' cut-off windows sit in LDS.
// CPython's heap of the replay (traceweaver_v3.py:304-307: heappush, beyond five entries heappop) in registers, the same values in
// every lane, every lane computing alike: the six slots are named, the sift paths of a heap this small are written out (push at slot
// n: parents (n - 1) >> 1; pop of a six-entry heap: the last entry sifts up from the root through the smaller children, then down
// again -- heapq._siftup / _siftdown).  Entries are (score, tuple as 8-bit positions per endpoint); Python's order on equal scores is
// start_mus of the first differing span: start_of(endpoint, position).  (A heap in LDS worked by lane 0 cost ~2.5 us a push, some
// fifty dependent LDS operations; on millisecond-granular traces -- twin candidates everywhere -- that was half of an item.)
struct RegHeap {
    double s[kTopK + 1];
    unsigned long long x[kTopK + 1];
    int n, E;
    __device__ __forceinline__ void clear(int endpoints) {
        E = endpoints; n = 0;
#pragma unroll
        for (int k = 0; k <= kTopK; k++) { s[k] = 0.0; x[k] = 0ull; }
    }
    template <class F>
    __device__ __forceinline__ bool lt(double sa, unsigned long long xa, double sb, unsigned long long xb, const F& start_of) const {   // Python's (score, [spans]) <
        if (sa != sb) return sa < sb;
        if (xa == xb) return false;
        for (int e = 0; e < E; e++) {
            const int a = (int)((xa >> (8 * e)) & 255ull), b = (int)((xb >> (8 * e)) & 255ull);
            if (a != b) return start_of(e, a) < start_of(e, b);
        }
        return false;
    }
    __device__ __forceinline__ bool below_root(double sc) const { return n == kTopK && sc < s[0]; }   // a push that leaves a full heap as it is (k_enumerate_heavy)
    template <class F>
    __device__ __forceinline__ void push(double sc, unsigned long long xc, const F& start_of) {
        static_assert(kTopK == 5, "the sift paths are written for a heap of five (+ 1)");
        if (n < kTopK) {   // heappush only: the new entry at slot n sifts towards the root
            if (n == 0) { s[0] = sc; x[0] = xc; }
            else if (n <= 2) {
                const bool up = lt(sc, xc, s[0], x[0], start_of);
                const double ds = up ? s[0] : sc; const unsigned long long dx = up ? x[0] : xc;
                if (n == 1) { s[1] = ds; x[1] = dx; } else { s[2] = ds; x[2] = dx; }
                if (up) { s[0] = sc; x[0] = xc; }
            } else {   // slot 3 or 4: parent 1, then 0
                const bool up1 = lt(sc, xc, s[1], x[1], start_of);
                const double ds = up1 ? s[1] : sc; const unsigned long long dx = up1 ? x[1] : xc;
                if (n == 3) { s[3] = ds; x[3] = dx; } else { s[4] = ds; x[4] = dx; }
                if (up1) {
                    if (lt(sc, xc, s[0], x[0], start_of)) { s[1] = s[0]; x[1] = x[0]; s[0] = sc; x[0] = xc; } else { s[1] = sc; x[1] = xc; }
                }
            }
            n++;
            return;
        }
        // heappush at slot 5 (parents 2, 0) ...
        double ls5 = sc; unsigned long long lx5 = xc;   // slot 5 after the push = what heappop takes off the end
        if (lt(sc, xc, s[2], x[2], start_of)) {
            ls5 = s[2]; lx5 = x[2];
            if (lt(sc, xc, s[0], x[0], start_of)) { s[2] = s[0]; x[2] = x[0]; s[0] = sc; x[0] = xc; } else { s[2] = sc; x[2] = xc; }
        }
        // ... heappop: the root leaves, the last entry sifts up from the root (five entries: slots 0 .. 4) through the smaller children
        int pos;
        if (!lt(s[1], x[1], s[2], x[2], start_of)) { s[0] = s[2]; x[0] = x[2]; pos = 2; }
        else {
            s[0] = s[1]; x[0] = x[1];
            if (!lt(s[3], x[3], s[4], x[4], start_of)) { s[1] = s[4]; x[1] = x[4]; pos = 4; } else { s[1] = s[3]; x[1] = x[3]; pos = 3; }
        }
        // ... and down again from the leaf it reached (_siftdown(heap, 0, pos))
        if (pos == 2) {
            if (lt(ls5, lx5, s[0], x[0], start_of)) { s[2] = s[0]; x[2] = x[0]; s[0] = ls5; x[0] = lx5; } else { s[2] = ls5; x[2] = lx5; }
        } else {
            const bool up1 = lt(ls5, lx5, s[1], x[1], start_of);
            const double ds = up1 ? s[1] : ls5; const unsigned long long dx = up1 ? x[1] : lx5;
            if (pos == 3) { s[3] = ds; x[3] = dx; } else { s[4] = ds; x[4] = dx; }
            if (up1) {
                if (lt(ls5, lx5, s[0], x[0], start_of)) { s[1] = s[0]; x[1] = x[0]; s[0] = ls5; x[0] = lx5; } else { s[1] = ls5; x[1] = lx5; }
            }
        }
    }
};
// the value lane j holds (j wave-uniform), in scalar registers: v_readlane instead of a shuffle through the LDS crossbar
#ifdef TW_HOST_EMULATION
__device__ __forceinline__ unsigned long long lane_value(unsigned long long v, int j) { return __shfl(v, j); }
#else
__device__ __forceinline__ unsigned long long lane_value(unsigned long long v, int j) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, j), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), j);
    return ((unsigned long long)hi << 32) | lo;
}
#endif
__device__ __forceinline__ double lane_value(double v, int j) { return __longlong_as_double((long long)lane_value((unsigned long long)__double_as_longlong(v), j)); }

// list.sort(reverse=True) of a few (score, tuple) entries under Python's order `lt` -- a strict partial order (tuples whose first
// differing spans start together are incomparable), so the steps of CPython's sort are followed one by one: reverse, count_run +
// binary insertion, reverse.
template <class LT>
__device__ __forceinline__ void py_sort_desc(double* hs, unsigned long long* hx, int n, const LT& lt) {
    auto reverse = [&](int m) {
        for (int i = 0, j = m - 1; i < j; i++, j--) {
            const double a = hs[i]; hs[i] = hs[j]; hs[j] = a;
            const unsigned long long b = hx[i]; hx[i] = hx[j]; hx[j] = b;
        }
    };
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

struct MergeHeap {
    double* hs;                      // [kTopK + 1] scores
    unsigned long long* hx;          // [kTopK + 1] positions in the cut-off windows, 8 bits per endpoint
    int n, E;
    const int64_t (*st)[64 * kCandWords];   // start times of the windows' spans [endpoint][position]
    __device__ bool lt(double sa, unsigned long long xa, double sb, unsigned long long xb) const {
        if (sa != sb) return sa < sb;
        for (int e = 0; e < E; e++) {
            const int a = (int)((xa >> (8 * e)) & 255ull), b = (int)((xb >> (8 * e)) & 255ull);
            if (a != b) return st[e][a] < st[e][b];
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
            const double ls = hs[n]; const unsigned long long lx = hx[n];
            if (n > 0) { hs[0] = ls; hx[0] = lx; siftup(0); }
        }
    }
    __device__ void sort_desc() {   // list.sort(reverse=True) of <= 5 entries
        py_sort_desc(hs, hx, n, [&](double sa, unsigned long long xa, double sb, unsigned long long xb) { return lt(sa, xa, sb, xb); });
    }
};

__global__ void __launch_bounds__(64) k_merge_parts(Dev P, int pass, int E_lo, int E_hi) {   // the split spans of the classes E_lo .. E_hi in one launch
    if (*P.err != 0) return;
    constexpr int kCandMax = kMaxParts * kTopK;
    __shared__ double sc[kCandMax];
    __shared__ int32_t sp[kCandMax][kMaxEp];
    __shared__ int64_t ss[kCandMax][kMaxEp];    // start of the tuple's span at every endpoint
    __shared__ uint8_t ok[kCandMax];
    __shared__ uint16_t rk[kCandMax];
    __shared__ int redo_flag;
    __shared__ int64_t w_st[kMaxEp][64 * kCandWords];           // log mode: start times of the cut-off windows' spans
    __shared__ double h_sc[kTopK + 1];
    __shared__ unsigned long long h_ix[kTopK + 1];
    __shared__ int h_n;
    const int t = threadIdx.x, nt = blockDim.x;
    int n_all = 0;
    for (int Ec = E_lo; Ec <= E_hi; Ec++) n_all += P.split_count[Ec];
    for (int s_all = (int)blockIdx.x; s_all < n_all; s_all += (int)gridDim.x) {
        int E = E_lo, s = s_all;   // the class of the record and its number within the class
        while (s >= P.split_count[E]) { s -= P.split_count[E]; E++; }
        const int rec = P.part_off[E] + s;
        const int unit = P.split_unit[rec], i = P.split_idx[rec], slot0 = P.split_slot[rec];
        const int nparts = P.split_parts[rec] & 255, wide_bit = P.split_parts[rec] & (1 << 24);
        const UnitDev& U = P.units[unit];
        const int C = nparts * kTopK;
        if (P.split_parts[rec] & kPartLogFlag) {
            // ---- log mode: replay the parts' logs
            bool redo = false;
            long long leaves = 0;
            for (int p = 0; p < nparts; p++) {   // (<= kMaxParts: every lane the same few loads)
                if (P.part_n[slot0 + p] >> 8) redo = true;   // a log that is not complete
                leaves += P.part_leaves[slot0 + p];
            }
            if (redo) {
                if (t == 0) {
                    const int q = P.heavy_big_off[E] + atomicAdd(&P.redo_count[E], 1);
                    P.heavy_big_unit[q] = unit; P.heavy_big_idx[q] = i; P.heavy_big_part[q] = 1 | wide_bit | kReplayFlag; P.heavy_big_slot[q] = 0;
                    atomicAdd(&P.split_count[0], 1);
                }
                wave_sync();
                continue;
            }
            for (int e = 0; e < E; e++) {
                const int lo = P.c_lo[ie_index(U, e, i)], w = P.c_hi[ie_index(U, e, i)] - lo + 1;
                for (int r = t; r < w && r < 64 * kCandWords; r += nt) w_st[e][r] = P.out_start[U.ep_off[e] + lo + r];
            }
            // the logs in part order = enumeration order, 64 entries at a time: an entry per lane, those that can change the heap pushed
            // one after the other (every lane keeps the heap, RegHeap); the final list.sort by lane 0 on the LDS copy
            RegHeap RH;
            RH.clear(E);
            auto start_of = [&](int e, int pos2) -> int64_t { return w_st[e][pos2]; };
            wave_sync();
            for (int p = 0; p < nparts; p++) {
                const int m = P.part_logn[slot0 + p];
                for (int base = 0; base < m; base += nt) {
                    const int k = base + t;
                    const bool valid = k < m;
                    const double sk = valid ? P.part_log_sc[(int64_t)(slot0 + p) * kPartLogCap + k] : 0.0;
                    const unsigned long long xk = valid ? P.part_log_ix[(int64_t)(slot0 + p) * kPartLogCap + k] : 0ull;
                    unsigned long long todo = __ballot(valid && !RH.below_root(sk));   // strictly below the root of a full heap: the push leaves the array as it is
                    while (todo) {
                        const int j = __ffsll((long long)todo) - 1;
                        todo &= todo - 1;
                        const double sj = lane_value(sk, j);
                        if (RH.below_root(sj)) continue;   // (the root has risen since the ballot)
                        RH.push(sj, lane_value(xk, j), start_of);
                    }
                }
            }
            MergeHeap H;
            H.hs = h_sc; H.hx = h_ix; H.n = RH.n; H.E = E; H.st = w_st;
            if (t == 0) {
#pragma unroll
                for (int k = 0; k < kTopK; k++) { h_sc[k] = RH.s[k]; h_ix[k] = RH.x[k]; }
                H.sort_desc(); h_n = H.n;
            }
            wave_sync();
            const int nout = h_n;
            const int64_t g = U.in_off + i;
            if (pass == 1) { if (t == 0) P.leaves0[g] = leaves; }
            else leaves = P.leaves0[g];
            if (t == 0) { P.tk_n[g] = nout; P.leaves[g] = leaves; P.rep[g] = 0; }
            for (int q = t; q < nout * (E + 1); q += nt) {   // entries a span does not have keep the -1 / NaN pattern of tw_load_batch
                const int k = q / (E + 1), f = q % (E + 1);
                if (f == E) P.tk_score[tks_index(U, k, i)] = h_sc[k];
                else P.tk_idx[tk_index(U, k, f, i)] = P.c_lo[ie_index(U, f, i)] + (int)((h_ix[k] >> (8 * f)) & 255ull);
            }
            if (pass == 1)
                for (int q = t; q < E * kCandWords; q += nt) {
                    const int e = q / kCandWords, w = q % kCandWords;
                    unsigned long long bits = 0ull;
                    for (int p = 0; p < nparts; p++) bits |= P.part_bits[((int64_t)(slot0 + p) * kMaxEp + e) * kCandWords + w];
                    P.c_bits[ie_index(U, e, i) * kCandWords + w] = bits;
                }
            wave_sync();
            continue;
        }
        if (t == 0) redo_flag = 0;
        for (int c = t; c < C; c += nt) {
            const int slot = slot0 + c / kTopK, k = c % kTopK;
            const int np = P.part_n[slot];
            const bool have = k < (np & 255);
            ok[c] = have ? 1 : 0;
            sc[c] = have ? P.part_score[(int64_t)slot * kTopK + k] : 0.0;
            for (int e = 0; e < E; e++) {
                const int x = have ? P.part_idx[((int64_t)slot * kTopK + k) * kMaxEp + e] : 0;
                sp[c][e] = x;
                ss[c][e] = have ? P.out_start[U.ep_off[e] + x] : 0;
            }
        }
        wave_sync();
        // rank of every kept tuple = number of kept tuples that are greater; an undecided pair matters when it reaches the
        // fifth place (both ranks are then off by at most the number of such pairs: any one of them within the first five
        // or straddling it is reported)
        bool undecided = false;
        for (int a = t; a < C; a += nt) {
            if (!ok[a]) { rk[a] = (uint16_t)kCandMax; continue; }
            int r = 0;
            bool tie_here = false;
            for (int b = 0; b < C; b++) {
                if (b == a || !ok[b]) continue;
                if (sc[b] > sc[a]) { r++; continue; }
                if (sc[b] < sc[a]) continue;
                int o = 0;   // +1: b greater
                for (int e = 0; e < E; e++)
                    if (sp[a][e] != sp[b][e]) { o = ss[b][e] > ss[a][e] ? 1 : (ss[b][e] < ss[a][e] ? -1 : 0); break; }
                if (o > 0) r++;
                if (o == 0) tie_here = true;
            }
            rk[a] = (uint16_t)r;
            if (tie_here && r <= kTopK) undecided = true;
        }
        int total = 0;
        long long leaves = 0;
        for (int p = 0; p < nparts; p++) {   // (<= kMaxParts: every lane the same few loads)
            const int np = P.part_n[slot0 + p];
            if (np >> 8) undecided = true;
            total += np & 255;
            leaves += P.part_leaves[slot0 + p];
        }
        if (undecided) redo_flag = 1;
        wave_sync();
        const bool redo = redo_flag != 0;
        if (redo) {
            if (t == 0) {
                const int q = P.heavy_big_off[E] + atomicAdd(&P.redo_count[E], 1);
                P.heavy_big_unit[q] = unit; P.heavy_big_idx[q] = i; P.heavy_big_part[q] = 1 | wide_bit | kReplayFlag; P.heavy_big_slot[q] = 0;
                atomicAdd(&P.split_count[0], 1);   // (classes start at E = 2: entry 0 counts the spans listed again, for tw_debug_worklists)
            }
        } else {
            const int64_t g = U.in_off + i;
            if (pass == 1) { if (t == 0) P.leaves0[g] = leaves; }
            else leaves = P.leaves0[g];   // (a pruned part of pass 2 does not count what it cuts: same tuples as in pass 1)
            if (t == 0) { P.tk_n[g] = total < kTopK ? total : kTopK; P.leaves[g] = leaves; P.rep[g] = 0; }
            for (int a = t; a < C; a += nt) {   // entries a span does not have keep the -1 / NaN pattern of tw_load_batch
                const int k = rk[a];
                if (k >= kTopK) continue;
                P.tk_score[tks_index(U, k, i)] = sc[a];
                for (int e = 0; e < E; e++) P.tk_idx[tk_index(U, k, e, i)] = sp[a][e];
            }
            if (pass == 1)
                for (int q = t; q < E * kCandWords; q += nt) {
                    const int e = q / kCandWords, w = q % kCandWords;
                    unsigned long long bits = 0ull;
                    for (int p = 0; p < nparts; p++) bits |= P.part_bits[((int64_t)(slot0 + p) * kMaxEp + e) * kCandWords + w];
                    P.c_bits[ie_index(U, e, i) * kCandWords + w] = bits;
                }
        }
        wave_sync();
    }
}

// ---------------------------------------------------------------------------------------------
// Segmented (per unit) scans over the incoming spans, three phases: tile-local inclusive scan,
// per-unit scan of the tile aggregates, fix-up.
template <class T, class C>
__device__ T block_scan_incl(T v, T* sh, C comb) {
    const int t = threadIdx.x, n = blockDim.x;
    sh[t] = v;
    __syncthreads();
    for (int off = 1; off < n; off <<= 1) {
        T a = sh[t];
        if (t >= off) a = comb(sh[t - off], a);
        __syncthreads();
        sh[t] = a;
        __syncthreads();
    }
    return sh[t];
}

struct ScanMaxEnd {  // running arg max of in_end, latest index on ties (PerfectCut's prev_index)
    typedef PairVI T;
    __device__ static T identity() { return PairVI{INT64_MIN, -1}; }
    __device__ static T comb(T a, T b) { return b.v >= a.v ? b : a; }
    __device__ static T load(const Dev& P, const UnitDev& U, int i) { return PairVI{P.in_end[U.in_off + i], i}; }
    __device__ static void store(const Dev& P, const UnitDev& U, int i, T v) { P.pm_val[U.in_off + i] = v.v; P.pm_idx[U.in_off + i] = v.i; }
    __device__ static T reload(const Dev& P, const UnitDev& U, int i) { return PairVI{P.pm_val[U.in_off + i], P.pm_idx[U.in_off + i]}; }
};
struct ScanSegStart {  // latest PerfectCut position <= i
    typedef int32_t T;
    __device__ static T identity() { return 0; }
    __device__ static T comb(T a, T b) { return a > b ? a : b; }
    __device__ static T load(const Dev& P, const UnitDev& U, int i) { return P.pc[U.in_off + i] ? i : 0; }
    __device__ static void store(const Dev& P, const UnitDev& U, int i, T v) { P.seg[U.in_off + i] = v; }
    __device__ static T reload(const Dev& P, const UnitDev& U, int i) { return P.seg[U.in_off + i]; }
};
struct ScanWinId {  // inclusive count of window ends; the window id is count - win_end[i]
    typedef int32_t T;
    __device__ static T identity() { return 0; }
    __device__ static T comb(T a, T b) { return a + b; }
    __device__ static T load(const Dev& P, const UnitDev& U, int i) { return P.win_end[U.in_off + i]; }
    __device__ static void store(const Dev& P, const UnitDev& U, int i, T v) { P.wid[U.in_off + i] = v; }
    __device__ static T reload(const Dev& P, const UnitDev& U, int i) { return P.wid[U.in_off + i]; }
};

template <class Tr>
__global__ void k_scan_local(Dev P, TileSet S, typename Tr::T* agg) {
    if (*P.err != 0) return;  // an earlier kernel of this pass reported an error: its outputs are not usable
    __shared__ typename Tr::T sh[kTile];
    const int tile = set_tile(S);
    const TileDev Tl = P.tiles[tile];
    const UnitDev& U = P.units[Tl.unit];
    const int i = Tl.first + threadIdx.x;
    typename Tr::T v = i < U.n_in ? Tr::load(P, U, i) : Tr::identity();
    v = block_scan_incl(v, sh, Tr::comb);
    if (i < U.n_in) Tr::store(P, U, i, v);
    if (threadIdx.x == blockDim.x - 1) agg[tile] = v;
}
template <class Tr>
__global__ void k_scan_spine(Dev P, TileSet S, typename Tr::T* agg) {  // one workgroup per unit: exclusive scan of its tiles
    if (*P.err != 0) return;  // an earlier kernel of this pass reported an error: its outputs are not usable
    __shared__ typename Tr::T sh[kCoop];
    __shared__ typename Tr::T carry_sh;
    const UnitDev& U = P.units[blockIdx.x];
    if (S.slot != 0 && U.E != S.slot) return;   // (a class' launch: the units of the other classes are not its business)
    const int t = threadIdx.x, n = blockDim.x;
    if (t == 0) carry_sh = Tr::identity();
    __syncthreads();
    for (int base = 0; base < U.ntile; base += n) {
        const int k = base + t;
        typename Tr::T v = k < U.ntile ? agg[U.tile_off + k] : Tr::identity();
        const typename Tr::T inc = block_scan_incl(v, sh, Tr::comb);
        const typename Tr::T carry = carry_sh;
        const typename Tr::T excl = t == 0 ? carry : Tr::comb(carry, sh[t - 1]);
        __syncthreads();
        if (k < U.ntile) agg[U.tile_off + k] = excl;
        if (t == n - 1) carry_sh = Tr::comb(carry, inc);
        __syncthreads();
    }
}
template <class Tr>
__global__ void k_scan_fix(Dev P, TileSet S, const typename Tr::T* agg) {
    if (*P.err != 0) return;  // an earlier kernel of this pass reported an error: its outputs are not usable
    const int tile = set_tile(S);
    const TileDev Tl = P.tiles[tile];
    const UnitDev& U = P.units[Tl.unit];
    const int i = Tl.first + threadIdx.x;
    if (i >= U.n_in || Tl.first == 0) return;
    Tr::store(P, U, i, Tr::comb(agg[tile], Tr::reload(P, U, i)));
}

// PerfectCut(i) (traceweaver_v3.py:1024-1039): candidates of the latest-ending earlier span and of
// span i are disjoint, and that earlier span ends no later than span i.
__device__ __forceinline__ uint8_t perfect_cut(const Dev& P, const UnitDev& U, int i) {
    const int64_t g = U.in_off + i;
    uint8_t cut = 0;
    if (i >= 1 && (i <= U.n_in - 2 || (U.part & TW_PART_BEFORE_CUT))) {   // (the service's last request is never tested: traceweaver_v3.py:1059)
        const int prev = P.pm_idx[g - 1];
        if (P.pm_val[g - 1] <= P.in_end[g]) {
            bool disjoint = true;
            for (int e = 0; e < U.E && disjoint; e++) {
                const int64_t a = ie_index(U, e, prev), b = ie_index(U, e, i);
                const int sh = P.c_lo[b] - P.c_lo[a];  // spans are sorted by start: lo is monotone, sh >= 0
                if (sh >= 64 * kCandWords) continue;
                for (int w = 0; w < kCandWords && disjoint; w++) {
                    // bits of `prev` that fall on word w of span i
                    const int bitpos = sh + 64 * w, ws = bitpos >> 6, bs = bitpos & 63;
                    uint64_t pa = 0;
                    if (ws < kCandWords) pa = P.c_bits[a * kCandWords + ws] >> bs;
                    if (bs != 0 && ws + 1 < kCandWords) pa |= P.c_bits[a * kCandWords + ws + 1] << (64 - bs);
                    if (pa & P.c_bits[b * kCandWords + w]) disjoint = false;
                }
            }
            cut = disjoint ? 1 : 0;
        }
    }
    return cut;
}
__global__ void k_perfect_cut(Dev P, TileSet S) {
    if (*P.err != 0) return;  // an earlier kernel of this pass reported an error: its outputs are not usable
    const TileDev Tl = P.tiles[set_tile(S)];
    const UnitDev& U = P.units[Tl.unit];
    const int i = Tl.first + threadIdx.x;
    if (i >= U.n_in) return;
    P.pc[U.in_off + i] = perfect_cut(P, U, i);
}

// Window ends (traceweaver_v3.py:1056-1076): the last span, the span before every PerfectCut, and a
// size cut every batch_size_mis spans counted from the segment start.
__device__ __forceinline__ uint8_t window_flag(const Dev& P, const UnitDev& U, int i, int s) {   // s = latest PerfectCut position <= i
    const int64_t g = U.in_off + i;
    const int n = U.n_in, B = P.batch_mis;
    bool end = (i == n - 1);
    if (!end && P.pc[g + 1]) end = true;  // pc is 0 outside [1, n-2] (n-1 for a part that a cut follows)
    if (!end && i >= 1) {
        const int d = i - s - ((s == 0 && !(U.part & TW_PART_AFTER_CUT)) ? B - 1 : B);   // (the service's first request counts twice: current_count starts at 1)
        if (d >= 0 && d % B == 0 && !P.pc[g]) end = true;
    }
    return end ? 1 : 0;
}
__device__ __forceinline__ void window_index(const Dev& P, const UnitDev& U, int unit, int i, int count) {   // count = window ends up to and including i
    const int64_t g = U.in_off + i;
    const int w = count - P.win_end[g];
    P.wid[g] = w;
    if (P.win_end[g]) P.w_last[U.in_off + w] = i;
    if (i == U.n_in - 1) P.unit_nwin[unit] = w + 1;
}
__global__ void k_window_flags(Dev P, TileSet S) {
    if (*P.err != 0) return;  // an earlier kernel of this pass reported an error: its outputs are not usable
    const TileDev Tl = P.tiles[set_tile(S)];
    const UnitDev& U = P.units[Tl.unit];
    const int i = Tl.first + threadIdx.x;
    if (i >= U.n_in) return;
    P.win_end[U.in_off + i] = window_flag(P, U, i, P.seg[U.in_off + i]);
}
__global__ void k_window_index(Dev P, TileSet S) {  // after the ScanWinId scan: wid currently holds the inclusive count
    if (*P.err != 0) return;  // an earlier kernel of this pass reported an error: its outputs are not usable
    const TileDev Tl = P.tiles[set_tile(S)];
    const UnitDev& U = P.units[Tl.unit];
    const int i = Tl.first + threadIdx.x;
    if (i >= U.n_in) return;
    window_index(P, U, Tl.unit, i, P.wid[U.in_off + i]);
}

// The same chain in five launches instead of nine (the window stage of a class sits between its enumeration and its selection: every
// launch is a dependent step of the pass): PerfectCut + the tile-local scan of the cut positions; the scan's fix-up + the window
// flags + the tile-local count of the flags; the count's fix-up + the window index.  (k_scan_spine between them as before.)
__global__ void __launch_bounds__(kTile) k_cut_scan(Dev P, TileSet S, int32_t* agg) {
    if (*P.err != 0) return;
    __shared__ int32_t sh[kTile];
    const int tile = set_tile(S);
    const TileDev Tl = P.tiles[tile];
    const UnitDev& U = P.units[Tl.unit];
    const int i = Tl.first + threadIdx.x;
    const bool live = i < U.n_in;
    const uint8_t cut = live ? perfect_cut(P, U, i) : 0;
    if (live) P.pc[U.in_off + i] = cut;
    int32_t v = (live && cut) ? i : ScanSegStart::identity();
    v = block_scan_incl(v, sh, ScanSegStart::comb);
    if (live) P.seg[U.in_off + i] = v;
    if (threadIdx.x == blockDim.x - 1) agg[tile] = v;
}
__global__ void __launch_bounds__(kTile) k_flags_scan(Dev P, TileSet S, const int32_t* agg_seg, int32_t* agg_win) {
    if (*P.err != 0) return;
    __shared__ int32_t sh[kTile];
    const int tile = set_tile(S);
    const TileDev Tl = P.tiles[tile];
    const UnitDev& U = P.units[Tl.unit];
    const int i = Tl.first + threadIdx.x;
    const bool live = i < U.n_in;
    uint8_t end = 0;
    if (live) {
        int32_t s = P.seg[U.in_off + i];
        if (Tl.first != 0) s = ScanSegStart::comb(agg_seg[tile], s);   // (the fix-up of k_scan_fix; P.seg itself is not read again)
        end = window_flag(P, U, i, s);
        P.win_end[U.in_off + i] = end;
    }
    int32_t v = block_scan_incl((int32_t)end, sh, ScanWinId::comb);
    if (live) P.wid[U.in_off + i] = v;
    if (threadIdx.x == blockDim.x - 1) agg_win[tile] = v;
}
__global__ void __launch_bounds__(kTile) k_index_fix(Dev P, TileSet S, const int32_t* agg_win) {
    if (*P.err != 0) return;
    const int tile = set_tile(S);
    const TileDev Tl = P.tiles[tile];
    const UnitDev& U = P.units[Tl.unit];
    const int i = Tl.first + threadIdx.x;
    if (i >= U.n_in) return;
    int32_t c = P.wid[U.in_off + i];
    if (Tl.first != 0) c = ScanWinId::comb(agg_win[tile], c);
    window_index(P, U, Tl.unit, i, c);
}

// ---------------------------------------------------------------------------------------------
// Exact maximum-weight independent set of one window's conflict graph.
struct CandView {  // the current candidate list of span i (recomputed list if rep[i])
    const int32_t* idx; const double* score; int n;
};
__device__ __forceinline__ int cand_n(const Dev& P, const UnitDev& U, int i) {
    const int64_t g = U.in_off + i;
    return P.rep[g] ? P.tkr_n[g] : P.tk_n[g];
}
__device__ __forceinline__ int32_t cand_idx(const Dev& P, const UnitDev& U, int i, int k, int e) {
    const int64_t g = U.in_off + i;
    return (P.rep[g] ? P.tkr_idx : P.tk_idx)[tk_index(U, k, e, i)];
}
__device__ __forceinline__ double cand_score(const Dev& P, const UnitDev& U, int i, int k) {
    const int64_t g = U.in_off + i;
    return (P.rep[g] ? P.tkr_score : P.tk_score)[tks_index(U, k, i)];
}
__device__ inline bool cands_share(const Dev& P, const UnitDev& U, int i, int a, int j, int b) {
    for (int e = 0; e < U.E; e++)
        if (cand_idx(P, U, i, a, e) == cand_idx(P, U, j, b, e)) return true;
    return false;
}

// Exact selection, canonical procedure (shared with the oracle so that exact ties resolve identically):
//   node weight 10000 + score (traceweaver_v3.py:1260) as an exact integer, rint((10000 + score) * 2^32): the
//   reference hands binary64 weights to a MILP solver with ~1e-7 tolerances, and binary64 sums depend on the order
//   of the additions in their last bits -- "the selection of maximum weight" would then depend on how a search
//   accumulates and bounds them; with integers (resolution 2.3e-10, equal scores stay equal) sums, bounds and
//   comparisons are exact.  Nodes with weight <= 0 are never selected; the window is split into connected
//   components of the span conflict relation; the answer for a component is the FIRST selection of maximum
//   weight in depth-first order: spans in index order, candidates in list order then "none", only strict
//   improvements replace the incumbent.
//   That selection does not depend on how the tree is searched or bounded, so the oracle (one depth-first
//   search with an additional matching relaxation) and this engine (many sub-trees at once) agree whenever
//   both searches complete.  Upper bound of a suffix of the component: the remaining spans cut into groups of
//   <= 3 consecutive spans, every group solved exactly on its own, cheapest cutting (the sum of the best
//   weights is the all-singletons cutting; pairs and triples see spans that compete for the same outgoing
//   spans).
//
// Mapping: k_select_fast settles the windows whose best candidates do not clash (one lane per span) and
// lists the others; k_select_heavy solves a listed window per wavefront, candidate data in LDS: components
// of <= kBruteMax spans by complete enumeration spread over the lanes, larger ones by select_search: a
// depth-first search in that canonical order by one lane (stack in LDS), cut where weight + bound <= incumbent.
// Bounds: the grouped bound of the suffix, and a transposition table -- once the sub-tree below a node has been
// searched, (incumbent then) - (weight above the node) bounds what any other way of reaching the same (depth,
// blocked candidates of the remaining spans) can still gain; assignments of the spans above that merely permute
// who took which outgoing span all meet in that one entry.  A component whose search visits more than kNodeBudget
// nodes keeps the best selection found; the window is counted in unit_stats[4] (not proven optimal).
typedef long long sel_w;
// (a weight <= 0 is never selected, so everything from there down is 0: the conversion of a double beyond the int64 range is
// undefined -- a mixture component collapsed onto one sample value scores a candidate a few hundred microseconds off at
// -10^11 -- and on gfx950 comes out as a small *positive* number)
__device__ __forceinline__ sel_w sel_weight(double score) {
    const double w = (10000.0 + score) * 4294967296.0;
    return w > 0.5 ? (sel_w)rint(w) : 0;
}
constexpr int kNodeBudget = 1 << 24; // search nodes per component
constexpr int kBlkWords = (kMaxWin * kTopK + 63) / 64;  // words of a mask over all candidates of a component
constexpr int kBigWindow = 8;        // windows of at least this many spans are served first by k_select_heavy
constexpr int kHugeWindow = 16;      // ... and from this many on by the instantiation with the full LDS layout (three-word masks)
constexpr int kBruteMax = 4;         // components of up to this many spans are solved by complete enumeration (see select_brute)
#ifndef TW_MEMO_SLOTS
#define TW_MEMO_SLOTS 256
#endif
// select_search consults the matching relaxation from this many nodes of a component on: early for one endpoint, where the
// bound is the exact optimum of the sub-problem (the search stops backtracking: measured on the nodejs shape, 164 774 nodes ->
// a few hundred), late otherwise (one consultation costs as much as ~50 plain nodes and a second endpoint's constraint is
// what the relaxation drops: 2 506 -> 2 356 nodes on the longest two-endpoint window) -- there it is what ends tie-saturated
// searches that would otherwise run into the node budget ...
#ifndef TW_MATCH_NODES_1
#define TW_MATCH_NODES_1 2048
#endif
#ifndef TW_MATCH_NODES
#define TW_MATCH_NODES 32768
#endif
constexpr int kMatchNodes1 = TW_MATCH_NODES_1, kMatchNodes = TW_MATCH_NODES;
#ifndef TW_MEMO_MIN_BELOW
#define TW_MEMO_MIN_BELOW 3
#endif
constexpr int kMemoMinBelow = TW_MEMO_MIN_BELOW;   // the transposition table is consulted / written only at nodes with more spans than this below: a sub-tree of the
// last levels costs less than a probe and a store (measured, nodejs shape: select 13.1 ms with 0, 10.5 with 3, 58.9 with 6; media shape unchanged)
constexpr int kMatchMinDepth = 4;    // ... at nodes with at least this many spans below
// The table lives in the LDS layout of the window class, and the layouts of the small classes are mostly table: 64 entries
// for windows of up to 16 spans let more workgroups share a CU (media shape, x16: select 3.64 -> 3.01 ms; 32 / 128 entries
// the same within noise), while the 17..32-span layout needs its 256 (128: one nodejs window grows to 190 ms; 512 .. 2048:
// the per-window sweep of the table costs more than it prunes, media select 3.6 -> 3.9 ms).
#ifndef TW_MEMO_MID
#define TW_MEMO_MID 64     // windows of up to 8 spans
#endif
#ifndef TW_MEMO_BIG
#define TW_MEMO_BIG 64     // 9..16 spans
#endif
constexpr int kMemoSlots = TW_MEMO_SLOTS;      // transposition table entries (LDS); a table of 2^20 entries prunes no better on the test workloads

// ---- cooperative path: one workgroup per window --------------------------------------------------
// MW = spans a window may hold; SEARCH = false leaves out what only select_search needs (windows of <= kBruteMax spans
// have no larger component: k_select_tiny, 1 KB instead of 22 KB of LDS per wavefront)
// DFS = false leaves out what only the depth-first search needs (transposition table, matching relaxation): the layouts of
// k_select_heavy, whose components are solved level by level (select_dp) or handed to k_select_dp -- more workgroups per CU
// EMAX: most endpoints of the units the layout serves -- the candidates' span indices are a fifth to a third of a layout, and the
// selection kernels hold as many windows at a time as their LDS lets them (k_select_heavy on the three-word layout: 18 KB with room
// for eight endpoints, 14 KB for two; the launches pick the instantiation by the tile set's largest endpoint count)
template <int MW, bool SEARCH, bool DFS = true, int EMAX = kMaxEp>
struct SelectLdsT {
    static constexpr bool kSearch = SEARCH, kDfs = DFS;
    static constexpr int kEmax = EMAX;
    // transposition table entries by layout (TW_MEMO_MID / TW_MEMO_BIG: the windows of 5-7 and 8-15 spans)
    static constexpr int kS = SEARCH ? MW : 1, kSlots = !(SEARCH && DFS) ? 1 : (MW <= 8 ? TW_MEMO_MID : (MW <= 16 ? TW_MEMO_BIG : kMemoSlots));
    static constexpr int kW = SEARCH ? (MW * kTopK + 63) / 64 : 1;   // words of a mask over all candidates of a component
    int32_t idx[MW][kTopK][EMAX];
    sel_w w[MW][kTopK];  // sel_weight(score); <= 0 means not eligible
    sel_w ub[kS + 1];
    sel_w red_val[kCoop / 64];
    int32_t red_idx[kCoop / 64];
    uint8_t ncand[MW], comp[MW], mem[MW];
    int8_t pick[MW];
    uint32_t adj[MW];  // span conflict relation as bit rows
    // conflict relation through an occupancy table: per endpoint one word per outgoing-span index in [occ_lo, occ_hi], bit b set <=>
    // span b has an eligible candidate that uses that outgoing span (windows whose index ranges add up to <= kOcc words)
    static constexpr int kOcc = SEARCH ? 256 : 1;
    uint32_t occ[kOcc];
    int32_t occ_lo[kMaxEp], occ_hi[kMaxEp], occ_off[kMaxEp + 1];
    uint32_t cmask[kBruteMax][kTopK], celig[kBruteMax];  // select_brute: conflicts among / eligibility of the component's candidates
    unsigned long long g2[kS], g3[kS];  // pair / triple optima of the grouped bound (weights are > 0)
    // select_search: conflicts of candidate k of member x with the candidates of the other members as a bit
    // mask over (member y, candidate k2) -> bit y*kTopK+k2
    unsigned long long cmask3[kS][kTopK][kW];
    // transposition table of select_search: what can still be gained below a node depends only on its depth and on which
    // candidates of the remaining spans are blocked, not on how the spans above were assigned
    struct Memo {      // one entry: read whole by a probe (independent loads, one wait)
        unsigned long long key[kW];
        sel_w val;
        unsigned int state;   // generation << 2 | (0 empty, 2 valid)
    } memo[kSlots];
    unsigned int memo_gen;
    int cm, budget_hit;
    unsigned long long nodes_total;   // search nodes of the window (all lanes), reported in unit_stats[5]
#ifdef TW_PROFILE_SEL
    long long pt[6];
#endif
    // matching relaxation (select_match_prunes): columns = outgoing spans of one endpoint addressed by index - base, then
    // one "unassigned" column per row
    static constexpr int kCols = (SEARCH && DFS) ? MW * 4 : 1;   // (candidates of a component lie close together: its spans overlap in time)
    sel_w hu[kS + 1], hv[kCols + 1], hminv[kCols + 1];
    uint8_t hp[kCols + 1], hway[kCols + 1], hused[kCols + 1], htouch[kCols + 1];
};
typedef SelectLdsT<kMaxWin, true> SelectLds;
typedef SelectLdsT<kBruteMax, false> SelectLdsTiny;
// k_select_heavy: the three window classes without the depth-first search's tables
template <int EMAX> using SelectLdsLvlE = SelectLdsT<kMaxWin, true, false, EMAX>;        // kHugeWindow .. kMaxWin spans: three-word masks
template <int EMAX> using SelectLdsBigLvlE = SelectLdsT<kHugeWindow, true, false, EMAX>; // windows of kBigWindow <= m < kHugeWindow spans: two-word masks
template <int EMAX> using SelectLdsMidLvlE = SelectLdsT<kBigWindow, true, false, EMAX>;  // windows of kBruteMax < m < kBigWindow spans: one-word masks
typedef SelectLdsLvlE<kMaxEp> SelectLdsLvl;
typedef SelectLdsBigLvlE<kMaxEp> SelectLdsBigLvl;
typedef SelectLdsMidLvlE<kMaxEp> SelectLdsMidLvl;

template <class LDS>
__device__ inline bool lds_share(const LDS& L, int E, int b1, int k1, int b2, int k2) {
    for (int e = 0; e < E; e++)
        if (L.idx[b1][k1][e] == L.idx[b2][k2][e]) return true;
    return false;
}

// Matching relaxation of the sub-problem "spans d..cm-1 of the component, given the candidates blocked so far" (one lane):
// keep the constraint of one endpoint only -- no outgoing span of endpoint e twice -- and drop the others.  What remains is
// a maximum-weight bipartite matching between the remaining spans and the outgoing spans of e, edge weight = the weight of
// the span's still-eligible candidate that uses it; a span may stay unassigned (a column of its own, weight 0).  Solved
// exactly by shortest augmenting paths with integer potentials (the Hungarian method; rows have <= kTopK + 1 entries, only
// the columns an augmentation touched are visited).  For one endpoint the bound is the exact optimum of the sub-problem; it is what
// ends the searches of windows in which more spans compete than outgoing spans are left -- the grouped bound of the
// suffix cannot see that.  (Any valid bound leaves the answer, the first selection of maximum weight in depth-first order,
// what it is.)
constexpr sel_w kNoBound = 0x1fffffffffffffffLL;   // select_match_bound: the endpoint's indices do not fit the column arrays
template <class LDS>
__device__ sel_w select_match_bound(LDS& L, int e, int d, unsigned long long b0, unsigned long long b1, unsigned long long b2) {
    constexpr int NC = LDS::kCols;
    const sel_w INF = kNoBound;
    const int cm = L.cm, nrow = cm - d;
    auto eligible = [&L, b0, b1, b2](int dd, int k) -> bool {   // (the masks by value: selected as values, not through pointers to them)
        const int b = L.mem[dd];
        if (k >= L.ncand[b] || !(L.w[b][k] > 0)) return false;
        const int bit = dd * kTopK + k;
        unsigned long long w = b2;
        if (bit < 128) w = b1;
        if (bit < 64) w = b0;
        return !((w >> (bit & 63)) & 1ull);
    };
    {
        int32_t base = 0x7fffffff, top = -0x7fffffff - 1;
        for (int r = 0; r < nrow; r++)
            for (int k = 0; k < kTopK; k++)
                if (eligible(d + r, k)) { const int32_t x = L.idx[L.mem[d + r]][k][e]; base = x < base ? x : base; top = x > top ? x : top; }
        if (top < base) { top = 0; base = 1; }
        const long long span = (long long)top - base + 1;
        if (span + nrow > NC - 1) return kNoBound;   // indices too far apart for the column arrays (ids are bytes): this endpoint gives no bound
        const int ncol = (int)span;       // columns 1..ncol: outgoing spans; ncol + 1 + r: span r stays unassigned
        for (int j = 0; j <= ncol + nrow; j++) { L.hv[j] = 0; L.hp[j] = 0; L.hminv[j] = INF; L.hused[j] = 0; }
        for (int i = 0; i <= nrow; i++) L.hu[i] = 0;
        for (int i = 1; i <= nrow; i++) {
            L.hp[0] = (uint8_t)i;
            int j0 = 0, ntouch = 0;
            do {
                L.hused[j0] = 1;
                const int i0 = L.hp[j0], dd = d + i0 - 1, b = L.mem[dd];
                const sel_w ui = L.hu[i0];
                for (int k = 0; k <= kTopK; k++) {   // the row's edges: its eligible candidates, then "unassigned"
                    int j;
                    sel_w a;
                    if (k < kTopK) {
                        if (!eligible(dd, k)) continue;
                        j = L.idx[b][k][e] - base + 1;
                        a = -L.w[b][k];
                    } else { j = ncol + i0; a = 0; }
                    if (L.hused[j]) continue;
                    const sel_w cur = a - ui - L.hv[j];
                    if (cur < L.hminv[j]) {
                        if (L.hminv[j] == INF) L.htouch[ntouch++] = (uint8_t)j;
                        L.hminv[j] = cur;
                        L.hway[j] = (uint8_t)j0;
                    }
                }
                sel_w delta = INF;
                int j1 = 0;
                for (int q = 0; q < ntouch; q++) {
                    const int j = L.htouch[q];
                    if (!L.hused[j] && L.hminv[j] < delta) { delta = L.hminv[j]; j1 = j; }
                }
                // (every row has its "unassigned" column, so an unused touched column always exists)
                L.hu[L.hp[0]] += delta;
                L.hv[0] -= delta;
                for (int q = 0; q < ntouch; q++) {
                    const int j = L.htouch[q];
                    if (L.hused[j]) { L.hu[L.hp[j]] += delta; L.hv[j] -= delta; }
                    else L.hminv[j] -= delta;
                }
                j0 = j1;
            } while (L.hp[j0] != 0);
            do { const int j1 = L.hway[j0]; L.hp[j0] = L.hp[j1]; j0 = j1; } while (j0);
            for (int q = 0; q < ntouch; q++) { const int j = L.htouch[q]; L.hminv[j] = INF; L.hused[j] = 0; }
            L.hused[0] = 0;
        }
        return L.hv[0];   // = -(minimum cost) = maximum weight of the relaxed sub-problem
    }
}
__device__ __forceinline__ int uni(int x);               // (the value of the first active lane; defined with the lane arrays below)
__device__ __forceinline__ long long uni(long long x);
// The same bound by the whole wavefront (every lane of the first wavefront calls, with wave-uniform arguments): lane j holds
// column j -- its potential, slack, tree flag, predecessor and matched row -- and lane i the potential of row i, in registers;
// one step of the shortest-augmenting-path search is a handful of broadcast row edges, one minimum over the lanes and one
// update of every lane's own column, instead of the serial walk over the touched columns in LDS (a matching of 18 rows:
// ~100 us in one lane, a tenth of that here).  The optimum of an assignment problem is unique, so both return the same bound
// whatever augmenting paths they take.  Needs rows + columns <= 63; otherwise (and in the one-thread emulation of the tests)
// the serial form runs in the first lane.
template <class LDS>
__device__ sel_w select_match_bound_wave(LDS& L, int e, int d, unsigned long long b0, unsigned long long b1, unsigned long long b2) {
    const int lane = (int)threadIdx.x & 63;
    const sel_w INF = kNoBound;
    const int cm = L.cm, nrow = cm - d;
    auto eligible = [&L, b0, b1, b2](int dd, int k) -> bool {
        const int b = L.mem[dd];
        if (k >= L.ncand[b] || !(L.w[b][k] > 0)) return false;
        const int bit = dd * kTopK + k;
        unsigned long long w = b2;
        if (bit < 128) w = b1;
        if (bit < 64) w = b0;
        return !((w >> (bit & 63)) & 1ull);
    };
    // the range of the endpoint's candidate indices: a row per lane, then the minimum / maximum over the lanes
    int32_t base = 0x7fffffff, top = -0x7fffffff - 1;
    if (lane < nrow)
        for (int k = 0; k < kTopK; k++)
            if (eligible(d + lane, k)) { const int32_t x = L.idx[L.mem[d + lane]][k][e]; base = x < base ? x : base; top = x > top ? x : top; }
    for (int off = 32; off >= 1; off >>= 1) {
        const int32_t ob = __shfl_xor(base, off), ot = __shfl_xor(top, off);
        base = ob < base ? ob : base; top = ot > top ? ot : top;
    }
    if (top < base) { top = 0; base = 1; }
    const long long span = (long long)top - base + 1;
    if (span + nrow > 63 || (int)blockDim.x < 64) {   // (uniform) too many columns for a lane each, or no lanes at all
        sel_w r = 0;
        if (threadIdx.x == 0) r = select_match_bound(L, e, d, b0, b1, b2);
        return uni((long long)r);
    }
    const int ncol = (int)span, m = ncol + nrow;   // columns 1..ncol: outgoing spans; ncol + i: row i stays unassigned; column 0: the search's root
#ifdef TW_DP_TRACE
    if (threadIdx.x == 0) printf("match_wave: %d rows, %d columns\n", nrow, m);
#endif
    sel_w v = 0, u = 0, minv = INF;
    int p = 0, way = 0;
    bool used = false, in_tree = false;
    for (int i = 1; i <= nrow; i++) {
        if (lane == 0) p = i;
        minv = INF; used = false; in_tree = false;
        int j0 = 0;
        while (true) {
            if (lane == j0) used = true;
            const int i0 = uni(__shfl(p, j0));
            if (lane == i0) in_tree = true;
            const sel_w ui = (sel_w)uni((long long)__shfl(u, i0));
            const int dd = d + i0 - 1, b = L.mem[dd];
            for (int k = 0; k <= kTopK; k++) {   // the row's edges: its eligible candidates, then "unassigned"
                int j;
                sel_w a;
                if (k < kTopK) {
                    if (!eligible(dd, k)) continue;   // (uniform)
                    j = L.idx[b][k][e] - base + 1;
                    a = -L.w[b][k];
                } else { j = ncol + i0; a = 0; }
                if (lane == j && !used) {
                    const sel_w cur = a - ui - v;
                    if (cur < minv) { minv = cur; way = j0; }
                }
            }
            // the unused column of least slack (every row has its "unassigned" column, so one always exists)
            sel_w best = (!used && lane >= 1 && lane <= m) ? minv : INF;
            for (int off = 32; off >= 1; off >>= 1) { const sel_w o = __shfl_xor(best, off); best = o < best ? o : best; }
            const sel_w delta = best;
            const unsigned long long at = __ballot(!used && lane >= 1 && lane <= m && minv == delta);
            const int j1 = __ffsll((long long)at) - 1;
            if (used) v -= delta; else if (minv != INF) minv -= delta;
            if (in_tree) u += delta;
            j0 = j1;
            if (uni(__shfl(p, j0)) == 0) break;
        }
        while (j0 != 0) {   // the augmenting path back to the root
            const int j1 = uni(__shfl(way, j0));
            const int pj = uni(__shfl(p, j1));
            if (lane == j0) p = pj;
            j0 = j1;
        }
    }
    return (sel_w)uni((long long)__shfl(v, 0));   // = -(minimum cost) = maximum weight of the relaxed sub-problem
}
// true when for some endpoint weight above + bound <= incumbent: nothing below the node can improve on it
template <class LDS>
__device__ bool select_match_prunes(LDS& L, int E, int d, sel_w acc, sel_w best_w, unsigned long long b0, unsigned long long b1, unsigned long long b2) {
    for (int e = 0; e < E; e++) {   // (called by every lane of the first wavefront: uniform arguments, uniform result)
        const sel_w bound = select_match_bound_wave(L, e, d, b0, b1, b2);
        if (bound != kNoBound && acc + bound <= best_w) return true;
    }
    return false;
}

// ---- a small array held across the lanes of one vector register --------------------------------------------------
// Element q lives in lane q; a read at a wave-uniform index is one v_readlane per 32 bits with its result in scalar
// registers, a write a compare and a select.  select_search keeps everything it indexes by depth this way: a walk whose every step
// waits for a chain of LDS loads in one lane (1.5 us per node, measured) becomes scalar code next to the register file.
// (The host emulation of the tests has no lanes to spread over: there every emulated lane holds the whole array.)
#ifdef TW_HOST_EMULATION
template <class T>
struct LaneArr {
    T v[64];
    template <class F> void load(int n, F f) { for (int q = 0; q < 64; q++) v[q] = q < n ? f(q) : T(0); }
    void fill(T x) { for (int q = 0; q < 64; q++) v[q] = x; }
    T get(int i) const { return v[i]; }
    void set(int i, T x) { v[i] = x; }
    template <class F> void store(int n, F f) const { if (threadIdx.x == 0) for (int q = 0; q < n; q++) f(q, v[q]); }
};
#else
template <class T>
struct LaneArr {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "one or two registers per lane");
    T v;
    template <class F> __device__ __forceinline__ void load(int n, F f) { v = (int)threadIdx.x < n ? f((int)threadIdx.x) : T(0); }
    __device__ __forceinline__ void fill(T x) { v = x; }
    __device__ __forceinline__ T get(int i) const {
        if constexpr (sizeof(T) == 4) return (T)__builtin_amdgcn_readlane((int)v, i);
        else {
            const unsigned long long x = (unsigned long long)v;
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, i), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), i);
            return (T)(((unsigned long long)hi << 32) | lo);
        }
    }
    __device__ __forceinline__ void set(int i, T x) { v = (int)threadIdx.x == i ? x : v; }   // (a compare and a select per register)
    template <class F> __device__ __forceinline__ void store(int n, F f) const { if ((int)threadIdx.x < n) f((int)threadIdx.x, v); }
};
#endif
// the value of the first active lane, for what the wavefront computed in one lane (or read from LDS at a uniform address)
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ long long uni(long long x) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)x), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)x >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// Depth-first search of a component of more than kBruteMax spans (see above): the wavefront walks the tree in the
// canonical order as one scalar thread of control -- all lanes take every branch together -- with what it indexes by depth
// in lane arrays (LaneArr: lane q <-> depth q): the spans' weights, eligibility and conflict masks, the suffix bound, and the
// stack (weight above every level, blocked candidates at every level: a return costs nothing).  A node is cut when weight +
// bound <= incumbent, the bound being the grouped bound of the suffix or the entry of the transposition table (LDS, lane 0)
// for (depth, blocked candidates of the remaining spans).  With the table the search is a dynamic
// programme over the sets of taken outgoing spans rather than over the assignments that produce them: the largest
// component of the heavy-load test workloads takes ~2e3 nodes instead of 4e6 (nodejs shape) / 8e8 (tie-saturated set).
// Needs L.cm, L.mem, L.ncand, L.w, L.ub, L.cmask3; writes L.pick of the members.
// node_cap > 0: a search that has not finished after that many nodes is given up (returns true, L.pick of the members is not
// written): the window goes to k_select_dp, which solves such components level by level over all lanes (select_dp).
template <class LDS>
__device__ bool select_search_body(LDS& L, int E, int node_cap, int match_after = kMatchNodes1) {
    constexpr int W = LDS::kW;   // words of the blocked mask: kept in up to three registers, the unused ones are constant zero
    static_assert(W >= 1 && W <= 3 && kBlkWords == 3, "the blocked mask is kept in three registers");
    static_assert(kMaxWin < 64 && kTopK <= 8, "one lane per depth, one bit per candidate");
    typedef unsigned long long ull;
    const int t = threadIdx.x;
    const int cm = uni(L.cm);
    // per depth: the member's candidates
    LaneArr<sel_w> wv[kTopK], ubv;
    LaneArr<ull> cmk[kTopK][W];
    LaneArr<int> elig, ncv;
#pragma unroll
    for (int k = 0; k < kTopK; k++) {
        wv[k].load(cm, [&](int q) { const int b = L.mem[q]; return (k < (int)L.ncand[b] && L.w[b][k] > 0) ? L.w[b][k] : (sel_w)0; });
#pragma unroll
        for (int w = 0; w < W; w++) cmk[k][w].load(cm, [&](int q) { return L.cmask3[q][k][w]; });
    }
    elig.load(cm, [&](int q) { const int b = L.mem[q]; int m = 0; for (int k = 0; k < (int)L.ncand[b]; k++) if (L.w[b][k] > 0) m |= 1 << k; return m; });
    ncv.load(cm, [&](int q) { return (int)L.ncand[L.mem[q]]; });
    ubv.load(cm + 1, [&](int q) { return L.ub[q]; });
    // the stack, the choice per level (ncand = "none") and the incumbent
    LaneArr<sel_w> sacc;
    LaneArr<ull> sb[W];
    LaneArr<int> cur, best;
    sacc.fill(0); cur.fill(-1); best.fill(-1);
#pragma unroll
    for (int w = 0; w < W; w++) sb[w].fill(0ull);
    int gen = 0;
    if (t == 0) {
        L.memo_gen++;   // entries of earlier components become stale without a sweep
        if ((L.memo_gen & 0x3fffffffu) == 0u) { L.memo_gen = 1u; for (int q = 0; q < LDS::kSlots; q++) L.memo[q].state = 0u; }
        gen = (int)L.memo_gen;
    }
    const unsigned int tag = ((unsigned)uni(gen) << 2) | 2u;
    // key of a node: the blocked candidates of the spans d.. (bits below d * kTopK cleared) with the depth in the low bits
    auto memo_key = [&](int d, ull b0, ull b1, ull b2, ull (&k)[kBlkWords]) -> unsigned {
        const int bit = d * kTopK, wd = bit >> 6;
        const ull keep = ~0ull << (bit & 63);
        k[0] = wd == 0 ? (b0 & keep) : 0ull; k[1] = wd == 1 ? (b1 & keep) : (wd < 1 ? b1 : 0ull); k[2] = wd == 2 ? (b2 & keep) : (wd < 2 ? b2 : 0ull);
        k[0] |= (ull)d;   // d >= 1: bits 0..4 belong to the first span and are clear
        ull h = (k[0] * 0x9E3779B97F4A7C15ull) ^ (k[1] * 0xC2B2AE3D27D4EB4Full) ^ (k[2] * 0x165667B19E3779F9ull);
        h ^= h >> 29;
        return (unsigned)h & (LDS::kSlots - 1);
    };
    auto key_eq = [&](const typename LDS::Memo& en, const ull (&k)[kBlkWords]) -> bool {
        bool eq = en.key[0] == k[0];
        if constexpr (W > 1) eq = eq && en.key[1] == k[1];
        if constexpr (W > 2) eq = eq && en.key[2] == k[2];
        return eq;
    };
    // the kTopK blocked bits of depth d (they may straddle two words)
    auto blocked_at = [&](int d, ull b0, ull b1, ull b2) -> unsigned {
        const int bit = d * kTopK, wd = bit >> 6, sh = bit & 63;
        const ull lo = wd == 0 ? b0 : (wd == 1 ? b1 : b2), hi = wd == 0 ? b1 : (wd == 1 ? b2 : 0ull);
        ull x = lo >> sh;
        if (sh > 64 - kTopK) x |= hi << (64 - sh);
        return (unsigned)x & ((1u << kTopK) - 1u);
    };
#ifdef TW_PROFILE_SEL
    const long long _s0 = wall_clock64();
#endif
    sel_w best_w = 0, acc = 0;      // only strict improvements replace the incumbent
    ull b0 = 0, b1 = 0, b2 = 0;
    int d = 0, nodes = 0;
    bool entered = true, over = false, capped = false;
    // One endpoint: a search that is still running after kMatchNodes1 nodes asks for the optimum of the whole component (one
    // matching) and from then on only looks for the first selection in depth-first order that reaches it: the incumbent is
    // floored at optimum - 1, every node is tested with the exact bound of its sub-problem, so the walk descends without
    // backtracking past a node that cannot reach the optimum, and ends at the first leaf that does.
    int exact = 0;   // 0 not asked yet, 1 known, -1 not available (indices too far apart)
    while (d >= 0) {
        int k = 0;
        if (entered) {
            if (++nodes > kNodeBudget) { over = true; break; }
            if (node_cap > 0 && nodes > node_cap) { capped = true; break; }
            if (E == 1 && exact == 0 && nodes > match_after) {
                sel_w opt = 0;
                opt = select_match_bound_wave(L, 0, 0, 0ull, 0ull, 0ull);
                exact = opt == kNoBound ? -1 : 1;
                if (exact == 1) {
                    if (best_w >= opt) break;   // the incumbent is the first selection of that weight in depth-first order
                    best_w = opt - 1;
                }
            }
            if (d == cm) {
                if (acc > best_w) {
                    best_w = acc; best = cur;
                    if (exact == 1) break;      // the optimum, first in depth-first order
                }
                d--; entered = false; continue;
            }
            bool cut = acc + ubv.get(d) <= best_w;
            if (!cut && d >= 1 && cm - d > kMemoMinBelow) {   // has the sub-tree below this (depth, blocked set) been searched already?
                ull kk[kBlkWords];
                const unsigned slot = memo_key(d, b0, b1, b2, kk);
                int hit = 0;
                if (t == 0)
                    for (int pr = 0; pr < 4; pr++) {
                        const typename LDS::Memo en = L.memo[(slot + pr) & (LDS::kSlots - 1)];
                        if (en.state != tag) break;   // empty: the chain ends here
                        if (key_eq(en, kk)) { hit = acc + en.val <= best_w; break; }
                    }
                cut = uni(hit) != 0;
            }
            // from kMatchNodes nodes on the matching relaxation is consulted as well (where enough spans remain below)
            if (!cut && (E == 1 ? exact == 1 : nodes > kMatchNodes) && cm - d >= kMatchMinDepth) {
#ifdef TW_PROFILE_SEL
                const long long _m0 = wall_clock64();
#endif
                int hit = 0;
                hit = select_match_prunes(L, E, d, acc, best_w, b0, b1, b2) ? 1 : 0;
                cut = hit != 0;
#ifdef TW_PROFILE_SEL
                if (t == 0) { L.pt[1] += wall_clock64() - _m0; L.pt[2] += 1; }
#endif
            }
            if (cut) { d--; entered = false; continue; }
        } else {
            acc = sacc.get(d); b0 = sb[0].get(d); if constexpr (W > 1) b1 = sb[1].get(d); if constexpr (W > 2) b2 = sb[2].get(d);
            k = cur.get(d) + 1;   // resume below the choice this level made last ("none" was stored as ncand)
        }
        // the next choice of this level: its first eligible, unblocked candidate from k on, then "none"
        const int nc = ncv.get(d);
        const unsigned open = k < kTopK ? ((unsigned)elig.get(d) & ~blocked_at(d, b0, b1, b2) & (~0u << k)) : 0u;
        if (open != 0u) k = __ffs((int)open) - 1;
        else if (k <= nc) k = nc;
        else k = -1;
        if (k >= 0) {
            cur.set(d, k);
            if (k < nc) {
#pragma unroll
                for (int c = 0; c < kTopK; c++)
                    if (c == k) { acc = acc + wv[c].get(d); b0 |= cmk[c][0].get(d); if constexpr (W > 1) b1 |= cmk[c][1].get(d); if constexpr (W > 2) b2 |= cmk[c][2].get(d); }
            }
            d++;
            sacc.set(d, acc); sb[0].set(d, b0); if constexpr (W > 1) sb[1].set(d, b1); if constexpr (W > 2) sb[2].set(d, b2);
            entered = true;
            continue;
        }
        // every way on from this node has been searched: remember what it can gain at most
        if (d >= 1 && cm - d > kMemoMinBelow) {
            ull kk[kBlkWords];
            const unsigned slot = memo_key(d, b0, b1, b2, kk);
            if (t == 0)
                for (int pr = 0; pr < 4; pr++) {
                    typename LDS::Memo& en = L.memo[(slot + pr) & (LDS::kSlots - 1)];
                    if (en.state == tag) {
                        if (key_eq(en, kk)) { en.val = best_w - acc; break; }
                        continue;
                    }
                    en.key[0] = kk[0]; if constexpr (W > 1) en.key[1] = kk[1]; if constexpr (W > 2) en.key[2] = kk[2]; en.val = best_w - acc;
                    en.state = tag;
                    break;
                }
        }
        d--; entered = false;
    }
    if (t == 0) {
#ifdef TW_PROFILE_SEL
        L.pt[0] += wall_clock64() - _s0; L.pt[3] += nodes;
#endif
        if (over) L.budget_hit = 1;
#ifdef TW_SEARCH_TRACE
        if (nodes > TW_SEARCH_TRACE) printf("search: component of %d spans, %d nodes\n", cm, nodes);
#endif
        L.nodes_total += (unsigned long long)nodes;
    }
    if (capped) return true;
    best.store(cm, [&](int q, int kq) {
        const int bq = L.mem[q];
        if (!(best_w > 0)) kq = -1;
        L.pick[bq] = (int8_t)((kq < 0 || kq == (int)L.ncand[bq]) ? -1 : kq);
    });
    return false;
}
template <class LDS>
__device__ bool select_search(LDS& L, int E, int node_cap = 0) {
    const bool capped = select_search_body(L, E, node_cap);
    group_sync();
    return capped;
}

// ---- level-synchronous search of a component (select_dp) --------------------------------------------------------------
// What can still be gained below a node of the canonical search depends only on its depth and on which candidates of the
// remaining spans are blocked (the transposition table of select_search says the same).  So the tree is a layered graph:
// level d holds the distinct *states* (blocked candidates of the spans d..) that some assignment of the spans 0..d-1 reaches,
// each with the best such assignment -- largest weight, among equal weights the one that the depth-first order visits first,
// i.e. the lexicographically smallest choice vector (candidates by list position, "none" last).  The first selection of
// maximum weight in depth-first order is then the survivor of the single state of level cm.  One level is expanded by all
// lanes at once: every (state, choice) pair is an item; children meet in a hash table in LDS (claim by a 64-bit tag, keys
// verified, atomicMax on the weight, atomicMin on the choice vector among the maxima: order-independent, so the result does
// not depend on the interleaving); a child that cannot reach the greedy leaf's weight is dropped (weight + grouped bound of
// the suffix < that weight: strictly below a feasible selection, so never part of a maximum).  Exact integers throughout.
// With that bound most levels hold one or two states (measured: the 31-span components of the nodejs shape that take the
// depth-first search 5-9 x 10^3 nodes hold 1 405 states over all levels).  A level that outgrows the table (or a tag collision
// between different keys) returns false: k_select_heavy (tables of kDpCapSmall states, one wavefront) then lists the window for
// k_select_dp (kDpCap states, kDpThreads lanes), and that one falls back to select_search.
// (k_select_dp: 384 states a level = 47 KB of tables + the window's 22 KB: two workgroups per CU.  768 / 1024 -- one per CU -- was
// measured: the nodejs shape's selection 2.56 instead of 2.35 ms per launch, 20.1 instead of 19.2 ms with 14 M spans resident; the
// few levels between 384 and 768 states go through the guessed thresholds instead.)
#ifndef TW_DP_CAP
#define TW_DP_CAP 384
#endif
#ifndef TW_DP_SLOTS
#define TW_DP_SLOTS 512
#endif
#ifndef TW_DP_CAP_SMALL
#define TW_DP_CAP_SMALL 32
#endif
#ifndef TW_DP_SLOTS_SMALL
#define TW_DP_SLOTS_SMALL 64
#endif
constexpr int kDpCap = TW_DP_CAP, kDpSlots = TW_DP_SLOTS;                    // k_select_dp: states per level, hash table slots
constexpr int kDpCapSmall = TW_DP_CAP_SMALL, kDpSlotsSmall = TW_DP_SLOTS_SMALL;   // k_select_heavy
constexpr int kDpThreads = 256;
constexpr unsigned long long kDpSlack0 = 16ull << 32;   // the first guess of select_dp's threshold: the sum of the best weights less 16.0 in score units
constexpr int kDpDigit = 3, kDpNone = 7, kDpWord0 = 21;   // choice vector: 3 bits a level, levels 0..20 in the first word (earlier = more significant)
static_assert(kTopK < kDpNone && kMaxWin <= 2 * kDpWord0, "choice vector in two words");
template <int W, int CAP, int SLOTS>
struct SelectDpT {
    typedef unsigned long long ull;
    static constexpr int kW = W, kCap = CAP, kSlots = SLOTS;
    static constexpr int kGuesses = CAP > TW_DP_CAP_SMALL ? 5 : 0;   // guessed thresholds after the greedy leaf's (the small tables of k_select_heavy: none, the window goes to k_select_dp)
    static_assert((SLOTS & (SLOTS - 1)) == 0 && SLOTS > CAP, "open addressing needs a free slot");
    ull ck[W][CAP], cacc[CAP], cp[2][CAP];                    // the states of the current level
    ull th[SLOTS], tk[W][SLOTS], tacc[SLOTS], tp[2][SLOTS];   // the next level as a hash table
    ull lb;
    int n_cur, n_next, n_claim, fail, fail2;   // fail: set in pass 1 (a level outgrew the table), fail2: in pass 2 (two keys with one tag)
};
struct SelectDpNone { static constexpr int kW = 0; };   // select_window_coop without the level-by-level solver (k_select_tiny, skip mode)
typedef SelectDpT<kBlkWords, kDpCap, kDpSlots> SelectDp;

template <class LDS, class DP>
__device__ bool select_dp(LDS& L, DP& D, int E) {
    typedef unsigned long long ull;
    constexpr int W = LDS::kW;
    static_assert(W == DP::kW && W >= 1 && W <= 3, "tables of the layout's mask width");
    const int t = threadIdx.x, nt = blockDim.x;
    const int cm = L.cm;
    if (t == 0) {   // the greedy leaf: the first leaf of the depth-first order
        ull b[W], acc = 0ull;
        for (int w = 0; w < W; w++) b[w] = 0ull;
        for (int d = 0; d < cm; d++) {
            const int bm = L.mem[d], bit = d * kTopK;
            for (int k = 0; k < (int)L.ncand[bm]; k++) {
                if (!(L.w[bm][k] > 0) || ((b[(bit + k) >> 6] >> ((bit + k) & 63)) & 1ull)) continue;
                acc += (ull)L.w[bm][k];
                for (int w = 0; w < W; w++) b[w] |= L.cmask3[d][k][w];
                break;
            }
        }
        D.lb = acc;
    }
    group_sync();
    // The threshold below which children are dropped: first the greedy leaf's weight (nearly always enough: one attempt).  Where
    // the levels outgrow the table with it -- the leaf can be two or three assignments short of the optimum, and then the levels
    // hold every state that has lost that much (measured on the nodejs shape: levels of 700 states with it, of 19 with the optimum
    // itself) -- the threshold is *guessed* instead: (sum of the spans' best weights) - slack, slack = 16, 256, 4096 in units of
    // the score (what second-best candidates cost), then 1/2, 3/2, ... average weights (what unassigned spans cost).  A guess
    // above the optimum kills every state at some level: raise the slack; a guess whose levels outgrow the table: halve the
    // distance to the largest slack that died; a guess that reaches level cm has kept, at every level, all states that can
    // reach its threshold <= optimum, i.e. every prefix of every maximum: its survivor is the answer.  Nothing between a dead
    // and an outgrown slack one score unit apart: given back.
    const ull lb_greedy = D.lb, ub_all = (ull)L.ub[0], unit = ub_all / (ull)(cm > 0 ? cm : 1) + 1ull;
    const ull slack_greedy = ub_all > lb_greedy ? ub_all - lb_greedy : 0ull;
    ull slack = slack_greedy, slack_dead = 0ull, slack_over = slack_greedy;   // died at <= slack_dead, outgrew the table at >= slack_over
    unsigned long long states = 0ull;
    bool dead = false, gave_up = false, exact_asked = false;
#ifdef TW_DP_TRACE
    int trace_max = 0, attempts = 0;
#endif
    for (int attempt = 0;; attempt++) {
    const ull lb = ub_all - slack;
    group_sync();   // (every lane has read the last attempt's D.n_cur)
    if (t == 0) {
        D.n_cur = 1; D.fail = 0; D.fail2 = 0;
        for (int w = 0; w < W; w++) D.ck[w][0] = 0ull;
        D.cacc[0] = 0ull; D.cp[0][0] = 0ull; D.cp[1][0] = 0ull;
    }
    group_sync();
    dead = false;
#ifdef TW_DP_TRACE
    attempts++;
#endif
    struct Child { ull k[3], acc, p[2], tag; unsigned slot; bool ok; };
    for (int d = 0; d < cm; d++) {
        for (int q = t; q < DP::kSlots; q += nt) { D.th[q] = 0ull; D.tacc[q] = 0ull; D.tp[0][q] = ~0ull; D.tp[1][q] = ~0ull; }
        if (t == 0) { D.n_next = 0; D.n_claim = 0; }
        group_sync();
        const int S = D.n_cur, bm = L.mem[d], nc = (int)L.ncand[bm];
        const int bit = d * kTopK, nbit = bit + kTopK, nwd = nbit >> 6;
        const ull keep = ~0ull << (nbit & 63), ubn = (ull)L.ub[d + 1];
        const int items = S * (kTopK + 1);
        states += (unsigned long long)S;
#ifdef TW_DP_TRACE
        if (t == 0 && S > trace_max) trace_max = S;
#endif
        // child c of state s: candidate c (c < kTopK; must be eligible and not blocked) or "none" (c == kTopK)
        auto child = [&](int item, Child& C) {
            const int s = item / (kTopK + 1), c = item % (kTopK + 1);
            ull k0 = D.ck[0][s], k1 = 0ull, k2 = 0ull;
            if constexpr (W > 1) k1 = D.ck[1][s];
            if constexpr (W > 2) k2 = D.ck[2][s];
            C.acc = D.cacc[s];
            C.ok = true;
            if (c < kTopK) {
                const int pos = bit + c;
                const ull word = (pos >> 6) == 0 ? k0 : ((pos >> 6) == 1 ? k1 : k2);
                C.ok = c < nc && L.w[bm][c] > 0 && !((word >> (pos & 63)) & 1ull);
                if (!C.ok) return;
                C.acc += (ull)L.w[bm][c];
                k0 |= L.cmask3[d][c][0];
                if constexpr (W > 1) k1 |= L.cmask3[d][c][1];
                if constexpr (W > 2) k2 |= L.cmask3[d][c][2];
            }
            if (C.acc + ubn < lb) { C.ok = false; return; }
            C.k[0] = nwd == 0 ? (k0 & keep) : 0ull;
            C.k[1] = nwd == 1 ? (k1 & keep) : (nwd < 1 ? k1 : 0ull);
            C.k[2] = nwd == 2 ? (k2 & keep) : (nwd < 2 ? k2 : 0ull);
            const ull digit = c < kTopK ? (ull)c : (ull)kDpNone;
            C.p[0] = D.cp[0][s]; C.p[1] = D.cp[1][s];
            if (d < kDpWord0) C.p[0] |= digit << ((kDpWord0 - 1 - d) * kDpDigit);
            else C.p[1] |= digit << ((2 * kDpWord0 - 1 - d) * kDpDigit);
            ull h = (C.k[0] * 0x9E3779B97F4A7C15ull) ^ (C.k[1] * 0xC2B2AE3D27D4EB4Full) ^ (C.k[2] * 0x165667B19E3779F9ull);
            h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
            C.slot = (unsigned)h & (DP::kSlots - 1);
            C.tag = h | 1ull;
        };
        auto key_is = [&](int q, const Child& C) -> bool {
            bool eq = D.tk[0][q] == C.k[0];
            if constexpr (W > 1) eq = eq && D.tk[1][q] == C.k[1];
            if constexpr (W > 2) eq = eq && D.tk[2][q] == C.k[2];
            return eq;
        };
        // the child's slot: the first slot of its probe sequence that carries its tag (slots are never released within a level)
        auto find = [&](const Child& C) -> int {
            unsigned q = C.slot;
            for (int pr = 0; pr < DP::kSlots; pr++, q = (q + 1) & (DP::kSlots - 1)) {
                const ull v = D.th[q];
                if (v == C.tag) return (int)q;
                if (v == 0ull) return -1;
            }
            return -1;
        };
        // pass 1: every child claims or finds the slot of its key
        for (int item = t; item < items; item += nt) {
            Child C;
            child(item, C);
            if (!C.ok) continue;
            unsigned q = C.slot;
            bool placed = false;
            for (int pr = 0; pr < DP::kSlots && !placed; pr++, q = (q + 1) & (DP::kSlots - 1)) {
                const ull old = atomicCAS(&D.th[q], 0ull, C.tag);
                if (old == 0ull) {
                    if (atomicAdd(&D.n_claim, 1) >= DP::kCap) D.fail = 1;
                    D.tk[0][q] = C.k[0];
                    if constexpr (W > 1) D.tk[1][q] = C.k[1];
                    if constexpr (W > 2) D.tk[2][q] = C.k[2];
                    placed = true;
                } else if (old == C.tag) placed = true;
                if (*(volatile int*)&D.fail) break;
            }
            if (!placed) D.fail = 1;
        }
        group_sync();
        if (D.fail) break;   // (uniform: read after the barrier)
        // pass 2: keys verified (two keys with one tag: give up), largest weight per state
        for (int item = t; item < items; item += nt) {
            Child C;
            child(item, C);
            if (!C.ok) continue;
            const int q = find(C);
            if (q < 0 || !key_is(q, C)) { D.fail2 = 1; continue; }   // (a flag of its own: a lane already in this pass must not change what a slower one reads after the barrier above)
            atomicMax(&D.tacc[q], C.acc);
        }
        group_sync();
        if (D.fail2) break;
        // passes 3 / 4: among the heaviest assignments of a state the one the depth-first order visits first
        for (int item = t; item < items; item += nt) {
            Child C;
            child(item, C);
            if (!C.ok) continue;
            const int q = find(C);
            if (q >= 0 && D.tacc[q] == C.acc) atomicMin(&D.tp[0][q], C.p[0]);
        }
        group_sync();
        if (d >= kDpWord0) {
            for (int item = t; item < items; item += nt) {
                Child C;
                child(item, C);
                if (!C.ok) continue;
                const int q = find(C);
                if (q >= 0 && D.tacc[q] == C.acc && D.tp[0][q] == C.p[0]) atomicMin(&D.tp[1][q], C.p[1]);
            }
            group_sync();
        }
        // the table becomes the list of the next level
        for (int q = t; q < DP::kSlots; q += nt) {
            if (D.th[q] == 0ull) continue;
            const int pos = atomicAdd(&D.n_next, 1);
            D.ck[0][pos] = D.tk[0][q];
            if constexpr (W > 1) D.ck[1][pos] = D.tk[1][q];
            if constexpr (W > 2) D.ck[2][pos] = D.tk[2][q];
            D.cacc[pos] = D.tacc[q]; D.cp[0][pos] = D.tp[0][q]; D.cp[1][pos] = d >= kDpWord0 ? D.tp[1][q] : 0ull;
        }
        group_sync();
        if (t == 0) D.n_cur = D.n_next;
        group_sync();
        if (D.n_cur == 0) { dead = true; break; }   // (uniform) the guess was above the optimum
    }
    if (!dead && D.fail == 0) break;            // reached level cm, or two keys with one tag (fail2)
    if (dead) slack_dead = slack; else slack_over = slack;
    if (E == 1 && DP::kGuesses > 0) {
        // one endpoint: the matching relaxation of the whole component is its optimum (select_match_bound; one lane, about as
        // long as fifty nodes of the depth-first search) -- the threshold that keeps nothing but prefixes of maxima
        if (exact_asked) { gave_up = true; break; }   // (the levels outgrow the table even so)
        exact_asked = true;
        if (t < 64) { const sel_w mo = select_match_bound_wave(L, 0, 0, 0ull, 0ull, 0ull); if (t == 0) D.lb = (ull)mo; }
        group_sync();
        const ull opt = D.lb;
        if (opt != (ull)kNoBound && opt <= ub_all && opt >= lb_greedy) { slack = ub_all - opt; continue; }
    }
    if (attempt >= DP::kGuesses) { gave_up = true; break; }   // (tie-saturated components outgrow the table at every slack that is not dead)
    {   // the next guess: the schedule's first slack above the largest dead one, or half way to the smallest outgrown one
        ull next = kDpSlack0;
        for (int j = 0; next <= slack_dead; j++) next = j < 2 ? next << 4 : (j == 2 ? unit / 2ull : next + unit);
        if (next >= slack_over) {
            if (slack_over - slack_dead <= (1ull << 32)) { gave_up = true; break; }
            next = slack_dead + (slack_over - slack_dead) / 2ull;
        }
        slack = next;
    }
    }
    if (t == 0) L.nodes_total += states;
    const bool failed = gave_up || dead || D.fail != 0 || D.fail2 != 0 || D.n_cur != 1;
#ifdef TW_DP_TRACE
    if (t == 0) printf("select_dp<%d>: component of %d spans, %llu states, widest level %d, attempts %d, %s\n", DP::kCap, cm, states, trace_max, attempts, failed ? "given back" : "solved");
#endif
    if (!failed) {
        const ull acc = D.cacc[0], p0 = D.cp[0][0], p1 = D.cp[1][0];
        for (int q = t; q < cm; q += nt) {
            const int digit = (int)((q < kDpWord0 ? p0 >> ((kDpWord0 - 1 - q) * kDpDigit) : p1 >> ((2 * kDpWord0 - 1 - q) * kDpDigit)) & 7ull);
            L.pick[L.mem[q]] = (int8_t)((acc > 0ull && digit != kDpNone) ? digit : -1);
        }
    }
    group_sync();
    return !failed;
}

// Complete enumeration of a component of cm <= kBruteMax spans by all lanes: combination index = the choices
// (candidate 0..kTopK-1, then kTopK = "none") as digits, first span most significant, so that increasing index is
// exactly the depth-first order of the canonical search; sums are accumulated left to right; the winner is the
// largest sum, the smallest index among equal sums, and must beat the empty selection (sum 0) strictly.  The
// canonical search visits at most 1 + 6 + ... + 6^4 = 1555 < kNodeBudget nodes on such a component, so it always
// completes and returns this very selection.
template <class LDS>
__device__ void select_brute(LDS& L, int E) {
    const int t = threadIdx.x, nt = blockDim.x;
    const int lane = t & 63, wave = t >> 6, nwave = (nt + 63) >> 6;
    constexpr int C = kTopK + 1;
    const int cm = L.cm;
    // conflicts among the component's <= 20 candidates as bit masks (bit y*kTopK+k2 of cmask[x][k]: candidate k of
    // member x shares an outgoing span with candidate k2 of member y), so that a combination is checked with one
    // mask per chosen candidate instead of 2E index comparisons per pair of chosen candidates
    for (int q = t; q < kBruteMax * kTopK + kBruteMax; q += nt) {
        if (q < kBruteMax * kTopK) L.cmask[q / kTopK][q % kTopK] = 0; else L.celig[q - kBruteMax * kTopK] = 0;
    }
    group_sync();
    for (int q = t; q < cm * kTopK * cm; q += nt) {
        const int x = q / (kTopK * cm), k = (q / cm) % kTopK, y = q % cm;
        const int bx = L.mem[x], by = L.mem[y];
        if (k >= L.ncand[bx] || !(L.w[bx][k] > 0)) continue;
        if (y == x) { atomicOr(&L.celig[x], 1u << k); continue; }
        uint32_t bits = 0;
        for (int k2 = 0; k2 < L.ncand[by]; k2++)
            if (L.w[by][k2] > 0 && lds_share(L, E, bx, k, by, k2)) bits |= 1u << (y * kTopK + k2);
        if (bits) atomicOr(&L.cmask[x][k], bits);
    }
    group_sync();
    int total = 1;
    for (int x = 0; x < cm; x++) total *= C;
    sel_w bsum = 0;
    int bidx = 0x7fffffff;
    for (int q = t; q < total; q += nt) {
        int rest = q;
        int ch[kBruteMax];
#pragma unroll
        for (int x = kBruteMax - 1; x >= 0; x--) {
            if (x < cm) { ch[x] = rest % C; rest /= C; } else ch[x] = kTopK;
        }
        bool ok = true;
        uint32_t sel = 0, clash = 0;
        sel_w sum = 0;
#pragma unroll
        for (int x = 0; x < kBruteMax; x++) {
            if (x < cm && ch[x] != kTopK) {
                ok = ok && ((L.celig[x] >> ch[x]) & 1u);
                sel |= 1u << (x * kTopK + ch[x]);
                clash |= L.cmask[x][ch[x]];
                sum += L.w[L.mem[x]][ch[x]];
            }
        }
        if (ok && (sel & clash) == 0 && sum > bsum) { bsum = sum; bidx = q; }  // own indices increase: the first maximum stays
    }
    for (int off = 32; off >= 1; off >>= 1) {
        if (off < nt) {
            const sel_w os = __shfl_down(bsum, off);
            const int oi = __shfl_down(bidx, off);
            if (os > bsum || (os == bsum && oi < bidx)) { bsum = os; bidx = oi; }
        }
    }
    if (nwave > 1) {
        group_sync();
        if (lane == 0) { L.red_val[wave] = bsum; L.red_idx[wave] = bidx; }
        group_sync();
        if (t == 0)
            for (int q = 1; q < nwave; q++)
                if (L.red_val[q] > bsum || (L.red_val[q] == bsum && L.red_idx[q] < bidx)) { bsum = L.red_val[q]; bidx = L.red_idx[q]; }
    }
    if (t == 0) {
        int rest = bidx;
        for (int x = cm - 1; x >= 0; x--) {
            const int k = bsum > 0 ? rest % C : kTopK;
            rest /= C;
            L.pick[L.mem[x]] = (int8_t)(k == kTopK ? -1 : k);
        }
    }
    group_sync();
}

// Optional phase timers of the selection kernel (build with -DTW_PROFILE_SEL; read back through tw_debug_profile):
// accumulated per wavefront in registers, flushed once when the wavefront retires.  Compiled out by default.
#ifdef TW_PROFILE_SEL
__device__ long long g_sel_acc[8];
#define TW_SEL_T0() long long _st = wall_clock64()
#define TW_SEL_TICK(k) do { const long long _n = wall_clock64(); sel_acc[k] += _n - _st; _st = _n; } while (0)
#define TW_SEL_DECL() long long sel_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define TW_SEL_FLUSH() do { if (threadIdx.x == 0) for (int _k = 0; _k < 8; _k++) atomicAdd((unsigned long long*)&P.prof[_k], (unsigned long long)sel_acc[_k]); } while (0)
#define TW_SEL_ARG , long long (&sel_acc)[8]
#define TW_SEL_PASS , sel_acc
#else
#define TW_SEL_T0() do {} while (0)
#define TW_SEL_TICK(k) do {} while (0)
#define TW_SEL_DECL() do {} while (0)
#define TW_SEL_FLUSH() do {} while (0)
#define TW_SEL_ARG
#define TW_SEL_PASS
#endif
// flush_nodes = false: the caller adds L.nodes_total to the unit's statistics itself (the window kernels: once per run of windows of
// one unit -- one same-address atomic per window kept every window's next loads waiting behind it)
// dp != nullptr: every searched component is first tried level by level (select_dp) on the tables *dp.  If a level outgrows them:
// dfs_fallback = false (k_select_heavy) ends the window unsolved -- returns true, nothing is written to P.chosen, the caller lists it
// for k_select_dp; dfs_fallback = true (k_select_dp) hands the component to the depth-first search (first wavefront).
template <class LDS, class DP = SelectDpNone>
__device__ bool select_window_coop(const Dev& P, const UnitDev& U, int unit, int first, int m, LDS& L TW_SEL_ARG, bool flush_nodes = true,
                                   DP* dp = nullptr, bool dfs_fallback = true) {
    const int t = threadIdx.x, nt = blockDim.x, E = U.E;
    TW_SEL_T0();
    for (int q = t; q < m * kTopK; q += nt) {
        const int b = q / kTopK, k = q % kTopK;
        // all loads of the candidate at once, from the list on all spans (nearly every span's current list): one memory round
        // trip; a span whose list was recomputed without the spans earlier windows took (rep) reads that one afterwards
        const int64_t g = U.in_off + first + b;
        const uint8_t rep = P.rep[g];
        int n = P.tk_n[g];
        double sc = P.tk_score[tks_index(U, k, first + b)];
        int32_t ix[kMaxEp];
#pragma unroll
        for (int e = 0; e < kMaxEp; e++) ix[e] = e < E ? P.tk_idx[tk_index(U, k, e, first + b)] : 0;
        if (rep) {
            n = P.tkr_n[g];
            sc = P.tkr_score[tks_index(U, k, first + b)];
#pragma unroll
            for (int e = 0; e < kMaxEp; e++) ix[e] = e < E ? P.tkr_idx[tk_index(U, k, e, first + b)] : 0;
        }
        L.w[b][k] = k < n ? sel_weight(sc) : 0;
#pragma unroll
        for (int e = 0; e < LDS::kEmax; e++) if (e < E) L.idx[b][k][e] = k < n ? ix[e] : -1 - q;
        if (k == 0) { L.ncand[b] = (uint8_t)n; L.comp[b] = (uint8_t)b; L.pick[b] = -1; }
    }
    if (t == 0) { L.budget_hit = 0; L.nodes_total = 0ull; }
#ifdef TW_PROFILE_SEL
    if (t == 0) for (int q = 0; q < 6; q++) L.pt[q] = 0;
#endif
    for (int b = t; b < m; b += nt) L.adj[b] = 0;
    group_sync();
    TW_SEL_TICK(0);
    // span conflict relation: b ~ c iff an eligible candidate of b shares an outgoing span with an eligible
    // candidate of c; one lane per pair
    // (a lane's chain of dependent LDS reads is what this phase costs: with few pairs the candidates of a pair are spread over
    // the lanes as well -- a window of four spans has 6 pairs for 64 lanes and up to 25 x E comparisons per pair)
    bool by_table = false;
    if constexpr (LDS::kSearch) {
        if (m > 2 * kBruteMax) {   // (long windows: one lane per pair walks 25 x E comparisons -- 20-50 us for 17 spans, measured)
            for (int e = t; e < E; e += nt) { L.occ_lo[e] = 0x7fffffff; L.occ_hi[e] = -0x7fffffff - 1; }
            group_sync();
            for (int q = t; q < m * kTopK; q += nt) {
                const int b = q / kTopK, k = q % kTopK;
                if (k >= L.ncand[b] || !(L.w[b][k] > 0)) continue;
                for (int e = 0; e < E; e++) { atomicMin(&L.occ_lo[e], L.idx[b][k][e]); atomicMax(&L.occ_hi[e], L.idx[b][k][e]); }
            }
            group_sync();
            if (t == 0) {
                long long at = 0;
                for (int e = 0; e < E; e++) {
                    L.occ_off[e] = (int32_t)at;
                    if (L.occ_hi[e] >= L.occ_lo[e]) at += (long long)L.occ_hi[e] - L.occ_lo[e] + 1;
                    if (at > LDS::kOcc) at = LDS::kOcc + 1;
                }
                L.occ_off[E] = (int32_t)at;
            }
            group_sync();
            by_table = L.occ_off[E] <= LDS::kOcc;   // (uniform)
            if (by_table) {
                for (int q = t; q < L.occ_off[E]; q += nt) L.occ[q] = 0u;
                group_sync();
                for (int q = t; q < m * kTopK; q += nt) {
                    const int b = q / kTopK, k = q % kTopK;
                    if (k >= L.ncand[b] || !(L.w[b][k] > 0)) continue;
                    for (int e = 0; e < E; e++) atomicOr(&L.occ[L.occ_off[e] + (L.idx[b][k][e] - L.occ_lo[e])], 1u << b);
                }
                group_sync();
                for (int q = t; q < m * kTopK; q += nt) {
                    const int b = q / kTopK, k = q % kTopK;
                    if (k >= L.ncand[b] || !(L.w[b][k] > 0)) continue;
                    uint32_t others = 0u;
                    for (int e = 0; e < E; e++) others |= L.occ[L.occ_off[e] + (L.idx[b][k][e] - L.occ_lo[e])];
                    others &= ~(1u << b);
                    if (others) atomicOr(&L.adj[b], others);
                }
            }
        }
    }
    if (by_table) {
    } else if (m * (m - 1) / 2 * kTopK * kTopK <= 4 * nt) {   // one (pair, candidate, candidate) per lane
        for (int q = t; q < m * m * kTopK * kTopK; q += nt) {
            const int kb = q % kTopK, ka = (q / kTopK) % kTopK, c = (q / (kTopK * kTopK)) % m, b = q / (kTopK * kTopK * m);
            if (c >= b || ka >= L.ncand[b] || kb >= L.ncand[c] || !(L.w[b][ka] > 0) || !(L.w[c][kb] > 0)) continue;
            if (lds_share(L, E, b, ka, c, kb)) { atomicOr(&L.adj[b], 1u << c); atomicOr(&L.adj[c], 1u << b); }
        }
    } else if (m * (m - 1) / 2 * kTopK <= 4 * nt) {   // one (pair, candidate) per lane
        for (int q = t; q < m * m * kTopK; q += nt) {
            const int ka = q % kTopK, c = (q / kTopK) % m, b = q / (kTopK * m);
            if (c >= b || ka >= L.ncand[b] || !(L.w[b][ka] > 0)) continue;
            bool hit = false;
            for (int kb = 0; kb < L.ncand[c] && !hit; kb++)
                if (L.w[c][kb] > 0 && lds_share(L, E, b, ka, c, kb)) hit = true;
            if (hit) { atomicOr(&L.adj[b], 1u << c); atomicOr(&L.adj[c], 1u << b); }
        }
    } else
    for (int q = t; q < m * m; q += nt) {
        const int b = q / m, c = q % m;
        if (c >= b) continue;
        bool hit = false;
        for (int ka = 0; ka < L.ncand[b] && !hit; ka++) {
            if (!(L.w[b][ka] > 0)) continue;
            for (int kb = 0; kb < L.ncand[c] && !hit; kb++)
                if (L.w[c][kb] > 0 && lds_share(L, E, b, ka, c, kb)) hit = true;
        }
        if (hit) { atomicOr(&L.adj[b], 1u << c); atomicOr(&L.adj[c], 1u << b); }
    }
    group_sync();
    TW_SEL_TICK(1);
    if (t == 0) {  // connected components, labelled by their smallest member
        uint32_t seen = 0;
        for (int b = 0; b < m; b++) {
            if ((seen >> b) & 1) continue;
            uint32_t members = 1u << b, frontier = 1u << b;
            while (frontier) {
                const int v = __ffs((int)frontier) - 1;
                frontier &= frontier - 1;
                const uint32_t fresh = L.adj[v] & ~members;
                members |= fresh;
                frontier |= fresh;
            }
            seen |= members;
            for (int v = 0; v < m; v++) if ((members >> v) & 1) L.comp[v] = (uint8_t)b;
        }
    }
    group_sync();
    TW_SEL_TICK(2);
    bool hard = false;
    for (int root = 0; root < m; root++) {
        if (L.comp[root] != root) continue;  // uniform: comp is in LDS and stable here
        TW_SEL_TICK(6);
        if (t == 0) {
            int cm = 0;
            for (int b = root; b < m; b++) if (L.comp[b] == root) L.mem[cm++] = (uint8_t)b;
            L.cm = cm;
            // (the rest prepares the search of a component too large for complete enumeration)
            for (int d = 0; d < cm && cm > kBruteMax; d++) { L.g2[d] = 0; L.g3[d] = 0; }
        }
        group_sync();
        TW_SEL_TICK(3);
        if (L.cm <= kBruteMax) { select_brute(L, E); TW_SEL_TICK(4); continue; }  // uniform: cm is in LDS and stable here
        if constexpr (LDS::kSearch) {
        {   // conflict masks of the component's candidates (only between members whose candidate lists meet at all)
            const int cm = L.cm;
            for (int q = t; q < cm * kTopK * LDS::kW; q += nt) (&L.cmask3[0][0][0])[q] = 0;
            group_sync();
            for (int q = t; q < cm * kTopK * cm; q += nt) {
                const int x = q / (kTopK * cm), k = (q / cm) % kTopK, y = q % cm;
                const int bx = L.mem[x], by = L.mem[y];
                if (y == x || !((L.adj[bx] >> by) & 1u) || k >= L.ncand[bx] || !(L.w[bx][k] > 0)) continue;
                for (int k2 = 0; k2 < L.ncand[by]; k2++)
                    if (L.w[by][k2] > 0 && lds_share(L, E, bx, k, by, k2)) {
                        const int bit = y * kTopK + k2;
                        atomicOr(&L.cmask3[x][k][bit >> 6], 1ull << (bit & 63));
                    }
            }
        }
        // Upper bound of a suffix d..cm-1 of the component.  The level-by-level solver runs on the sum of the spans' best weights
        // (it keeps states by their blocked sets; the tighter bound below saves it 5-10 % of its states and costs more than that);
        // the depth-first search gets the grouped bound: the suffix cut into groups of 1-3 consecutive spans, every group solved
        // exactly on its own (conflicts inside the group only), the cheapest cutting.  It sees spans that compete for the same
        // outgoing spans (one of them must stay unassigned: -10000), which the sum of best weights does not.  All pairs and
        // triples are evaluated at once, one (group, combination) per lane; weights are positive integers.
        auto suffix_bound = [&](bool grouped) {
            if (t == 0) {
                const int cm = L.cm;
                L.ub[cm] = 0;
                for (int d = cm - 1; d >= 0; d--) {
                    sel_w mx = 0;
                    for (int k = 0; k < L.ncand[L.mem[d]]; k++) if (L.w[L.mem[d]][k] > mx) mx = L.w[L.mem[d]][k];
                    sel_w u = mx + L.ub[d + 1];
                    if (grouped && d + 2 <= cm) { const sel_w c2 = (sel_w)L.g2[d] + L.ub[d + 2]; if (c2 < u) u = c2; }
                    if (grouped && d + 3 <= cm) { const sel_w c3 = (sel_w)L.g3[d] + L.ub[d + 3]; if (c3 < u) u = c3; }
                    L.ub[d] = u;
                }
            }
            group_sync();
        };
        bool solved = false;
        if constexpr (DP::kW != 0) {
            if (dp != nullptr) {
                group_sync();
                suffix_bound(false);
                TW_SEL_TICK(3);
                solved = select_dp(L, *dp, E);
            }
        }
        if (!solved) {
            if (!LDS::kDfs || (dp != nullptr && !dfs_fallback)) { hard = true; break; }   // (uniform)
            if constexpr (LDS::kDfs) {   // (the layouts of k_select_heavy carry neither the search's tables nor its code)
            if (dp != nullptr && t == 0) atomicAdd((unsigned long long*)&P.unit_stats[(int64_t)unit * 8 + 7], 1ull);
            {
                const int cm = L.cm, npair = cm > 1 ? cm - 1 : 0, ntrip = cm > 2 ? cm - 2 : 0;
                constexpr int C = kTopK + 1;  // choices per span: a candidate or "none"
                const int total = npair * C * C + ntrip * C * C * C;
                for (int q = t; q < total; q += nt) {
                    int d, g, ch[3];
                    if (q < npair * C * C) { d = q / (C * C); const int c = q % (C * C); ch[0] = c / C; ch[1] = c % C; ch[2] = kTopK; g = 2; }
                    else { const int r = q - npair * C * C; d = r / (C * C * C); const int c = r % (C * C * C); ch[0] = c / (C * C); ch[1] = (c / C) % C; ch[2] = c % C; g = 3; }
                    bool ok = true;
                    sel_w sum = 0;
                    for (int x = 0; x < g && ok; x++) {
                        if (ch[x] == kTopK) continue;  // "none"
                        const int b = L.mem[d + x];
                        if (ch[x] >= L.ncand[b] || !(L.w[b][ch[x]] > 0)) { ok = false; break; }
                        for (int y = 0; y < x && ok; y++)
                            if (ch[y] != kTopK && lds_share(L, E, L.mem[d + y], ch[y], b, ch[x])) ok = false;
                        sum += L.w[b][ch[x]];
                    }
                    if (ok && sum > 0) atomicMax(g == 2 ? &L.g2[d] : &L.g3[d], (unsigned long long)sum);
                }
            }
            group_sync();
            suffix_bound(true);
            if (t < 64) select_search_body(L, E, 0, dp != nullptr ? 0 : kMatchNodes1);   // (behind the level-by-level solver: a hard one, ask for the optimum at once)
            group_sync();
            }
        }
        TW_SEL_TICK(5);
        }
    }
    TW_SEL_TICK(6);
    if (!hard) for (int b = t; b < m; b += nt) P.chosen[U.in_off + first + b] = L.pick[b];
    if (t == 0 && L.budget_hit) atomicAdd((unsigned long long*)&P.unit_stats[(int64_t)unit * 8 + 4], 1ull);
    if (flush_nodes && t == 0 && L.nodes_total) atomicAdd((unsigned long long*)&P.unit_stats[(int64_t)unit * 8 + 5], L.nodes_total);
    group_sync();
    TW_SEL_TICK(7);
    return hard;
}

// Puts window w of the unit on a work list (`listed` lanes only; every lane of the wavefront calls): windows of up to
// kBruteMax spans -- nearly all -- on the list of k_select_tiny, the others on the list of k_select_heavy.  There, long windows
// can take a thousand times longer than short ones: they are listed from the front and served first, the short ones from the
// back of the same array, so that no long search starts when the kernel is about to drain.
constexpr int kHardCount = 4, kHardNext = 5;   // P.heavy_next[kHardCount]: windows listed for k_select_dp, [kHardNext]: its work cursor
constexpr int kSelSeg = 32;      // segments of the selection work lists (one counter each)
constexpr int kCtrStride = 32;   // ints between two counters: a cache line each
__device__ __forceinline__ int sel_first_tile(int s, int n_tiles) { return (int)(((long long)s * n_tiles + kSelSeg - 1) / kSelSeg); }  // segment s = tiles [this, next)
constexpr int kSelSlots = kMaxEp + 1;   // tile sets with selection lists of their own: all tiles (the repair rounds), the classes (the first solve)
__device__ __forceinline__ int32_t* sel_counter(const Dev& P, const TileSet& S, int list, int s) { return P.heavy_count + ((S.slot * 4 + list) * kSelSeg + s) * kCtrStride; }
__device__ __forceinline__ int32_t* sel_cursor(const Dev& P, const TileSet& S, int list) { return P.heavy_next + S.slot * 8 + list; }
__device__ __forceinline__ int hard_base(const Dev& P, const TileSet& S) { return (int)((long long)S.base * P.tile_spans / (kBruteMax + 1)) + S.slot; }   // the set's share of hard_unit / hard_win
// (called by the per-span kernels, one workgroup per tile; `at` = the tile's place in the set S)
__device__ __forceinline__ void list_window(const Dev& P, const TileSet& S, int at, const UnitDev& U, int unit, int w, bool listed) {
    int cls = -1;   // 0 short (<= kBruteMax), 1 long (kBigWindow ..), 2 middle, 3 very long (kHugeWindow ..)
    int first = 0, m = 0;
    if (listed) {
        first = w == 0 ? 0 : P.w_last[U.in_off + w - 1] + 1;
        m = P.w_last[U.in_off + w] - first + 1;
        cls = m <= kBruteMax ? 0 : (m < kBigWindow ? 2 : (m < kHugeWindow ? 1 : 3));
    }
    const int s = (int)((long long)at * kSelSeg / S.n);
    const int s0 = wave_append(sel_counter(P, S, 0, s), cls == 0), s1 = wave_append(sel_counter(P, S, 1, s), cls == 1);
    const int s2 = wave_append(sel_counter(P, S, 2, s), cls == 2), s3 = wave_append(sel_counter(P, S, 3, s), cls == 3);
    if (!listed) return;
    const int base = (S.base + sel_first_tile(s, S.n)) * P.tile_spans, end = (S.base + sel_first_tile(s + 1, S.n)) * P.tile_spans;
    // two arrays, each filled from both ends of the segment: short windows and very long ones share tiny_*, long and middle ones heavy_*
    // (a segment has room for every window of its tiles)
    // (the lists of the repair rounds -- slot 0, positions by tile number -- have arrays of their own: a class' first round of the span
    // consumption lists into them while other classes are still working through the lists of their first solve, whose positions follow
    // the tiles' order by class.  In one pair of arrays those entries overwrote one another whenever units of different endpoint counts
    // alternate in the batch: a window was then solved twice and another not at all, differently from run to run)
    const bool rep = S.slot == 0;
    int32_t *au = (cls == 0 || cls == 3) ? (rep ? P.rtiny_unit : P.tiny_unit) : (rep ? P.rheavy_unit : P.heavy_unit);
    int32_t *aw = (cls == 0 || cls == 3) ? (rep ? P.rtiny_win : P.tiny_win) : (rep ? P.rheavy_win : P.heavy_win);
    const int pos = cls == 0 ? base + s0 : (cls == 1 ? base + s1 : (cls == 2 ? end - 1 - s2 : end - 1 - s3));
    // (the window as its first span and its size, 26 + 6 bits: the consumer's chain of dependent loads -- item, unit, window
    // bounds, candidates -- is what a short window costs)
    au[pos] = unit;
    aw[pos] = (int32_t)(((uint32_t)first << 6) | (uint32_t)(m - 1));
}

// Consumers: the segment counts of one list as prefix sums in LDS (first[kSelSeg] = the list's size), and the position of item i
struct SelSegs { int first[kSelSeg + 1]; };
__device__ __forceinline__ void sel_segments(const Dev& P, const TileSet& S, int list, SelSegs& G) {
    for (int s = threadIdx.x; s < kSelSeg; s += blockDim.x) G.first[s + 1] = *sel_counter(P, S, list, s);
    group_sync();
    if (threadIdx.x == 0) {
        G.first[0] = 0;
        for (int s = 0; s < kSelSeg; s++) G.first[s + 1] += G.first[s];
    }
    group_sync();
}
__device__ __forceinline__ int sel_position(const Dev& P, const TileSet& S, const SelSegs& G, int item, bool from_back) {
    int lo = 0, hi = kSelSeg - 1;   // the segment with first[s] <= item < first[s + 1]
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (G.first[mid] <= item) lo = mid; else hi = mid - 1; }
    const int off = item - G.first[lo];
    return from_back ? (S.base + sel_first_tile(lo + 1, S.n)) * P.tile_spans - 1 - off : (S.base + sel_first_tile(lo, S.n)) * P.tile_spans + off;
}

// Fast path, one lane per incoming span.  When the best candidates (list position 0) of a window's spans
// use pairwise different outgoing spans, "everybody takes position 0" is the selection the canonical search
// returns: it is the first leaf of the depth-first order (position 0 is tried first and nothing clashes), its
// weight equals the upper bound sum-of-best-weights, and only strict improvements replace the incumbent.
// A span whose best weight is <= 0 has no eligible candidate at all (lists are sorted) and stays unassigned
// in every selection.  ~90-99 % of the windows end here; the others are listed for k_select_heavy.
__global__ void __launch_bounds__(kTile) k_select_fast(Dev P, TileSet S) {
    if (*P.err != 0) return;  // an earlier kernel of this pass reported an error: its outputs are not usable
    const TileDev Tl = P.tiles[set_tile(S)];
    const UnitDev& U = P.units[Tl.unit];
    const int i = Tl.first + threadIdx.x;
    if (i >= U.n_in) return;
    const int64_t g = U.in_off + i;
    const bool ok = P.tk_n[g] > 0 && sel_weight(P.tk_score[tks_index(U, 0, i)]) > 0;
    P.chosen[g] = ok ? 0 : -1;
    const int w = P.wid[g];
    const int first = w == 0 ? 0 : P.w_last[U.in_off + w - 1] + 1;
    if (i - first >= kMaxWin) { raise_err(P, TW_ERR_WINDOW_SIZE); return; }
    if (!ok || first >= i) return;
    uint32_t elig = 0;
    for (int j = first; j < i; j++)
        if (P.tk_n[U.in_off + j] > 0 && sel_weight(P.tk_score[tks_index(U, 0, j)]) > 0) elig |= 1u << (j - first);
    bool clash = false;
    for (int e = 0; e < U.E && !clash; e++) {
        const int32_t mine = P.tk_idx[tk_index(U, 0, e, i)];
        for (uint32_t rest = elig; rest != 0 && !clash; rest &= rest - 1)
            clash = P.tk_idx[tk_index(U, 0, e, first + __ffs((int)rest) - 1)] == mine;
    }
    // the first lane that finds a clash puts the window on the work list of k_select_heavy
    const bool listed = clash && atomicExch(&P.w_conf[U.in_off + w], 1) == 0;
    list_window(P, S, (int)blockIdx.x, U, Tl.unit, w, listed);
}

// LIST 1: the long windows (kBigWindow .. kHugeWindow - 1 spans; front of the segments of heavy_*; two-word masks, 12 KB of LDS),
// LIST 2: the middle ones (back of those segments; one word, 7 KB: 21 wavefronts per CU), LIST 3: the very long ones (back of the
// segments of tiny_*; the full layout: 22 KB, 7 wavefronts per CU)
template <class LDS, int LIST>
__global__ void __launch_bounds__(64) k_select_heavy(Dev P, TileSet S) {  // persistent workgroups, one deferred window at a time
    if (*P.err != 0) return;  // an earlier kernel of this pass reported an error: its outputs are not usable
    __shared__ LDS L;
    __shared__ SelectDpT<LDS::kW, kDpCapSmall, kDpSlotsSmall> D;
    __shared__ int next_item;
    __shared__ SelSegs G;
    sel_segments(P, S, LIST, G);
    const int count = G.first[kSelSeg];
    if ((int)blockIdx.x >= count) return;   // nothing for this workgroup (the usual case outside heavy load)
    for (int q = threadIdx.x; q < LDS::kSlots; q += blockDim.x) L.memo[q].state = 0u;
    if (threadIdx.x == 0) L.memo_gen = 0u;
    group_sync();
    int chunk_pos = 0, chunk_end = 0;
    bool first_chunk = true;
    int nodes_unit = 0;
    unsigned long long nodes_sum = 0ull;
    TW_SEL_DECL();
    while (true) {
        if (chunk_pos == chunk_end) {  // dynamic distribution (search effort varies by orders of magnitude), kWorkChunk windows per atomic;
                                       // the first chunk of a workgroup is its own (no same-address atomic storm at kernel start)
            if (first_chunk) { chunk_pos = (int)blockIdx.x * kWorkChunk; first_chunk = false; }
            else {
                if (threadIdx.x == 0) next_item = (int)gridDim.x * kWorkChunk + atomicAdd(sel_cursor(P, S, LIST), kWorkChunk);
                group_sync();
                chunk_pos = next_item;
                group_sync();
            }
            const int limit = chunk_pos < (int)gridDim.x * kWorkChunk ? (int)gridDim.x * kWorkChunk : count;   // static chunks are remapped below
            chunk_end = chunk_pos + kWorkChunk < limit ? chunk_pos + kWorkChunk : limit;
            if (chunk_pos >= count && chunk_pos >= (int)gridDim.x * kWorkChunk) {
                if (threadIdx.x == 0 && nodes_sum) atomicAdd((unsigned long long*)&P.unit_stats[(int64_t)nodes_unit * 8 + 5], nodes_sum);
                TW_SEL_FLUSH();
                break;
            }
        }
        int item = chunk_pos++;
        {   // the static chunks are strided over the workgroups: the long windows at the front of the list go to different
            // wavefronts instead of four in a row to the same one
            const int nstatic = (int)gridDim.x * kWorkChunk;
            if (item < nstatic) item = (item % kWorkChunk) * (int)gridDim.x + item / kWorkChunk;
            if (item >= count) continue;
        }
        const int pos = sel_position(P, S, G, item, LIST != 1);
        const bool rep = S.slot == 0;
        const int32_t *au = LIST == 3 ? (rep ? P.rtiny_unit : P.tiny_unit) : (rep ? P.rheavy_unit : P.heavy_unit), *aw = LIST == 3 ? (rep ? P.rtiny_win : P.tiny_win) : (rep ? P.rheavy_win : P.heavy_win);
        const int unit = __builtin_amdgcn_readfirstlane(au[pos]);  // wave-uniform: scalar loads below
        const uint32_t fm = (uint32_t)__builtin_amdgcn_readfirstlane(aw[pos]);
        const UnitDev& U = P.units[unit];
        const int first = (int)(fm >> 6), last = first + (int)(fm & 63u);
#ifdef TW_PROFILE_SEL
        const long long _w0 = wall_clock64();
#endif
        if (select_window_coop(P, U, unit, first, last - first + 1, L TW_SEL_PASS, false, &D, false)) {   // (uniform) a level outgrew the tables: listed for k_select_dp
            if (threadIdx.x == 0) {
                const int at = hard_base(P, S) + atomicAdd(sel_cursor(P, S, kHardCount), 1);
                P.hard_unit[at] = unit; P.hard_win[at] = (int32_t)fm;
                atomicAdd((unsigned long long*)&P.unit_stats[(int64_t)unit * 8 + 6], 1ull);
            }
        }
        if (unit != nodes_unit) {   // (uniform) search nodes per unit, added up over the windows this workgroup serves
            if (threadIdx.x == 0 && nodes_sum) atomicAdd((unsigned long long*)&P.unit_stats[(int64_t)nodes_unit * 8 + 5], nodes_sum);
            nodes_unit = unit; nodes_sum = 0ull;
        }
        nodes_sum += L.nodes_total;
#ifdef TW_PROFILE_SEL
        if (threadIdx.x == 0) {
            const unsigned long long dur = (unsigned long long)(wall_clock64() - _w0);
            atomicMax((unsigned long long*)&P.prof[8], (dur << 24) | ((unsigned long long)(last - first + 1) << 16) | (L.nodes_total >> 4 > 0xffffull ? 0xffffull : L.nodes_total >> 4));
            atomicAdd((unsigned long long*)&P.prof[9], dur);
            if ((dur << 24) >= (P.prof[8] & ~0xffffffull)) { P.prof[10] = (unsigned long long)L.pt[0]; P.prof[11] = (unsigned long long)L.pt[1]; P.prof[12] = (unsigned long long)L.pt[2]; P.prof[13] = (unsigned long long)L.pt[3]; P.prof[14] = (unsigned long long)U.E; }
        }
#endif
    }
}

// The listed windows of up to kBruteMax spans: every component is solved by complete enumeration (select_brute), so the
// workgroup needs neither the search stack nor the transposition table -- 1 KB of LDS, and the SIMDs fill up (the full kernel
// is limited to seven wavefronts per CU by its 22 KB, and what it does is latency-bound).  Same procedure, same results.
__global__ void __launch_bounds__(64) k_select_tiny(Dev P, TileSet S) {  // persistent workgroups, one listed window at a time
    if (*P.err != 0) return;
    __shared__ SelectLdsTiny L;
    __shared__ int next_item;
    __shared__ SelSegs G;
    sel_segments(P, S, 0, G);
    const int count = G.first[kSelSeg];
    if ((int)blockIdx.x * kWorkChunk >= count) return;
    int chunk_pos = (int)blockIdx.x * kWorkChunk, chunk_end = chunk_pos + kWorkChunk;   // the first chunk of a workgroup is its own
    TW_SEL_DECL();
    while (true) {
        if (chunk_pos == chunk_end) {
            if (threadIdx.x == 0) next_item = (int)gridDim.x * kWorkChunk + atomicAdd(sel_cursor(P, S, 0), kWorkChunk);
            group_sync();
            chunk_pos = next_item;
            group_sync();
            chunk_end = chunk_pos + kWorkChunk;
        }
        if (chunk_pos >= count) { TW_SEL_FLUSH(); break; }
        const int item = chunk_pos++;
        const int pos = sel_position(P, S, G, item, false);
        const int unit = __builtin_amdgcn_readfirstlane((S.slot == 0 ? P.rtiny_unit : P.tiny_unit)[pos]);
        const uint32_t fm = (uint32_t)__builtin_amdgcn_readfirstlane((S.slot == 0 ? P.rtiny_win : P.tiny_win)[pos]);
        const UnitDev& U = P.units[unit];
        const int first = (int)(fm >> 6), last = first + (int)(fm & 63u);
        select_window_coop(P, U, unit, first, last - first + 1, L TW_SEL_PASS, false);   // (complete enumeration only: no search nodes to report)
    }
}

// The windows k_select_heavy gave up on (a level of a component outgrew its small tables): one workgroup of kDpThreads lanes per
// window, the same solver on tables of kDpCap states; a component that outgrows those too is searched depth first.  Launched after
// the three instantiations of k_select_heavy have finished; with nothing listed the workgroups leave at once.
__global__ void __launch_bounds__(kDpThreads) k_select_dp(Dev P, TileSet S) {
    if (*P.err != 0) return;
    __shared__ SelectLds L;
    __shared__ SelectDp D;
    __shared__ int next_item;
    const int count = *sel_cursor(P, S, kHardCount), hb = hard_base(P, S);
    if ((int)blockIdx.x >= count) return;
    for (int q = threadIdx.x; q < SelectLds::kSlots; q += blockDim.x) L.memo[q].state = 0u;
    if (threadIdx.x == 0) L.memo_gen = 0u;
    group_sync();
    TW_SEL_DECL();
    int item = (int)blockIdx.x;
    while (item < count) {
        const int unit = __builtin_amdgcn_readfirstlane(P.hard_unit[hb + item]);
        const uint32_t fm = (uint32_t)__builtin_amdgcn_readfirstlane(P.hard_win[hb + item]);
        const UnitDev& U = P.units[unit];
        select_window_coop(P, U, unit, (int)(fm >> 6), (int)(fm & 63u) + 1, L TW_SEL_PASS, true, &D, true);
        if (threadIdx.x == 0) next_item = (int)gridDim.x + atomicAdd(sel_cursor(P, S, kHardNext), 1);
        group_sync();
        item = next_item;
        group_sync();
    }
    TW_SEL_FLUSH();
}

// ---------------------------------------------------------------------------------------------
// Span consumption (traceweaver_v1.py:457-463).  The reference removes the spans chosen by window w before it
// enumerates window w+1: the candidate list of a span is the top-5 over the tuples that avoid every span taken by an
// EARLIER window, and a window's selection is a function of its spans' lists.  Here every window is first solved on
// the full lists (which is top_k_2, traceweaver_v3.py:1185, needed anyway); then rounds of
//     k_claim        owner[x] = smallest window that currently chooses outgoing span x
//     k_detect_gone  per span: which of its candidate spans belong to an earlier window now?  If that set differs from
//                    the one its current list was computed with, the span is listed for k_enumerate_heavy (mode 1:
//                    those spans are not staged) and its window for k_select_heavy
// run until no set changes.  At that fixed point every window's selection is the one the sequential walk produces:
// by induction over the windows, window w's set of taken spans is exact once all earlier windows are final, and the
// first window never changes.  Windows that do not share candidates (perfect cuts) never interact, chains of
// size-capped windows settle front to back; all flagged windows of all units are re-solved concurrently by the same
// wavefront kernels as the first solve.
__device__ __forceinline__ void reset_class_lists(const Dev& P, int E, int t) {
    if (t < 2) P.heavy_in_count[t * (kMaxEp + 1) + E] = 0;
    if (t < 8) P.heavy_in_next[t * (kMaxEp + 1) + E] = 0;
    if (t == 0) { P.heavy_big_count[E] = 0; P.redo_count[E] = 0; P.fb_count[E] = 0; }
}
// reset_E > 0 (a class' own first round, launch_class_stage): the first workgroup also resets what a repair round resets of the
// class' work lists -- the class' enumeration is complete, k_detect_gone, which lists into them, comes after this kernel
__global__ void k_claim(Dev P, TileSet S, int reset_E) {
    if (*P.err != 0) return;  // an earlier kernel of this pass reported an error: its outputs are not usable
    if (reset_E > 0 && blockIdx.x == 0) reset_class_lists(P, reset_E, threadIdx.x);
    const TileDev Tl = P.tiles[set_tile(S)];
    const UnitDev& U = P.units[Tl.unit];
    const int i = Tl.first + threadIdx.x;
    if (i >= U.n_in) return;
    const int c = P.chosen[U.in_off + i];
    if (c < 0) return;
    const int w = P.wid[U.in_off + i];
    for (int e = 0; e < U.E; e++) atomicMin(&P.owner[U.ep_off[e] + cand_idx(P, U, i, c, e)], w);
}

__device__ __forceinline__ void finalize_span(const Dev& P, const UnitDev& U, int unit, int i);
__device__ __forceinline__ void gaps_span(const Dev& P, const UnitDev& U, int unit, int i);

// finish (a class' own first round, in anticipation of nothing to repair): 1 = the span's parents and counters (k_finalize) at once,
// 2 = and its gap samples (k_gaps, pass 1).  What they read of other threads' writes -- w_dirty -- only changes when the round
// finds something to repair, and then run_pass runs both again over all tiles.
__global__ void __launch_bounds__(kTile) k_detect_gone(Dev P, TileSet S, int round, int finish) {
    if (*P.err != 0) return;  // an earlier kernel of this pass reported an error: its outputs are not usable
    const int tile = set_tile(S);
    const TileDev Tl = P.tiles[tile];
    const UnitDev& U = P.units[Tl.unit];
    const int i = Tl.first + threadIdx.x;
    const bool live = i < U.n_in;
    const int64_t g = U.in_off + (live ? i : 0);
    const int w = P.wid[g];
    bool changed = false, any = false, narrow = true;
    if (live) {
        // (the words of P.gone are those of this pass once gone_valid says so: until a span's first taken candidate they read as zero --
        // no fill of the whole array per pass, 16 B per (span, endpoint))
        const bool valid = P.gone_valid[g] != 0;
        static_assert(kMaxEp * kCandWords <= 32, "zero_words holds one bit per (endpoint, word)");
        uint32_t zero_words = 0;   // words found zero while the span's words were not valid: written when the span turns valid below
        for (int e = 0; e < U.E; e++) {
            const int64_t b = ie_index(U, e, i);
            const int lo = P.c_lo[b];
            narrow &= P.c_hi[b] - lo + 1 <= kNarrow;
            for (int wd = 0; wd < kCandWords; wd++) {
                uint64_t bits = P.c_bits[b * kCandWords + wd], gone = 0;
                while (bits) {
                    const int r = __ffsll((long long)bits) - 1;
                    bits &= bits - 1;
                    if (P.owner[U.ep_off[e] + lo + 64 * wd + r] < w) gone |= 1ull << r;
                }
                const uint64_t old = valid ? P.gone[b * kCandWords + wd] : 0ull;
                if (gone != old) { P.gone[b * kCandWords + wd] = gone; changed = true; }
                else if (!valid) zero_words |= 1u << (e * kCandWords + wd);
                any |= gone != 0;
            }
        }
        if (!valid && changed) {
            for (int q = 0; q < U.E * kCandWords; q++)
                if ((zero_words >> q) & 1u) P.gone[ie_index(U, q / kCandWords, i) * kCandWords + q % kCandWords] = 0ull;
            P.gone_valid[g] = 1;
        }
        if (changed && !any) P.rep[g] = 0;   // nothing of this span is taken any more: its list on all spans is its list again
        if (changed) P.w_dirty[U.in_off + w] = 1;
    }
    heavy_append_rt(P, U.E, changed && any, narrow, Tl.unit, i);
    // the first lane of a window that sees a change lists the window for k_select_heavy (stamp = round: no reset between rounds)
    const bool listed = changed && atomicExch(&P.w_conf[U.in_off + w], round + 2) != round + 2;
    list_window(P, all_tiles(P), tile, U, Tl.unit, w, listed);   // (the lists of the repair rounds: over all tiles)
    const unsigned long long m = __ballot(changed);
    if (m != 0 && (threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(P.round_changed, __popcll(m));
    if (finish >= 1 && live) finalize_span(P, U, Tl.unit, i);
    if (finish >= 2 && live) gaps_span(P, U, Tl.unit, i);
}

// ---------------------------------------------------------------------------------------------
// (idempotent but for the three counters it adds to: k_reset_stats before it runs again)
__global__ void k_reset_stats(Dev P) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < P.n_units) { P.unit_stats[(int64_t)u * 8 + 0] = 0; P.unit_stats[(int64_t)u * 8 + 1] = 0; P.unit_stats[(int64_t)u * 8 + 3] = 0; }
}
__global__ void k_finalize(Dev P, TileSet S) {
    if (*P.err != 0) return;  // an earlier kernel of this pass reported an error: its outputs are not usable
    const TileDev Tl = P.tiles[set_tile(S)];
    const UnitDev& U = P.units[Tl.unit];
    const int i = Tl.first + threadIdx.x;
    if (i >= U.n_in) return;
    finalize_span(P, U, Tl.unit, i);
}
__device__ __forceinline__ void finalize_span(const Dev& P, const UnitDev& U, int unit, int i) {
    const int c = P.chosen[U.in_off + i];
    if (P.rep[U.in_off + i]) P.leaves[U.in_off + i] = P.leaves_r[U.in_off + i];   // tuples of the top_k call on the remaining spans
    if (P.win_end[U.in_off + i] && P.w_dirty[U.in_off + P.wid[U.in_off + i]]) atomicAdd((unsigned long long*)&P.unit_stats[(int64_t)unit * 8 + 3], 1ull);
    for (int e = 0; e < U.E; e++) {
        const int32_t x = c >= 0 ? cand_idx(P, U, i, c, e) : -1;
        P.parent[ie_index(U, e, i)] = x <= -TW_SKIP_BASE ? -2 : x;   // skip spans: ("Skip","Skip")
    }
    if (c != 0) atomicAdd((unsigned long long*)&P.unit_stats[(int64_t)unit * 8 + 0], 1ull);  // traceweaver_v3.py:1201-1207
    if (c < 0) atomicAdd((unsigned long long*)&P.unit_stats[(int64_t)unit * 8 + 1], 1ull);   // traceweaver_v3.py:1217
    if (i == 0) P.unit_stats[(int64_t)unit * 8 + 2] = P.unit_nwin[unit];
}

// Gap samples of the current assignment per scored slot (traceweaver_v3.py:717-762); NaN = dropped.
__global__ void k_gaps(Dev P, TileSet S) {
    if (*P.err != 0) return;  // an earlier kernel of this pass reported an error: its outputs are not usable
    const TileDev Tl = P.tiles[set_tile(S)];
    const UnitDev& U = P.units[Tl.unit];
    const int i = Tl.first + threadIdx.x;
    if (i >= U.n_in) return;
    gaps_span(P, U, Tl.unit, i);
}
__device__ __forceinline__ void gaps_span(const Dev& P, const UnitDev& U, int unit, int i) {
    const int E = U.E;
    double* out = P.gaps + P.gs_off[unit];
    const int64_t ist = P.in_start[U.in_off + i], ien = P.in_end[U.in_off + i];
    int32_t x[kMaxEp];
    for (int e = 0; e < E; e++) x[e] = P.parent[ie_index(U, e, i)];
    // only the rows of scored slots are written; the others were filled with NaN once at load time
    for (int e = 0; e < E; e++) {
        const bool have = x[e] >= 0;
        const int64_t st = have ? P.out_start[U.ep_off[e] + x[e]] : 0, en = have ? P.out_end[U.ep_off[e] + x[e]] : 0;
        if (U.npred[e] == 0) out[(int64_t)slot_root(E, e) * U.n_in + i] = have ? (double)(st - ist) * U.tscale : dnan();
        for (int j = 0; j < U.npred[e]; j++) {
            if (!U.pred_prim[e][j]) continue;
            const int p = U.pred_list[e][j];
            out[(int64_t)slot_prim(E, p, e) * U.n_in + i] = (have && x[p] >= 0) ? (double)(st - P.out_end[U.ep_off[p] + x[p]]) * U.tscale : dnan();
        }
        out[(int64_t)slot_close(E, e) * U.n_in + i] = have ? (double)(ien - en) * U.tscale : dnan();
    }
}

}  // namespace tw

#include "tw_tile.h"
#include "tw_lean.h"
