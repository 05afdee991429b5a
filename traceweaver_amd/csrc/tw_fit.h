// tw_fit.h -- device-side refit of the per-edge delay mixtures between the two passes: the reference's procedure,
// ComputeEpPairDistParams5 (traceweaver_v3.py:764-786), for every scored edge of every unit at once.
//
//     for n in 1..min(5, #unique):  GaussianMixture(n, covariance_type="diag").fit(x)      V3:772-779  (numpy's global RNG)
//     n_selected = the n with the smallest BIC among the fits that did not raise            V3:780
//     GaussianMixture(n_selected, random_state=100).fit(x)            (full covariance)     V3:784-785
//
// GaussianMixture.fit (scikit-learn, third-party; restated from its published algorithm, oracle/tw_refit.py is the CPU
// restatement pinned against scikit-learn itself) = KMeans(n, n_init=1) labels -- k-means++ seeding on the mean-centred
// samples *in the order of the requests*, then Lloyd iterations -- one-hot responsibilities, an M step, then EM until the
// mean log-likelihood moves by less than 1e-3 (<= 100 iterations), reg_covar 1e-6; a covariance <= 0 is scikit-learn's
// ValueError, which the reference skips (V3:777-779).
//
// Restated from scikit-learn 1.7.2 (the version the tests pin it against; the reference's requirements.txt names 1.5.1): the
// diagonal covariance is avg_X2 - means^2 + reg_covar (two terms, as that version computes it -- where a component sits on a
// single large sample value the sign of a covariance next to reg_covar = 1e-6 is rounding noise in either form, and such a
// fit "raises" or not depending on the summation order, in scikit-learn itself too), and the first k-means++ centre is drawn
// with `choice(n, p)` (one uniform: scikit-learn >= 1.3), which gives the draw schedule below.
//
// Randomness: the only random draws are the k-means++ seeds -- one uniform for the first centre (`choice(n, p)`), then
// 2 + int(ln n) per further centre -- 1, 3, 7, 10, 13 doubles for n = 1..5, independent of the data.  They come from a
// *tape* of uniforms: either handed over by the host (the doubles numpy's global MT19937 would have produced at that point
// of a seeded reference run: tw_fit_mixtures_tape) or drawn by the library from its own MT19937 (tw_fit_mixtures).  The
// refit of the selected count uses the first doubles of MT19937 seeded with 100, as `random_state=100` does.
//
// Gaps are integer microseconds (or integer multiples of 2^-k for load-scaled units), so a row of 10^5 samples holds only
// 10^2..10^4 distinct values: k_fit_compress turns every sorted row into (value, multiplicity) runs once; Lloyd and EM run
// over the runs with the multiplicity as weight (the same sums, grouped by value).  Only the k-means++ seeding, whose
// sampling walks a cumulative sum in request order, reads the unsorted row.  Sums of samples are exact in binary64 (integers
// below 2^53), so the mean that centres the samples is bit-identical to numpy's; sums of products are not, and differ from
// scikit-learn's BLAS sums in their last bits like scikit-learn's own results differ between machines.
//
// One workgroup runs one (edge, n) fit start to finish: every reduction has a fixed order, the result is a pure
// function of (samples in request order, tape).
#pragma once
#include "tw_device.h"

namespace tw {

#ifndef TW_FIT_THREADS
#define TW_FIT_THREADS 512
#endif
constexpr int kFitThreads = TW_FIT_THREADS;
constexpr int kFitWaves = kFitThreads / 64;
constexpr int kFitStats = 3 * kMaxComp + 2;
constexpr int kFitMaxIter = 100;          // GaussianMixture(max_iter=100)
constexpr double kFitTol = 1.0e-3;        // GaussianMixture(tol=1e-3)
constexpr double kFitRegCovar = 1.0e-6;   // GaussianMixture(reg_covar=1e-6)
constexpr int kKmMaxIter = 300;           // KMeans(max_iter=300)
constexpr double kKmTol = 1.0e-4;         // KMeans(tol=1e-4), times the variance of the samples
constexpr double kEps10 = 10.0 * 2.220446049250313e-16;
constexpr int kModelStride = 1 + 3 * kMaxComp;  // bic (inf = the fit raised), w[5], mu[5], precision_cholesky[5]
constexpr int kFitMaxTrials = 3;          // 2 + int(ln 5)
constexpr int kFitRowTape = 34;           // 1 + 3 + 7 + 10 + 13 doubles: the model-selection fits of one row

__host__ __device__ inline int fit_trials(int k) { return k < 3 ? 2 : 3; }                    // 2 + int(ln k), k = 1..5
__host__ __device__ inline int fit_draws(int k) { return 1 + (k - 1) * fit_trials(k); }       // 1, 3, 7, 10, 13
__host__ __device__ inline int fit_tape_pos(int k) { return k == 1 ? 0 : k == 2 ? 1 : k == 3 ? 4 : k == 4 ? 11 : 21; }

struct FitDev {
    const UnitDev* units;
    int32_t n_units;
    int64_t n_slots;
    const double* gaps;     // gap rows in request order, NaN = dropped; row q of unit u at gs_off[u] + q*n_in
    const double* sorted;   // the rows sorted ascending, NaN last (same offsets)
    const int64_t* gs_off;
    const int32_t* slot_unit;  // [n_slots] owning unit
    const uint8_t* slot_scored;  // [n_slots] 1 <=> the slot is a scored edge (its gap row is sorted and fitted)
    double* uval;           // run-length form of every sorted row: distinct values ascending (row offsets as in `sorted`) ...
    int32_t* ustart;        // ... and the index of each value's first occurrence in the sorted row
    int32_t* row_n;         // [n_slots] samples (non-NaN entries) of the row
    int32_t* row_uniq;      // [n_slots] distinct values of the row
    double* models;         // [n_slots][kMaxComp][kModelStride]
    int32_t* mix_n;         // [n_slots]
    double* mix_p;          // [n_slots][kMaxComp][3] weight, mean, precision_cholesky
    const double* tape;     // uniforms of the model-selection fits
    const int64_t* tape_off;  // [n_slots] first double of the row's fits (n = 1, 2, ... behind one another)
    const double* tape100;  // [kMaxComp][13] draws of the refit with n components: MT19937(100)
    int32_t* err;
    double* centres;        // [n_slots][kMaxComp][1 + kMaxComp] k-means start of every fit: mean of the samples, final centres (centred)
    int32_t k_only;         // model selection: > 0 = this launch fits only this component count, one workgroup per row (fit_run: a stream per count)
    int32_t by_row;         // model selection: 1 = the fits of a row in neighbouring workgroups (the row's walks meet in the last-level cache), 0 = all rows' fits of one count behind one another, the largest count first
};

// deterministic block reduction of `cnt` doubles per thread: lanes of a wavefront by shuffles (fixed
// butterfly order), wavefronts by thread 0 in wavefront order.  Result broadcast to every thread.
template <int cnt>
__device__ inline void block_reduce(double (&vals)[cnt], double* sh, int nt) {   // nt: threads of the workgroup that take part
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6, nwave = (nt + 63) >> 6;
#pragma unroll
    for (int c = 0; c < cnt; c++) {
        double v = vals[c];
        for (int off = 32; off >= 1; off >>= 1)
            if (off < nt) v += __shfl_down(v, off);
        if (lane == 0) sh[c * kFitWaves + wave] = v;
    }
    __syncthreads();
    if (t < cnt) {  // thread c sums the wavefront partials of statistic c in wavefront order
        double v = sh[t * kFitWaves];
        for (int w = 1; w < nwave; w++) v += sh[t * kFitWaves + w];
        sh[t * kFitWaves] = v;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < cnt; c++) vals[c] = sh[c * kFitWaves];
    __syncthreads();
}

// One workgroup per scored row: distinct values and their first positions (a head is a sample that differs
// from its left neighbour; NaN = dropped samples, sorted last, are not counted).
constexpr int kCompressThreads = 1024;   // a row of 10^5 samples in a dozen rounds (256 threads, 4 chunks a round: a hundred rounds of two barriers each, 0.54 ms)
__global__ void __launch_bounds__(kCompressThreads) k_fit_compress(FitDev F) {
    constexpr int R = 8;  // chunks of blockDim samples per round: their loads are in flight together
    __shared__ int32_t wave_heads[R][kCompressThreads / 64];
    __shared__ int32_t carry_sh, n_sh;
    const int64_t q = blockIdx.x;
    const int t = threadIdx.x, nt = blockDim.x;
    const int lane = t & 63, wave = t >> 6, nwave = (nt + 63) >> 6;
    if (!F.slot_scored[q]) {
        if (t == 0) { F.row_n[q] = 0; F.row_uniq[q] = 0; }
        return;
    }
    const UnitDev& U = F.units[F.slot_unit[q]];
    const int n_all = U.n_in;
    const int64_t row = F.gs_off[F.slot_unit[q]] + (q - U.slot_off) * (int64_t)n_all;
    const double* x = F.sorted + row;
    if (t == 0) { carry_sh = 0; n_sh = 0; }
    __syncthreads();
    for (int base = 0; base < n_all; base += R * nt) {
        double xi[R];
        bool head[R];
        int before[R];
#pragma unroll
        for (int j = 0; j < R; j++) {
            const int i = base + j * nt + t;
            xi[j] = 0.0;
            head[j] = false;
            if (i < n_all) {
                xi[j] = x[i];
                const bool valid = xi[j] == xi[j];
                head[j] = valid && (i == 0 || x[i - 1] != xi[j]);
                if (valid && (i == n_all - 1 || !(x[i + 1] == x[i + 1]))) n_sh = i + 1;  // last sample of the row
            }
        }
        // position of a head = heads before it: within the wavefront from the ballot, across wavefronts and chunks through LDS
#pragma unroll
        for (int j = 0; j < R; j++) {
            const unsigned long long heads = __ballot(head[j]);
            before[j] = __popcll(heads & ((1ull << lane) - 1ull));
            if (lane == 0) wave_heads[j][wave] = __popcll(heads);
        }
        __syncthreads();
        int pos = carry_sh;
#pragma unroll
        for (int j = 0; j < R; j++) {
            int chunk = 0, mine = 0;
            for (int w = 0; w < nwave; w++) { const int c = wave_heads[j][w]; if (w < wave) mine += c; chunk += c; }
            if (head[j]) {
                F.uval[row + pos + mine + before[j]] = xi[j];
                F.ustart[row + pos + mine + before[j]] = base + j * nt + t;
            }
            pos += chunk;
        }
        __syncthreads();
        if (t == 0) carry_sh = pos;  // (the next round's barrier orders this update before its use)
    }
    __syncthreads();
    if (t == 0) { F.row_n[q] = n_sh; F.row_uniq[q] = carry_sh; }
}

// The same run-length form WITHOUT sorting the row (the default route of fit_prepare; the sort + k_fit_compress above is its
// fallback).  A row of 10^5 gap samples holds 10^2..10^4 distinct values and every value is a non-negative integer multiple of
// the unit's time scale: one workgroup per scored row counts the samples per value in an open-addressing hash table in LDS
// (one 64-bit word per value: integer value << 32 | multiplicity, claimed by a compare-and-swap, counted by an add -- both
// order-independent), sorts the table's words (bitonic network, empty words last) and writes the runs: distinct values
// ascending, exclusive prefix sums of the multiplicities = index of each value's first occurrence in the sorted row.  The row
// is read once (the sort route: one pack, four to five radix passes, one unpack and k_fit_compress's read -- 10 GB per step on
// the bench workload).  A row the table cannot take (a sample that is negative, not an integer multiple of the scale or beyond
// 32 bits; more than kHashMax distinct values) raises `*fallback`: the host then runs the sort route for the batch.
#ifndef TW_FIT_HASH_SLOTS
#define TW_FIT_HASH_SLOTS 16384
#endif
constexpr int kHashSlots = TW_FIT_HASH_SLOTS;           // words of the table (128 KB of LDS: one workgroup per CU)
constexpr int kHashMax = kHashSlots / 4 * 3;            // distinct values of a row on this route
constexpr unsigned long long kHashEmpty = ~0ull;
constexpr int kRunsThreads = 1024;
static_assert((kHashSlots & (kHashSlots - 1)) == 0 && kHashSlots >= 64, "the table size is a power of two");

__global__ void __launch_bounds__(kRunsThreads) k_fit_runs(FitDev F, int32_t* fallback) {
    constexpr int R = 8;  // samples per thread and round: their loads are in flight together
    __shared__ unsigned long long tab[kHashSlots];
    __shared__ int32_t s_part[kRunsThreads / 64];
    __shared__ int32_t s_uniq, s_n, s_bad;
    const int64_t q = blockIdx.x;
    const int t = threadIdx.x, nt = blockDim.x;
    const int lane = t & 63, wave = t >> 6;
    if (!F.slot_scored[q]) {
        if (t == 0) { F.row_n[q] = 0; F.row_uniq[q] = 0; }
        return;
    }
    const UnitDev& U = F.units[F.slot_unit[q]];
    const int n_all = U.n_in;
    const int64_t row = F.gs_off[F.slot_unit[q]] + (q - U.slot_off) * (int64_t)n_all;
    const double* x = F.gaps + row;
    const double scale = U.tscale, inv = 1.0 / U.tscale;   // a power of two: both exact
    // a short row takes a short table (a power of two of at least twice its samples): what is cleared and sorted is the table, not
    // the row -- 16 384 words cost a row of 600 samples 0.14 ms
    int slots = kHashSlots < 256 ? kHashSlots : 256;
    while (slots < kHashSlots && slots < 2 * n_all) slots <<= 1;
    const int slot_shift = 32 - __builtin_ctz((unsigned)slots), hash_max = slots / 4 * 3;
    for (int r = t; r < slots; r += nt) tab[r] = kHashEmpty;
    if (t == 0) { s_uniq = 0; s_n = 0; s_bad = 0; }
    __syncthreads();
    int mine = 0;
    for (int base = 0; base < n_all; base += R * nt) {
        double xi[R];
#pragma unroll
        for (int j = 0; j < R; j++) { const int i = base + j * nt + t; xi[j] = i < n_all ? x[i] : dnan(); }
#pragma unroll
        for (int j = 0; j < R; j++) {
            if (!(xi[j] == xi[j])) continue;   // NaN = dropped sample
            mine++;
            const double v = xi[j] * inv;
            const uint32_t key = (uint32_t)v;
            if (__double_as_longlong(xi[j]) < 0 || !(v < 4294967295.0) || (double)key != v) { s_bad = 1; continue; }
            uint32_t slot = (key * 2654435761u) >> slot_shift;   // multiplicative hashing: the product's top bits
            const unsigned long long word = (unsigned long long)key << 32;
            bool placed = false;
            for (int probe = 0; probe < slots && !placed; probe++) {
                const unsigned long long old = atomicCAS(&tab[slot], kHashEmpty, word);
                if (old == kHashEmpty) { if (atomicAdd(&s_uniq, 1) >= hash_max) s_bad = 1; placed = true; }
                else if ((uint32_t)(old >> 32) == key) placed = true;
                else slot = (slot + 1) & (uint32_t)(slots - 1);
            }
            if (placed) atomicAdd(&tab[slot], 1ull); else s_bad = 1;
        }
        if (s_bad) break;   // (every thread leaves within a round of the first refusal; the table never fills: kHashMax + threads < slots, or the probe bound ends it)
    }
    atomicAdd(&s_n, mine);
    __syncthreads();
    if (s_bad) {
        if (t == 0) atomicExch(fallback, 1);
        return;
    }
    // bitonic sort of the table's words, ascending (a value's word orders by the value; empty words are the largest)
    for (int k = 2; k <= slots; k <<= 1) {
        for (int j = k >> 1; j >= 1; j >>= 1) {
            for (int p = t; p < slots / 2; p += nt) {
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));   // the p-th index whose bit j is clear
                const unsigned long long a = tab[i], b = tab[i | j];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { tab[i] = b; tab[i | j] = a; }
            }
            __syncthreads();
        }
    }
    // runs: every thread a stretch of consecutive words; exclusive prefix sums of the multiplicities (wavefront scan by shuffles,
    // the wavefronts' totals through LDS)
    const int uniq = s_uniq, per = (uniq + nt - 1) / nt;
    const int r_lo = t * per < uniq ? t * per : uniq, r_hi = r_lo + per < uniq ? r_lo + per : uniq;
    int sum = 0;
    for (int r = r_lo; r < r_hi; r++) sum += (int)(uint32_t)tab[r];
    int inc = sum;
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(inc, off); if (lane >= off) inc += o; }
    if (lane == 63 || t == nt - 1) s_part[wave] = inc;
    __syncthreads();
    int before = inc - sum;
    for (int w = 0; w < wave; w++) before += s_part[w];
    for (int r = r_lo; r < r_hi; r++) {
        const unsigned long long w = tab[r];
        F.uval[row + r] = (double)(uint32_t)(w >> 32) * scale;
        F.ustart[row + r] = before;
        before += (int)(uint32_t)w;
    }
    if (t == 0) { F.row_n[q] = s_n; F.row_uniq[q] = uniq; }
}

// sklearn.metrics.pairwise._euclidean_distances(c, X, squared=True) for one feature: ((-2 (c x)) + c c) + x x, clipped at 0
__device__ __forceinline__ double fit_dist(double c, double x) {
    double d = -2.0 * (c * x);
    d += c * c;
    d += x * x;
    return d > 0.0 ? d : 0.0;
}
// label of a centred sample under Lloyd's rule: first minimum of c^2 - 2 x c (the strict `<` scan of _update_chunk_dense)
__device__ __forceinline__ int fit_label(const double* cen, int k, double x) {
    double best = cen[0] * cen[0] + (-2.0) * (x * cen[0]);
    int lab = 0;
#pragma unroll
    for (int j = 1; j < kMaxComp; j++) {
        if (j < k) {
            const double d = cen[j] * cen[j] + (-2.0) * (x * cen[j]);
            if (d < best) { best = d; lab = j; }
        }
    }
    return lab;
}

// k_fit_seed holds no mixture arithmetic: built to 128 registers, two workgroups per CU hide the latency of its walks
#ifndef TW_FIT_SEED_ATTR
#define TW_FIT_SEED_ATTR
#endif
constexpr int kRunCache = 4096;   // runs of a row kept in LDS (value + multiplicity = 48 KB) by the iterating kernels; longer rows read the rest from global memory
constexpr int kSeedBlocks = 16;   // a wavefront's stretch of the row is summed in this many blocks (k_fit_seed)

struct FitRow {
    int64_t q, row;
    int k, n_all, n, uniq;
    bool live;
};
// Which (row, component count) a workgroup serves.  Model selection: the fits with the most components first (they run
// longest), `skip1` leaves k = 1 out (k-means of one cluster is no work); refit: k = the selected count.
template <bool kFull>
__device__ __forceinline__ FitRow fit_row(const FitDev& F, bool skip1) {
    FitRow R;
    const int nfit = kFull ? 1 : (int)(gridDim.x / F.n_slots);   // counts fitted per row by this launch
    R.q = (kFull || F.k_only > 0) ? (int64_t)blockIdx.x : (F.by_row ? (int64_t)(blockIdx.x / nfit) : (int64_t)(blockIdx.x % F.n_slots));
    R.k = kFull ? F.mix_n[R.q] : (F.k_only > 0 ? F.k_only : kMaxComp - (F.by_row ? (int)(blockIdx.x % nfit) : (int)(blockIdx.x / F.n_slots)));
    const UnitDev& U = F.units[F.slot_unit[R.q]];
    R.n_all = U.n_in;
    R.row = F.gs_off[F.slot_unit[R.q]] + (R.q - U.slot_off) * (int64_t)R.n_all;
    R.n = F.row_n[R.q];
    R.uniq = F.row_uniq[R.q];
    R.live = F.slot_scored[R.q] && R.n > 0 && R.k >= (skip1 ? 2 : 1) && R.k <= R.uniq && R.k <= kMaxComp;
    return R;
}

// One workgroup per (row, component count >= 2): the k-means start of GaussianMixture(k).fit(row) --
// KMeans(k, n_init=1): k-means++ seeding on the mean-centred samples in request order, then Lloyd over the runs.  Leaves the
// mean and the final centres in F.centres.  Few registers (no mixture arithmetic): several workgroups per CU hide the
// latency of the walks over the row.
template <bool kFull>
__global__ void __launch_bounds__(kFitThreads) TW_FIT_SEED_ATTR k_fit_seed(FitDev F) {
    __shared__ double sh[(2 * kMaxComp + 1) * kFitWaves];
    __shared__ double cen[kMaxComp];
    __shared__ double blk[kFitWaves][kSeedBlocks];    // closest squared distances summed per block of a wavefront's stretch
    __shared__ double cand[kFitMaxTrials][kFitWaves]; // candidate found by each wavefront (NaN = none)
    __shared__ double lastv[kFitWaves];
    __shared__ int32_t lasti[kFitWaves], cnt_w[kFitWaves];
    __shared__ double par[2 * kMaxComp];
    __shared__ double run_x[kRunCache];
    __shared__ int32_t run_c[kRunCache];
    const FitRow R = fit_row<kFull>(F, true);
    if (!R.live) return;
    const int k = R.k, n = R.n, uniq = R.uniq, n_all = R.n_all;
    const int64_t q = R.q;
    const int t = threadIdx.x, nt = blockDim.x;
    const int W = nt < 64 ? nt : 64, lane = t & 63, wave = t >> 6, nwave = (nt + 63) >> 6;
    const double* xr = F.gaps + R.row;      // request order
    const double* xv = F.uval + R.row;      // distinct values, ascending
    const int32_t* xa = F.ustart + R.row;   // first index of each value in the sorted row
    const double* tape = kFull ? F.tape100 + (k - 1) * 13 : F.tape + F.tape_off[q] + fit_tape_pos(k);
    const double dn = (double)n;
    for (int r = t; r < uniq && r < kRunCache; r += nt) { run_x[r] = xv[r]; run_c[r] = (r + 1 < uniq ? xa[r + 1] : n) - xa[r]; }
    __syncthreads();
#define RUN_X(r) ((r) < kRunCache ? run_x[r] : xv[r])
#define RUN_C(r) ((double)((r) < kRunCache ? run_c[r] : ((r) + 1 < uniq ? xa[(r) + 1] : n) - xa[r]))

    // ---- mean and variance of the samples (exact sum: integers below 2^53)
    double mv[1] = {0.0};
    for (int r = t; r < uniq; r += nt) mv[0] += RUN_X(r) * RUN_C(r);
    block_reduce<1>(mv, sh, nt);
    const double mean = mv[0] / dn;
    mv[0] = 0.0;
    for (int r = t; r < uniq; r += nt) { const double d = RUN_X(r) - mean; mv[0] += (d * d) * RUN_C(r); }
    block_reduce<1>(mv, sh, nt);
    const double km_tol = mv[0] / dn * kKmTol;

    // ---- k-means++ seeding (sklearn.cluster._kmeans._kmeans_plusplus) on the centred samples in request order.
    // Every wavefront owns one stretch of the row, cut into kSeedBlocks blocks of whole strips of 64 samples.
    const int strips = (n_all + W - 1) / W, strips_w = (strips + nwave - 1) / nwave;
    const int strips_b = (strips_w + kSeedBlocks - 1) / kSeedBlocks;
    const int s_lo = wave * strips_w * W, s_hi = (s_lo + strips_w * W) < n_all ? (s_lo + strips_w * W) : n_all;
    // the first centre: sample number floor(u n) of the valid samples
    int64_t cid = (int64_t)(tape[0] * dn);
    if (cid > n - 1) cid = n - 1;
    double x_last = 0.0;
    if (n == n_all) {   // no dropped samples (the usual case): positions are indices
        if (t == 0) cen[0] = xr[cid] - mean;
        x_last = xr[n_all - 1] - mean;
        __syncthreads();
    } else {
        {   // valid samples per stretch, the last valid sample
            int c = 0, li = -1;
            double lv = 0.0;
            for (int b = s_lo; b < s_hi; b += W) {
                const int i = b + lane;
                const double x = i < s_hi ? xr[i] : dnan();
                const bool ok = x == x;
                c += ok ? 1 : 0;
                if (ok) { li = i; lv = x; }
            }
            for (int off = 32; off >= 1; off >>= 1) {
                if (off < W) {
                    c += __shfl_down(c, off);
                    const int oi = __shfl_down(li, off);
                    const double ov = __shfl_down(lv, off);
                    if (oi > li) { li = oi; lv = ov; }
                }
            }
            if (lane == 0) { cnt_w[wave] = c; lasti[wave] = li; lastv[wave] = lv; }
        }
        __syncthreads();
        { int li = -1; for (int w = 0; w < nwave; w++) if (lasti[w] > li) { li = lasti[w]; x_last = lastv[w]; } }
        x_last -= mean;
        int64_t before = 0;
        int w0 = 0;
        for (; w0 < nwave - 1 && before + cnt_w[w0] <= cid; w0++) before += cnt_w[w0];
        if (wave == w0) {  // (wave-uniform branch) walk the stretch to the cid-th valid sample
            int64_t seen = before;
            for (int b = s_lo; b < s_hi; b += W) {
                const int i = b + lane;
                const double x = i < s_hi ? xr[i] : dnan();
                const unsigned long long okm = __ballot(x == x);
                const int c = __popcll(okm);
                if (seen + c > cid) {
                    const int rank = (int)(cid - seen);  // the rank-th set bit of okm
                    const bool mine = (x == x) && __popcll(okm & ((1ull << lane) - 1ull)) == rank;
                    if (mine) cen[0] = x - mean;
                    break;
                }
                seen += c;
            }
        }
        __syncthreads();
    }
    const int trials = fit_trials(k);
    int tp = 1;
    for (int c = 1; c < k; c++) {
        double cc[kMaxComp - 1];
#pragma unroll
        for (int j = 0; j < kMaxComp - 1; j++) cc[j] = j < c ? cen[j] : 0.0;
        // pass A1: the closest squared distances summed per block of the row (request order); eight strips' loads in flight
        for (int bi = 0; bi < kSeedBlocks; bi++) {
            const int b_lo = s_lo + bi * strips_b * W, b_hi = (b_lo + strips_b * W) < s_hi ? (b_lo + strips_b * W) : s_hi;
            double a = 0.0;
            for (int b = b_lo; b < b_hi; b += 8 * W) {
                double x8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { const int i = b + u * W + lane; x8[u] = i < b_hi ? xr[i] : dnan(); }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (x8[u] == x8[u]) {
                        const double xc = x8[u] - mean;
                        double v = fit_dist(cc[0], xc);
#pragma unroll
                        for (int j = 1; j < kMaxComp - 1; j++) if (j < c) { const double d = fit_dist(cc[j], xc); if (d < v) v = d; }
                        a += v;
                    }
                }
            }
            for (int off = 32; off >= 1; off >>= 1) if (off < W) a += __shfl_down(a, off);
            if (lane == 0) blk[wave][bi] = a;
        }
        __syncthreads();
        double pot = 0.0, base = 0.0, mine_sum = 0.0;
        for (int w = 0; w < nwave; w++) {
            double sw = 0.0;
            for (int bi = 0; bi < kSeedBlocks; bi++) sw += blk[w][bi];
            if (w == wave) { base = pot; mine_sum = sw; }
            pot += sw;
        }
        const double next = base + mine_sum;
        // pass A2: np.searchsorted(cumsum(closest), target) -- the first sample whose running sum reaches the target lies in the
        // stretch, and there in the block, whose range of running sums holds the target: that block is walked with the running sum
#pragma unroll
        for (int j = 0; j < kFitMaxTrials; j++) {
            double found = dnan();
            if (j < trials) {
                const double target = tape[tp + j] * pot;
                if ((wave == 0 && target <= next) || (target > base && target <= next)) {   // (wave-uniform)
                    double run = base;
                    int bi = 0;
                    for (; bi < kSeedBlocks - 1 && run + blk[wave][bi] < target; bi++) run += blk[wave][bi];
                    const int b_lo = s_lo + bi * strips_b * W, b_hi = (b_lo + strips_b * W) < s_hi ? (b_lo + strips_b * W) : s_hi;
                    double lastx = dnan();
                    for (int b = b_lo; b < b_hi; b += W) {
                        const int i = b + lane;
                        const double x = i < b_hi ? xr[i] : dnan();
                        const bool ok = x == x;
                        const double xc = x - mean;
                        double v = 0.0;
                        if (ok) {
                            v = fit_dist(cc[0], xc);
#pragma unroll
                            for (int jj = 1; jj < kMaxComp - 1; jj++) if (jj < c) { const double d = fit_dist(cc[jj], xc); if (d < v) v = d; }
                        }
                        double sc = v;  // inclusive scan over the lanes
                        for (int off = 1; off < W; off <<= 1) { const double o = __shfl_up(sc, off); if (lane >= off) sc += o; }
                        const unsigned long long m = __ballot(ok && run + sc >= target);
                        if (m) { found = __shfl(xc, __ffsll((long long)m) - 1); break; }
                        const unsigned long long okm = __ballot(ok);
                        if (okm) lastx = __shfl(xc, 63 - __clzll((long long)okm));
                        run += __shfl(sc, W - 1);
                    }
                    if (!(found == found)) found = lastx;   // (rounding: the walk's sum fell short of the target)
                    if (!(found == found)) found = x_last;
                }
            }
            if (lane == 0) cand[j][wave] = found;
        }
        tp += trials;
        __syncthreads();
        double cv[kFitMaxTrials];
#pragma unroll
        for (int j = 0; j < kFitMaxTrials; j++) {
            cv[j] = x_last;  // np.clip(candidate_ids, None, n - 1): a target beyond the last running sum
            for (int w = nwave - 1; w >= 0; w--) { const double v = cand[j][w]; if (v == v) cv[j] = v; }
        }
        // pass B: the potential of every candidate = sum of min(closest, distance to the candidate) -- over the runs
        // (the same terms grouped by value)
        double a[kFitMaxTrials] = {0.0, 0.0, 0.0};
        for (int r = t; r < uniq; r += nt) {
            const double xc = RUN_X(r) - mean;
            const double cw = RUN_C(r);
            double v = fit_dist(cc[0], xc);
#pragma unroll
            for (int j = 1; j < kMaxComp - 1; j++) if (j < c) { const double d = fit_dist(cc[j], xc); if (d < v) v = d; }
#pragma unroll
            for (int j = 0; j < kFitMaxTrials; j++) if (j < trials) { const double d = fit_dist(cv[j], xc); a[j] += cw * (d < v ? d : v); }
        }
        block_reduce<kFitMaxTrials>(a, sh, nt);
        double bp = a[0], bc = cv[0];   // np.argmin: the first minimum
        if (trials > 1 && a[1] < bp) { bp = a[1]; bc = cv[1]; }
        if (trials > 2 && a[2] < bp) { bp = a[2]; bc = cv[2]; }
        if (t == 0) cen[c] = bc;
        __syncthreads();
    }

    // ---- Lloyd iterations (_kmeans_single_lloyd) over the runs.  Every thread holds the centres in registers: the sums
    // come back from the reduction to all threads, so all compute the same update.
    double cc[kMaxComp], cp[kMaxComp];
#pragma unroll
    for (int j = 0; j < kMaxComp; j++) { cc[j] = j < k ? cen[j] : 0.0; cp[j] = 0.0; }
    bool strict = false;
    for (int it = 0; it < kKmMaxIter; it++) {
        double v[2 * kMaxComp + 1];
#pragma unroll
        for (int j = 0; j < 2 * kMaxComp + 1; j++) v[j] = 0.0;
        for (int r = t; r < uniq; r += nt) {
            const double xc = RUN_X(r) - mean;
            const double cw = RUN_C(r);
            const int lab = fit_label(cc, k, xc);
            const int old = it == 0 ? -1 : fit_label(cp, k, xc);
            if (old != lab) v[2 * kMaxComp] += 1.0;   // labels != labels_old
#pragma unroll
            for (int j = 0; j < kMaxComp; j++) if (j == lab) { v[j] += cw; v[kMaxComp + j] += xc * cw; }
        }
        block_reduce<2 * kMaxComp + 1>(v, sh, nt);
        bool any_empty = false;
#pragma unroll
        for (int j = 0; j < kMaxComp; j++) if (j < k && v[j] == 0.0) any_empty = true;
        if (any_empty) {
            // _relocate_empty_clusters_dense: the samples farthest from their centres found the empty clusters (one sample each,
            // labels untouched).  Rare (never on the reference's corpora); thread 0 walks the runs.
            if (t == 0) {
                int taken_run[kMaxComp], taken_cnt[kMaxComp], ntaken = 0;
                for (int e = 0; e < k; e++) {
                    if (v[e] != 0.0) continue;
                    int fr = -1;
                    double fd = -1.0;
                    for (int r = 0; r < uniq; r++) {
                        int used = 0;
                        for (int z = 0; z < ntaken; z++) if (taken_run[z] == r) used = taken_cnt[z];
                        if ((r + 1 < uniq ? xa[r + 1] : n) - xa[r] <= used) continue;
                        const double xc = xv[r] - mean;
                        const int lb = fit_label(cc, k, xc);
                        double cl = cc[0];
#pragma unroll
                        for (int j = 1; j < kMaxComp; j++) if (j == lb) cl = cc[j];
                        const double d0 = xc - cl;
                        if (d0 * d0 > fd) { fd = d0 * d0; fr = r; }
                    }
                    if (fr < 0) break;
                    bool again = false;
                    for (int z = 0; z < ntaken; z++) if (taken_run[z] == fr) { taken_cnt[z]++; again = true; }
                    if (!again) { taken_run[ntaken] = fr; taken_cnt[ntaken] = 1; ntaken++; }
                    const double xc = xv[fr] - mean;
                    const int old = fit_label(cc, k, xc);
#pragma unroll
                    for (int j = 0; j < kMaxComp; j++) {
                        if (j == old) { v[kMaxComp + j] -= xc; v[j] -= 1.0; }
                        if (j == e) { v[kMaxComp + j] = xc; v[j] = 1.0; }
                    }
                }
#pragma unroll
                for (int j = 0; j < kMaxComp; j++) { par[j] = v[j]; par[kMaxComp + j] = v[kMaxComp + j]; }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kMaxComp; j++) { v[j] = par[j]; v[kMaxComp + j] = par[kMaxComp + j]; }
            __syncthreads();
        }
        double shift = 0.0;
#pragma unroll
        for (int j = 0; j < kMaxComp; j++) {
            if (j < k) {
                const double nc = v[kMaxComp + j] * (1.0 / v[j]);   // _average_centers: the sum times the reciprocal of the weight
                const double d = nc - cc[j];
                const double sd = sqrt(d * d);                      // _center_shift, then (center_shift ** 2).sum()
                shift += sd * sd;
                cp[j] = cc[j];
                cc[j] = nc;
            }
        }
        if (v[2 * kMaxComp] == 0.0) { strict = true; break; }
        if (shift <= km_tol) break;
    }
    // the labels KMeans returns: those of the last assignment (strict convergence) or a fresh one under the final centres
    if (t == 0) {
        double* out = F.centres + (q * kMaxComp + (k - 1)) * (kMaxComp + 1);
        out[0] = mean;
#pragma unroll
        for (int j = 0; j < kMaxComp; j++) out[1 + j] = strict ? cp[j] : cc[j];
    }
}

// One workgroup per (row, component count): GaussianMixture(k).fit(row) from the k-means labels on.  kFull = false: the
// model-selection fits (diagonal covariance; result = BIC + parameters in F.models); kFull = true: the refit of the
// selected count (full covariance; result = the row of the mixture table).
template <bool kFull>
__global__ void __launch_bounds__(kFitThreads) k_fit_em(FitDev F) {
    __shared__ double sh[kFitStats * kFitWaves];
    __shared__ double run_x[kRunCache];
    __shared__ int32_t run_c[kRunCache];
    const FitRow R = fit_row<kFull>(F, false);
    const int k = R.k, n = R.n, uniq = R.uniq;
    const int64_t q = R.q;
    const int t = threadIdx.x;
    double* model = kFull ? nullptr : F.models + (q * kMaxComp + (k - 1)) * kModelStride;
    if (!R.live) {
        if (!kFull && t == 0 && k >= 1 && k <= kMaxComp) model[0] = dinf();
        return;
    }
    const int nt = blockDim.x;
    const double* xv = F.uval + R.row;      // distinct values, ascending
    const int32_t* xa = F.ustart + R.row;   // first index of each value in the sorted row
    const double dn = (double)n;
    const double* seed = F.centres + (q * kMaxComp + (k - 1)) * (kMaxComp + 1);
    const double mean = k > 1 ? seed[0] : 0.0;
    for (int r = t; r < uniq && r < kRunCache; r += nt) { run_x[r] = xv[r]; run_c[r] = (r + 1 < uniq ? xa[r + 1] : n) - xa[r]; }
    __syncthreads();

    // ---- GaussianMixture._initialize: one-hot responsibilities of the k-means labels, then the M step.  From here on the
    // parameters live in registers: every reduction hands its sums to all threads, all compute the same update.
    bool failed = false;
    double pw[kMaxComp], pm[kMaxComp], pv[kMaxComp];   // weights, means, covariances
    {
        double cc[kMaxComp];
#pragma unroll
        for (int j = 0; j < kMaxComp; j++) cc[j] = (k > 1 && j < k) ? seed[1 + j] : 0.0;
        double v[3 * kMaxComp];
#pragma unroll
        for (int j = 0; j < 3 * kMaxComp; j++) v[j] = 0.0;
        for (int r = t; r < uniq; r += nt) {
            const double x = RUN_X(r);
            const double cw = RUN_C(r);
            const int lab = k > 1 ? fit_label(cc, k, x - mean) : 0;
#pragma unroll
            for (int j = 0; j < kMaxComp; j++) if (j == lab) { v[j] += cw; v[kMaxComp + j] += cw * x; v[2 * kMaxComp + j] += cw * (x * x); }
        }
        block_reduce<3 * kMaxComp>(v, sh, nt);
#pragma unroll
        for (int j = 0; j < kMaxComp; j++) {
            const double nk = v[j] + kEps10;
            pw[j] = nk / dn;
            pm[j] = j < k ? v[kMaxComp + j] / nk : 0.0;
            pv[j] = (v[2 * kMaxComp + j] / nk - pm[j] * pm[j]) + kFitRegCovar;   // _estimate_gaussian_covariances_diag
        }
        if (kFull) {   // _estimate_gaussian_covariances_full: squared deviations from the means
            double c2[kMaxComp];
#pragma unroll
            for (int j = 0; j < kMaxComp; j++) c2[j] = 0.0;
            for (int r = t; r < uniq; r += nt) {
                const double x = RUN_X(r);
                const double cw = RUN_C(r);
                const int lab = k > 1 ? fit_label(cc, k, x - mean) : 0;
#pragma unroll
                for (int j = 0; j < kMaxComp; j++) if (j == lab) { const double d = x - pm[j]; c2[j] += cw * (d * d); }
            }
            block_reduce<kMaxComp>(c2, sh, nt);
#pragma unroll
            for (int j = 0; j < kMaxComp; j++) pv[j] = c2[j] / (v[j] + kEps10) + kFitRegCovar;
        }
#pragma unroll
        for (int j = 0; j < kMaxComp; j++) if (j < k && !(pv[j] > 0.0)) failed = true;
    }

    // ---- EM (BaseMixture.fit_predict); the last round only scores the final parameters (bic() re-scores).
    // Responsibilities as exp(wl - max) / sum (scikit-learn: exp(wl - logsumexp), the same number up to rounding): one
    // exponential per component and sample.  The full-covariance M step takes its moments around the previous means
    // (exact algebra of sum r (x - new mean)^2; the shift of a mean per iteration is small against the spread).
    double lower = -dinf();
    bool last = false;
    for (int iter = 0; iter <= kFitMaxIter && !failed; iter++) {
        if (iter == kFitMaxIter) last = true;
        double a0[kMaxComp], a1[kMaxComp], a2[kMaxComp], cst[kMaxComp], lwt[kMaxComp], v[kFitStats];
        // The sweep's constants of component j (a reciprocal square root and two logarithms, ~300 dependent f64 instructions a
        // component) used to be computed by every thread for every component: a fifth of a sweep over a long row and most of a sweep
        // over a short one.  Lane j of every wavefront computes component j's and hands them to the other lanes by shuffle -- the
        // same operations on the same operands, once.
        if (nt >= 64) {
            const int lane = t & 63;
            double wj = pw[0], mj = pm[0], vj = pv[0];
#pragma unroll
            for (int j = 1; j < kMaxComp; j++) if (lane == j) { wj = pw[j]; mj = pm[j]; vj = pv[j]; }
            double b0 = 0.0, b1 = 0.0, b2 = 0.0, bc = 0.0, bl = 0.0;
            if (lane < k) {
                const double pc = 1.0 / sqrt(vj);   // _compute_precision_cholesky
                if (kFull) { b0 = mj * pc; b1 = pc; }
                else { const double prec = pc * pc; b0 = (mj * mj) * prec; b1 = mj * prec; b2 = prec; }
                bc = log(pc);
                bl = log(wj);
            }
#pragma unroll
            for (int j = 0; j < kMaxComp; j++) { a0[j] = __shfl(b0, j); a1[j] = __shfl(b1, j); a2[j] = __shfl(b2, j); cst[j] = __shfl(bc, j); lwt[j] = __shfl(bl, j); }
        } else {   // (workgroups of less than a wavefront: the host emulation of the tests)
#pragma unroll
            for (int j = 0; j < kMaxComp; j++) {
                a0[j] = 0.0; a1[j] = 0.0; a2[j] = 0.0; cst[j] = 0.0;
                if (j < k) {
                    const double pc = 1.0 / sqrt(pv[j]);   // _compute_precision_cholesky
                    if (kFull) { a0[j] = pm[j] * pc; a1[j] = pc; }
                    else { const double prec = pc * pc; a0[j] = (pm[j] * pm[j]) * prec; a1[j] = pm[j] * prec; a2[j] = prec; }
                    cst[j] = log(pc);
                }
                lwt[j] = j < k ? log(pw[j]) : 0.0;
            }
        }
#pragma unroll
        for (int c = 0; c < kFitStats; c++) v[c] = 0.0;
        for (int r = t; r < uniq; r += nt) {
            const double x = RUN_X(r);
            const double cw = RUN_C(r);
            double wl[kMaxComp], mx = -dinf();
#pragma unroll
            for (int j = 0; j < kMaxComp; j++) {
                wl[j] = -dinf();
                if (j < k) {
                    double lp;
                    if (kFull) { const double y = x * a1[j] - a0[j]; lp = y * y; }                       // _estimate_log_gaussian_prob, "full"
                    else lp = (a0[j] - 2.0 * (x * a1[j])) + (x * x) * a2[j];                             // ... "diag"
                    wl[j] = (-0.5 * (kLog2Pi + lp) + cst[j]) + lwt[j];
                    if (wl[j] > mx) mx = wl[j];
                }
            }
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < kMaxComp; j++) { wl[j] = j < k ? exp(wl[j] - mx) : 0.0; s += wl[j]; }
            v[3 * kMaxComp] += cw * (log(s) + mx);   // scipy.special.logsumexp
            if (!last) {
                const double inv = cw / s;
#pragma unroll
                for (int j = 0; j < kMaxComp; j++) {
                    if (j < k) {
                        const double rj = wl[j] * inv;
                        const double d = kFull ? x - pm[j] : x;
                        v[j] += rj;
                        v[kMaxComp + j] += rj * d;
                        v[2 * kMaxComp + j] += rj * (d * d);
                    }
                }
            }
        }
        block_reduce<kFitStats>(v, sh, nt);
        const double lb = v[3 * kMaxComp] / dn;
        if (last) { lower = lb; break; }
        double nks = 0.0;
        if (nt >= 64) {   // the M step's three divisions a component: lane j of every wavefront for component j (see the sweep's constants)
            const int lane = t & 63;
            double s0 = v[0], s1 = v[kMaxComp], s2 = v[2 * kMaxComp], mj = pm[0];
#pragma unroll
            for (int j = 1; j < kMaxComp; j++) if (lane == j) { s0 = v[j]; s1 = v[kMaxComp + j]; s2 = v[2 * kMaxComp + j]; mj = pm[j]; }
            const double nk = s0 + kEps10;
            const double m1 = s1 / nk;
            const double nv = (s2 / nk - m1 * m1) + kFitRegCovar;
            const double nm = kFull ? mj + m1 : m1;
#pragma unroll
            for (int j = 0; j < kMaxComp; j++) {
                if (j < k) {
                    pw[j] = __shfl(nk, j);
                    nks += pw[j];
                    pv[j] = __shfl(nv, j);
                    pm[j] = __shfl(nm, j);
                    if (!(pv[j] > 0.0)) failed = true;
                }
            }
            double wq = pw[0];
#pragma unroll
            for (int j = 1; j < kMaxComp; j++) if (lane == j) wq = pw[j];
            wq /= nks;
#pragma unroll
            for (int j = 0; j < kMaxComp; j++) if (j < k) pw[j] = __shfl(wq, j);
        } else {
#pragma unroll
            for (int j = 0; j < kMaxComp; j++) {
                if (j < k) {
                    const double nk = v[j] + kEps10;
                    const double m1 = v[kMaxComp + j] / nk;
                    pw[j] = nk;
                    nks += nk;
                    pv[j] = (v[2 * kMaxComp + j] / nk - m1 * m1) + kFitRegCovar;
                    pm[j] = kFull ? pm[j] + m1 : m1;
                    if (!(pv[j] > 0.0)) failed = true;
                }
            }
#pragma unroll
            for (int j = 0; j < kMaxComp; j++) if (j < k) pw[j] /= nks;
        }
        const double change = lb - lower;
        lower = lb;
        if (fabs(change) < kFitTol) last = true;   // converged: one more sweep scores the parameters just set
    }
    if (t != 0) return;
    if (kFull) {
        if (failed) { atomicCAS(F.err, 0, (int)TW_ERR_FIT); return; }
#pragma unroll
        for (int j = 0; j < kMaxComp; j++) {
            F.mix_p[(q * kMaxComp + j) * 3 + 0] = j < k ? pw[j] : 0.0;
            F.mix_p[(q * kMaxComp + j) * 3 + 1] = j < k ? pm[j] : 0.0;
            F.mix_p[(q * kMaxComp + j) * 3 + 2] = j < k ? 1.0 / sqrt(pv[j]) : 0.0;
        }
    } else {
        model[0] = failed ? dinf() : -2.0 * lower * dn + (double)(3 * k - 1) * log(dn);   // GaussianMixture.bic
#pragma unroll
        for (int j = 0; j < kMaxComp; j++) {
            model[1 + j] = j < k ? pw[j] : 0.0;
            model[1 + kMaxComp + j] = j < k ? pm[j] : 0.0;
            model[1 + 2 * kMaxComp + j] = j < k ? 1.0 / sqrt(pv[j]) : 0.0;
        }
    }
}

#undef RUN_X
#undef RUN_C

// n_selected = n_comps[argmin(bic)] over the fits that did not raise (first minimum, V3:780).  A row without samples keeps
// n = 0 (the reference stores (0, 0) and scores with a sigma = 0.001 Gaussian, V3:765-766); a row all of whose fits raised
// makes the reference raise too (argmin of an empty list): TW_ERR_FIT.
__global__ void k_fit_select(FitDev F) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= F.n_slots) return;
    int best = -1;
    double bic = dinf();
    for (int k = 0; k < kMaxComp; k++) {
        const double b = F.models[(q * kMaxComp + k) * kModelStride];
        if (b < bic) { bic = b; best = k; }
    }
    F.mix_n[q] = best + 1;
    if (best < 0 && F.slot_scored[q] && F.row_n[q] > 0) atomicCAS(F.err, 0, (int)TW_ERR_FIT);
    for (int j = 0; j < kMaxComp; j++) {   // (rows without samples; the others are written by the refit)
        F.mix_p[(q * kMaxComp + j) * 3 + 0] = 0.0;
        F.mix_p[(q * kMaxComp + j) * 3 + 1] = 0.0;
        F.mix_p[(q * kMaxComp + j) * 3 + 2] = 0.0;
    }
}

}  // namespace tw
