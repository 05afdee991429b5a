// tw_fit.h -- device-side refit of the per-edge delay mixtures between the two passes
// (ComputeEpPairDistParams5, traceweaver_v3.py:764-786).
//
// The reference fits, per scored edge, 1..min(5, #unique) one-dimensional Gaussian mixtures with
// scikit-learn (k-means++ initialisation drawn from numpy's *unseeded* global RNG), keeps the
// component count with the smallest BIC and refits it.  That procedure is not reproducible run to run
// (SURVEY.md hazard H9: +-1 pp end-to-end accuracy between identical runs), so there is no bit-level
// target to hit.  The device refit keeps the model family, the EM iteration, its stopping rule
// (|delta mean log-likelihood| < 1e-3, <= 100 iterations, reg_covar 1e-6) and the BIC selection, and
// replaces the random initialisation by a deterministic one: the k equal-count buckets of the sorted
// samples.  One workgroup runs one (edge, k) fit start to finish, so every reduction has a fixed
// order and the result is a pure function of the samples.
//
// Gaps are integer microseconds, so a row of 10^5 samples holds only 10^2..10^4 distinct values:
// k_fit_compress turns every sorted row into (value, multiplicity) runs once, and all EM sums run over
// the runs with the multiplicity as weight (the same sums, grouped by value).
//
// traceweaver_amd/gmm.py::fit_edge_sklearn keeps the reference procedure for comparison; pass 2 is
// bit-exact against the oracle for *any* mixture table (tests feed the fitted tables to both).
#pragma once
#include "tw_device.h"

namespace tw {

constexpr int kFitThreads = 512;
constexpr int kFitWaves = kFitThreads / 64;
constexpr int kFitStats = 3 * kMaxComp + 1;  // per component: nk, sum r*(x-c), sum r*(x-c)^2; + log-likelihood
constexpr int kFitMaxIter = 100;
constexpr double kFitTol = 1.0e-3;
constexpr double kFitRegCovar = 1.0e-6;
constexpr int kModelStride = 1 + 3 * kMaxComp;  // bic, w[5], mu[5], var[5]

struct FitDev {
    const UnitDev* units;
    int32_t n_units;
    int64_t n_slots;
    const double* sorted;   // gap rows sorted ascending, NaN (dropped) last; row q of unit u at gs_off[u] + q*n_in
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
};

// deterministic block reduction of `cnt` doubles per thread: lanes of a wavefront by shuffles (fixed
// butterfly order), wavefronts by thread 0 in wavefront order.  Result broadcast to every thread.
template <int cnt>
__device__ inline void block_reduce(double (&vals)[cnt], double* sh) {
    const int t = threadIdx.x, nt = blockDim.x;
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
__global__ void __launch_bounds__(kCoop) k_fit_compress(FitDev F) {
    constexpr int R = 4;  // chunks of blockDim samples per round: their loads are in flight together
    __shared__ int32_t wave_heads[R][kCoop / 64];
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

__global__ void __launch_bounds__(kFitThreads) k_fit_em(FitDev F) {
    __shared__ double sh[kFitStats * kFitWaves];
    __shared__ double par[3 * kMaxComp];  // w, mu, var
    const int64_t q = blockIdx.x / kMaxComp;
    const int k = (int)(blockIdx.x % kMaxComp) + 1;
    const int t = threadIdx.x, nt = blockDim.x;
    const UnitDev& U = F.units[F.slot_unit[q]];
    const int64_t row = F.gs_off[F.slot_unit[q]] + (q - U.slot_off) * (int64_t)U.n_in;
    const double* xv = F.uval + row;      // distinct values, ascending
    const int32_t* xa = F.ustart + row;   // first index of each value in the sorted row
    double* model = F.models + (q * kMaxComp + (k - 1)) * kModelStride;
    const int n = F.row_n[q], uniq = F.row_uniq[q];
    if (!F.slot_scored[q] || n == 0 || k > uniq || k > kMaxComp) {
        if (t == 0) model[0] = dinf();
        return;
    }
    // every per-thread array below is indexed by compile-time constants only (loops over the components are
    // unrolled to kMaxComp and predicated with j < k), so nothing spills to scratch memory
    // initialisation: equal-count buckets of the sorted samples -- sample i belongs to bucket floor(i*k/n),
    // i.e. bucket c owns the index range [ceil(c*n/k), ceil((c+1)*n/k))
    double b2[2 * kMaxComp];
#pragma unroll
    for (int c = 0; c < 2 * kMaxComp; c++) b2[c] = 0.0;
    for (int r = t; r < uniq; r += nt) {
        const int64_t a = xa[r], b = r + 1 < uniq ? xa[r + 1] : n;
        const double xi = xv[r];
#pragma unroll
        for (int c = 0; c < kMaxComp; c++) {
            if (c < k) {
                const int64_t lo = ((int64_t)c * n + k - 1) / k, hi = ((int64_t)(c + 1) * n + k - 1) / k;
                const int64_t ov = (b < hi ? b : hi) - (a > lo ? a : lo);
                if (ov > 0) { b2[c] += (double)ov; b2[kMaxComp + c] += (double)ov * xi; }
            }
        }
    }
    block_reduce<2 * kMaxComp>(b2, sh);
    if (t == 0) {
#pragma unroll
        for (int j = 0; j < kMaxComp; j++) if (j < k) { par[j] = b2[j] / (double)n; par[kMaxComp + j] = b2[kMaxComp + j] / b2[j]; }
    }
    __syncthreads();
    double b1[kMaxComp];
#pragma unroll
    for (int c = 0; c < kMaxComp; c++) b1[c] = 0.0;
    for (int r = t; r < uniq; r += nt) {
        const int64_t a = xa[r], b = r + 1 < uniq ? xa[r + 1] : n;
        const double xi = xv[r];
#pragma unroll
        for (int c = 0; c < kMaxComp; c++) {
            if (c < k) {
                const int64_t lo = ((int64_t)c * n + k - 1) / k, hi = ((int64_t)(c + 1) * n + k - 1) / k;
                const int64_t ov = (b < hi ? b : hi) - (a > lo ? a : lo);
                const double d = xi - par[kMaxComp + c];
                if (ov > 0) b1[c] += (double)ov * (d * d);
            }
        }
    }
    block_reduce<kMaxComp>(b1, sh);
    if (t == 0) {
#pragma unroll
        for (int j = 0; j < kMaxComp; j++) if (j < k) par[2 * kMaxComp + j] = b1[j] / (par[j] * (double)n) + kFitRegCovar;
    }
    __syncthreads();
    // EM over the runs, multiplicity as weight
    double prev_lb = -dinf();
    for (int iter = 0; iter <= kFitMaxIter; iter++) {
        double lw[kMaxComp], mu[kMaxComp], iv[kMaxComp], v[kFitStats];
#pragma unroll
        for (int j = 0; j < kMaxComp; j++) {
            if (j < k) {
                mu[j] = par[kMaxComp + j];
                iv[j] = 1.0 / par[2 * kMaxComp + j];
                lw[j] = log(par[j]) - 0.5 * (kLog2Pi + log(par[2 * kMaxComp + j]));
            } else { mu[j] = 0.0; iv[j] = 0.0; lw[j] = -dinf(); }
        }
#pragma unroll
        for (int c = 0; c < kFitStats; c++) v[c] = 0.0;
        for (int r = t; r < uniq; r += nt) {
            const double xi = xv[r];
            const double cw = (double)((r + 1 < uniq ? xa[r + 1] : n) - xa[r]);
            double lp[kMaxComp], mx = -dinf();
#pragma unroll
            for (int j = 0; j < kMaxComp; j++) {
                if (j < k) {
                    const double d = xi - mu[j];
                    lp[j] = lw[j] - 0.5 * d * d * iv[j];
                    if (lp[j] > mx) mx = lp[j];
                }
            }
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < kMaxComp; j++) if (j < k) { lp[j] = exp(lp[j] - mx); s += lp[j]; }
            v[3 * kMaxComp] += cw * (mx + log(s));
            const double inv = cw / s;
#pragma unroll
            for (int j = 0; j < kMaxComp; j++) {
                if (j < k) {
                    const double r2 = lp[j] * inv, d = xi - mu[j];
                    v[j] += r2;
                    v[kMaxComp + j] += r2 * d;
                    v[2 * kMaxComp + j] += r2 * d * d;
                }
            }
        }
        block_reduce<kFitStats>(v, sh);
        const double lb = v[3 * kMaxComp] / (double)n;  // mean log-likelihood under the current parameters
        // last round only evaluates the likelihood of the final parameters (sklearn's bic() re-scores)
        const bool stop = (iter == kFitMaxIter) || (fabs(lb - prev_lb) < kFitTol);
        if (stop) {
            if (t == 0) {
                model[0] = -2.0 * lb * (double)n + (double)(3 * k - 1) * log((double)n);
#pragma unroll
                for (int j = 0; j < kMaxComp; j++) {
                    model[1 + j] = j < k ? par[j] : 0.0;
                    model[1 + kMaxComp + j] = j < k ? par[kMaxComp + j] : 0.0;
                    model[1 + 2 * kMaxComp + j] = j < k ? par[2 * kMaxComp + j] : 1.0;
                }
            }
            return;
        }
        prev_lb = lb;
        __syncthreads();
        if (t == 0) {  // M step (shifted moments: c = previous mean of the component)
            double ws = 0.0;
#pragma unroll
            for (int j = 0; j < kMaxComp; j++) {
                if (j < k) {
                    const double nk = v[j] + 10.0 * 2.220446049250313e-16;
                    const double dm = v[kMaxComp + j] / nk;
                    par[j] = nk / (double)n;
                    par[kMaxComp + j] = mu[j] + dm;
                    par[2 * kMaxComp + j] = v[2 * kMaxComp + j] / nk - dm * dm + kFitRegCovar;
                    ws += par[j];
                }
            }
#pragma unroll
            for (int j = 0; j < kMaxComp; j++) if (j < k) par[j] /= ws;
        }
        __syncthreads();
    }
}

// smallest BIC wins (first minimum, like np.argmin over n = 1..5)
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
    for (int j = 0; j < kMaxComp; j++) {
        double w = 0.0, mu = 0.0, pc = 0.0;
        if (best >= 0 && j <= best) {
            const double* m = F.models + (q * kMaxComp + best) * kModelStride;
            w = m[1 + j]; mu = m[1 + kMaxComp + j]; pc = 1.0 / sqrt(m[1 + 2 * kMaxComp + j]);
        }
        F.mix_p[(q * kMaxComp + j) * 3 + 0] = w;
        F.mix_p[(q * kMaxComp + j) * 3 + 1] = mu;
        F.mix_p[(q * kMaxComp + j) * 3 + 2] = pc;
    }
}

}  // namespace tw
