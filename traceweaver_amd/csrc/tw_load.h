// tw_load.h -- load scaling on the resident span table: the reference's repeat_change_spans
// (helpers/transforms.py:10-40), which `--compress_factor N` applies to every service before the predictor runs
// (executor.py:1086-1097,1146-1148; exps/exp5/run_experiment.sh:60-156 runs six load levels per call graph).
//
//   per request i of a unit with load factor f:   x  = in.start / f                       (binary64 division)
//                                                 y  = x + (out.start - in.start)         for the request's own call at every endpoint
//                                                 durations untouched; ends are fl(start + duration), as every use forms them
//   every list re-sorted by (start, end), stable from trace-id order; ground truth re-indexed (helpers/utils.py:22-32)
//
// The scaled timestamps are binary64 values; the engine keeps them exactly as int64 multiples of 2^-k with the smallest k
// that represents every value of the unit (tw_batch.unit_time_scale semantics, see traceweaver_amd/transforms.py).
// All of it is elementwise HBM traffic plus radix sorts of (key, index) pairs: 16 B/span read, 16 B/span written, and
// 12 B per span and sort pass.  The original table stays resident, so every load level of one corpus costs no upload.
#pragma once
#include "tw_device.h"

namespace tw {

struct LoadDev {
    const UnitDev* units;
    const TileDev* tiles;
    const int64_t *is, *ie, *os, *oe;   // the original table (integer microseconds)
    const int32_t* truth;                // [ie_off + e*n_in + i] the request's own call at endpoint e
    const int32_t* in_trace;             // [n_in_total] or null
    const int32_t* trace_rank;           // [n_in_total] order of the trace ids inside the unit, or null (= current order)
    const int32_t* factor;               // [n_units]
    const int32_t* seg_base;             // [n_units] number of endpoint lists of earlier units
    double *x, *xe, *y, *ye;             // scaled timestamps (binary64), later overwritten in place by their int64 images
    int32_t *rank_in, *rank_out;         // tie order of every span (trace order of its request)
    int32_t *row_in, *row_out;           // list id of every span (unit / endpoint list)
    int32_t* owner_req;                  // [n_out_total] request that owns the outgoing span (-1: none)
    int32_t* kmax;                       // [n_units] binary digits after the point the unit needs
    int32_t* err;
};

// digits after the binary point of a binary64 value: smallest k >= 0 with v * 2^k an integer
__device__ __forceinline__ int frac_bits(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const int ex = (int)((b >> 52) & 0x7ff);
    unsigned long long m = b & 0xfffffffffffffull;
    if (ex == 0 && m == 0) return 0;                 // +-0
    int E2;
    if (ex == 0) E2 = -1074;                         // subnormal: m * 2^-1074
    else { m |= 1ull << 52; E2 = ex - 1075; }        // m * 2^(ex-1075)
    const int tz = __ffsll((long long)m) - 1;
    const int k = -(E2 + tz);
    return k > 0 ? k : 0;
}

__device__ __forceinline__ void unit_max(int32_t* slot, int v) {   // one atomic per wavefront
    for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(v, o); v = w > v ? w : v; }
    if ((threadIdx.x & 63) == 0 && v > 0) atomicMax(slot, v);
}

// incoming spans: x = start / f, end image, tie rank, row id
__global__ void __launch_bounds__(kTile) k_load_scale_in(LoadDev L) {
    const TileDev T = L.tiles[blockIdx.x];
    const UnitDev& U = L.units[T.unit];
    const int i = T.first + (int)threadIdx.x;
    int k = 0;
    if (i < U.n_in) {
        const int64_t g = U.in_off + i;
        const double x = (double)L.is[g] / (double)L.factor[T.unit];   // transforms.py:21
        const double xe = x + (double)(L.ie[g] - L.is[g]);
        L.x[g] = x; L.xe[g] = xe;
        L.rank_in[g] = L.trace_rank != nullptr ? L.trace_rank[g] : i;
        L.row_in[g] = T.unit;
        k = max(frac_bits(x), frac_bits(xe));
    }
    unit_max(&L.kmax[T.unit], k);
}

// outgoing spans, addressed through their owning request: y = x + (out.start - in.start)   (transforms.py:30)
__global__ void __launch_bounds__(kTile) k_load_scale_out(LoadDev L) {
    const TileDev T = L.tiles[blockIdx.x];
    const UnitDev& U = L.units[T.unit];
    const int i = T.first + (int)threadIdx.x;
    int k = 0;
    if (i < U.n_in) {
        const int64_t g = U.in_off + i;
        const double x = L.x[g];
        const int rank = L.trace_rank != nullptr ? L.trace_rank[g] : i;
        for (int e = 0; e < U.E; e++) {
            const int own = L.truth[U.ie_off + (int64_t)e * U.n_in + i];
            if (own < 0 || own >= U.n_in) { atomicCAS(L.err, 0, (int)TW_ERR_ARG); continue; }
            const int64_t j = U.ep_off[e] + own;
            const double y = x + (double)(L.os[j] - L.is[g]);
            const double ye = y + (double)(L.oe[j] - L.os[j]);
            L.y[j] = y; L.ye[j] = ye;
            L.rank_out[j] = rank;
            L.row_out[j] = L.seg_base[T.unit] + e;
            L.owner_req[j] = i;
            const int kk = max(frac_bits(y), frac_bits(ye));
            k = kk > k ? kk : k;
        }
    }
    unit_max(&L.kmax[T.unit], k);
}

// binary64 -> int64 image at the unit's scale (exact by construction of kmax; |v| 2^k must stay below 2^62), in place;
// also checks that the requests' own calls are a permutation of every endpoint list (helpers/transforms.py:25-29 pairs
// position i of every partition after sorting by trace id and asserts equal trace ids)
__device__ __forceinline__ int64_t to_image(double v, int k, int32_t* err) {
    const double s = ldexp(v, k);
    if (!(fabs(s) < 4611686018427387904.0)) { atomicCAS(err, 0, (int)TW_ERR_ARG); return 0; }
    return (int64_t)s;
}

__global__ void __launch_bounds__(kTile) k_load_to_int(LoadDev L) {
    const TileDev T = L.tiles[blockIdx.x];
    const UnitDev& U = L.units[T.unit];
    const int i = T.first + (int)threadIdx.x;
    if (i >= U.n_in) return;
    const int k = L.kmax[T.unit];
    const int64_t g = U.in_off + i;
    ((int64_t*)L.x)[g] = to_image(L.x[g], k, L.err);
    ((int64_t*)L.xe)[g] = to_image(L.xe[g], k, L.err);
    for (int e = 0; e < U.E; e++) {
        const int64_t j = U.ep_off[e] + i;
        const int req = L.owner_req[j];
        if (req < 0 || L.truth[U.ie_off + (int64_t)e * U.n_in + req] != i) { atomicCAS(L.err, 0, (int)TW_ERR_ARG); continue; }
        ((int64_t*)L.y)[j] = to_image(L.y[j], k, L.err);
        ((int64_t*)L.ye)[j] = to_image(L.ye[j], k, L.err);
    }
}

// sort keys of one stable pass, gathered through the current permutation; signed values are biased to unsigned order
__global__ void k_load_keys64(const int64_t* src, const uint32_t* perm, int64_t n, unsigned long long* keys) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x)
        keys[q] = (unsigned long long)src[perm[q]] ^ (1ull << 63);
}
__global__ void k_load_keys32(const int32_t* src, const uint32_t* perm, int64_t n, unsigned long long* keys) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x)
        keys[q] = (unsigned long long)(uint32_t)src[perm[q]];
}
__global__ void k_load_iota(uint32_t* perm, int64_t n) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) perm[q] = (uint32_t)q;
}

// the sorted table: new position q holds old span perm[q]; pos[old] = q
__global__ void k_load_place(const uint32_t* perm, int64_t n, const int64_t* s, const int64_t* e, int64_t* s_out, int64_t* e_out, int32_t* pos) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t o = perm[q];
        s_out[q] = s[o]; e_out[q] = e[o];
        pos[o] = (int32_t)q;
    }
}

// ground truth and trace numbers in the new order (GetGroundTruth after the re-sort, helpers/utils.py:22-32); the
// permutations are handed back as list-local indices (old position of every new position)
__global__ void __launch_bounds__(kTile) k_load_retruth(LoadDev L, const int32_t* pos_in, const int32_t* pos_out, int32_t* truth_out, int32_t* trace_out) {
    const TileDev T = L.tiles[blockIdx.x];
    const UnitDev& U = L.units[T.unit];
    const int i = T.first + (int)threadIdx.x;
    if (i >= U.n_in) return;
    const int64_t g = U.in_off + i;
    const int ni = pos_in[g] - (int)U.in_off;
    if (trace_out != nullptr) trace_out[U.in_off + ni] = L.in_trace[g];
    for (int e = 0; e < U.E; e++) {
        const int own = L.truth[U.ie_off + (int64_t)e * U.n_in + i];
        truth_out[U.ie_off + (int64_t)e * U.n_in + ni] = pos_out[U.ep_off[e] + own] - (int)U.ep_off[e];
    }
}
__global__ void k_load_local(uint32_t* perm, const int32_t* row, const uint32_t* row_base, int64_t n) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x)
        perm[q] = (uint32_t)((int64_t)perm[q] - row_base[row[perm[q]]]);
}

}  // namespace tw
