// tw_eval.h -- the two neighbours of the hot path that work on the same resident arrays:
//   k_find_order   FindOrder, executor.py:214-285: the call-order DAG a service's scorer is given
//   k_evaluate     AccuracyForService / TopKAccuracyForService / AccuracyEndToEnd, helpers/utils.py:62-117
#pragma once
#include "tw_device.h"

namespace tw {

// FindOrder (executor.py:214-285) keeps the edge a -> b iff in *every* request the call to a finished no later
// than the call to b started (x.start + x.duration > y.start removes it, executor.py:249-252).  One thread per
// request; bit a*kMaxEp+b of viol[unit] is set when some request violates a -> b.  Requests with an unknown
// true child (index < 0) say nothing about the pairs that involve it.
struct OrderDev {
    int32_t n_units;
    const int64_t* unit_in_off;   // [n_units + 1]
    const int32_t* unit_E;        // [n_units]
    const int64_t* ep_base;       // [n_units]  index of the unit's first endpoint in ep_off
    const int64_t* ep_off;        // [sum E + 1 per unit ...] global offsets into out_start / out_end (tw_batch layout)
    const int64_t* ie_off;        // [n_units]  base of the unit's [E][n_in] block in truth
    const int64_t *out_start, *out_end;
    const int32_t* truth;         // true child index per (endpoint, request)
    unsigned long long* viol;     // [n_units]
};

__global__ void k_find_order(OrderDev O) {
    const int u = blockIdx.y;
    const int64_t n = O.unit_in_off[u + 1] - O.unit_in_off[u];
    const int E = O.unit_E[u];
    const int64_t* eo = O.ep_off + O.ep_base[u];
    unsigned long long v = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t st[kMaxEp], en[kMaxEp];
        bool have[kMaxEp];
#pragma unroll
        for (int e = 0; e < kMaxEp; e++) {
            have[e] = false; st[e] = 0; en[e] = 0;
            if (e < E) {
                const int32_t x = O.truth[O.ie_off[u] + (int64_t)e * n + i];
                if (x >= 0) { have[e] = true; st[e] = O.out_start[eo[e] + x]; en[e] = O.out_end[eo[e] + x]; }
            }
        }
#pragma unroll
        for (int a = 0; a < kMaxEp; a++)
#pragma unroll
            for (int b = 0; b < kMaxEp; b++)
                if (a != b && a < E && b < E && have[a] && have[b] && en[a] > st[b]) v |= 1ull << (a * kMaxEp + b);
    }
    for (int off = 32; off >= 1; off >>= 1)
        if (off < (int)blockDim.x) v |= __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0 && (v & ~O.viol[u]) != 0) atomicOr(&O.viol[u], v);
}

// Accuracy against ground truth on the resident result arrays, one lane per incoming span:
//   exact  every endpoint's chosen span is the true one                         utils.py:62-79
//   top-k  some tuple of the span's top-5 list equals the true tuple            utils.py:81-97
// and, when a trace index per incoming span is given, the per-trace flags of AccuracyEndToEnd /
// TopKAccuracyEndToEnd (utils.py:99-145): a trace is correct iff no span of it is wrong in any unit.
// counts[unit] = {-, requests that are wrong, requests whose list misses the true tuple, requests left unassigned}.
__global__ void __launch_bounds__(kTile) k_evaluate(Dev P, const int32_t* truth, const int32_t* in_trace, unsigned long long* counts,
                                                   uint8_t* trace_bad, uint8_t* trace_bad_topk) {
    const TileDev Tl = P.tiles[blockIdx.x];
    const UnitDev& U = P.units[Tl.unit];
    const int i = Tl.first + threadIdx.x;
    const bool live = i < U.n_in;
    bool exact = live, topk = false, unassigned = false;
    if (live) {
        const int64_t g = U.in_off + i;
        for (int e = 0; e < U.E; e++) {
            const int32_t p = P.parent[ie_index(U, e, i)];
            exact = exact && p == truth[ie_index(U, e, i)];
            unassigned = unassigned || p < 0;
        }
        const int n = P.tk_n[g];
        for (int k = 0; k < n && !topk; k++) {
            bool same = true;
            for (int e = 0; e < U.E && same; e++) {
                const int32_t x = P.tk_idx[tk_index(U, k, e, i)];
                same = (x <= -TW_SKIP_BASE ? -2 : x) == truth[ie_index(U, e, i)];
            }
            topk = same;
        }
        if (in_trace != nullptr) {
            const int32_t tr = in_trace[g];
            if (tr >= 0) {
                if (!exact) trace_bad[tr] = 1;
                if (!topk) trace_bad_topk[tr] = 1;
            }
        }
    }
    // what is counted is the rare outcome (wrong / not in the list / unassigned): most wavefronts add nothing
    const unsigned long long me = __ballot(live && !exact), mt = __ballot(live && !topk), mu = __ballot(unassigned);
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* c = counts + (int64_t)Tl.unit * 4;
        if (me) atomicAdd(&c[1], (unsigned long long)__popcll(me));
        if (mt) atomicAdd(&c[2], (unsigned long long)__popcll(mt));
        if (mu) atomicAdd(&c[3], (unsigned long long)__popcll(mu));
    }
}

// Streaming copy, 16 B per lane and load (tw_measure_hbm_copy): the HBM rate a plain kernel reaches on this device, the
// measured ceiling quoted next to the 8 TB/s specification in the roofline figures.  U loads of a thread are in flight
// before its first store; with U = 1 and one workgroup per 4 KB (no grid-stride loop) consecutive workgroups walk
// consecutive DRAM pages: 6.2 TB/s, where the grid-stride form reaches 4.4-4.7.  NT: stores that bypass the caches.
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (NT) {
                __builtin_nontemporal_store(v[u].x, &dst[i + u * stride].x); __builtin_nontemporal_store(v[u].y, &dst[i + u * stride].y);
                __builtin_nontemporal_store(v[u].z, &dst[i + u * stride].z); __builtin_nontemporal_store(v[u].w, &dst[i + u * stride].w);
            } else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

__global__ void k_count_flags(const uint8_t* a, const uint8_t* b, int64_t n, unsigned long long* out) {  // out[0] += #a[i]==0, out[1] += #b[i]==0
    unsigned long long ca = 0, cb = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        ca += a[i] == 0;
        cb += b[i] == 0;
    }
    for (int off = 32; off >= 1; off >>= 1)
        if (off < (int)blockDim.x) { ca += __shfl_down(ca, off); cb += __shfl_down(cb, off); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], ca); atomicAdd(&out[1], cb); }
}

}  // namespace tw
