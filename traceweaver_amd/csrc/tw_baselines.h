// tw_baselines.h -- the reference's baseline predictors on the resident span table, so that a run of
// exps/exp1 / exp5 ("--predictor_indices 3,4,7,10") does not leave the device between ingest and the result files:
//
//   FCFS   algorithms/fcfs.py:10-26      request i takes the i-th call of every endpoint
//   WAP5   algorithms/wap5.py:257-351    most likely preceding request under an exponential delay model
//   vPath  algorithms/vpath.py:48-89     event sweep: a call belongs to the request that was last seen active
//
// The reference walks merged, time-sorted lists of Python objects; here both lists of a (unit, endpoint) are already
// sorted by start, so "position in the merged list" is a bisection and the sequential state of the sweeps is restated as
// "the last event before mine that sets it":
//
// WAP5.  merged list = incoming + the endpoint's outgoing spans, stable by start (incoming first on ties).
//   BuildDistributions (:257-275): a call's delay sample = distance to the nearest preceding incoming span, if within the
//   longest request duration of the service; samples accumulate per callee *name* over the services of a run (the class
//   keeps them), mean = statistics.mean (exact) -> the host adds up (sum, count) per name between the two kernels.
//   ScoreParents (:282-316): every call scans back over the spans within 4 x mean and marks every unpicked incoming span
//   it passes as picked; its parent is the best-scoring (= nearest) of them.  An incoming span before the previous call
//   of the endpoint and within reach was within that call's reach too, hence picked: the candidates of call x are the
//   incoming spans strictly between call x-1 and call x in the merged list, the winner the nearest start (among equal
//   starts the earliest position: the stable sort of (parent, score) keeps the last appended on top).
// vPath.  events (time, kind) with kinds request < call < return < response on ties (vpath.py:52-64); `latest` is set by
//   request starts (to the request), responses (to none) and returns (to the request the call truly belongs to); a call is
//   given to `latest`.  For a call at time t the last setter is the latest of: the last request start <= t, the last
//   response < t, the last return < t over all endpoints -- three bisections (end times through the rank arrays of
//   k_rank_ends).  When several calls are given to one request the last one in event order stays (atomicMax of the index).
//
// Skip-mode batches (lists no longer sorted) and ArrivalOrder stay host code (traceweaver_amd/baselines.py, which is what
// the tests pin these kernels to).
#pragma once
#include "tw_kernels.h"

namespace tw {

struct BaseDev {
    const int32_t* truth;        // tw_set_truth (vPath: which request a returning call belongs to)
    int32_t* parent;             // [n_ie] result
    int32_t* count;              // [n_ie] WAP5: calls given to (endpoint, request)
    int64_t* unit_maxdur;        // [n_units] longest request of the unit, microseconds
    long long* delay_sum;        // [n_units][kMaxEp] WAP5 delay samples of this batch, in timestamp units
    int32_t* delay_cnt;          // [n_units][kMaxEp]
    const double* mean;          // [n_units][kMaxEp] WAP5: mean delay per (unit, endpoint), microseconds (host: all services so far)
    const int32_t* in_rank_inv;  // [n_in_total] incoming span at every position of the sorted end times
    const int32_t* out_rank_inv; // [n_out_total]
    int32_t* call_request;       // [n_out_total] WAP5: the request every call was given to (-1 = none: "Spontaneous" or nothing in reach)
    int32_t* owner;              // [n_out_total] vPath: request whose true call the span is (smallest on duplicates), kNoOwner = none
};

__global__ void k_base_fcfs(Dev P, BaseDev B) {
    const TileDev Tl = P.tiles[blockIdx.x];
    const UnitDev& U = P.units[Tl.unit];
    const int i = Tl.first + threadIdx.x;
    if (i >= U.n_in) return;
    for (int e = 0; e < U.E; e++) B.parent[ie_index(U, e, i)] = i < (int)(U.ep_off[e + 1] - U.ep_off[e]) ? i : -1;
}

__global__ void k_base_prepare(Dev P, BaseDev B) {   // longest request per unit; owner of every outgoing span; parent = -1, count = 0
    const TileDev Tl = P.tiles[blockIdx.x];
    const UnitDev& U = P.units[Tl.unit];
    const int i = Tl.first + threadIdx.x;
    if (i >= U.n_in) return;
    const int64_t d = P.in_end[U.in_off + i] - P.in_start[U.in_off + i];
    const long long dur = U.float_time ? (long long)rint((double)d * U.tscale) : (long long)d;   // durations are the integers they were
    atomicMax((long long*)&B.unit_maxdur[Tl.unit], dur);
    for (int e = 0; e < U.E; e++) {
        B.parent[ie_index(U, e, i)] = -1;
        B.count[ie_index(U, e, i)] = 0;
        if (B.truth != nullptr) {
            const int x = B.truth[ie_index(U, e, i)];
            if (x >= 0) atomicMin(&B.owner[U.ep_off[e] + x], i);
        }
    }
}

// incoming spans with start <= t
__device__ __forceinline__ int in_upper(const Dev& P, const UnitDev& U, int64_t t) { return upper_bound_i64(P.in_start + U.in_off, U.n_in, t); }

__global__ void k_wap5_delays(Dev P, BaseDev B) {
    const TileDev Tl = P.tiles[blockIdx.x];
    const UnitDev& U = P.units[Tl.unit];
    const int x = Tl.first + threadIdx.x;
    const bool live = x < U.n_in;   // (the lanes beyond the unit stay for the wavefront reduction below)
    const double large = (double)B.unit_maxdur[Tl.unit];
    for (int e = 0; e < U.E; e++) {
        long long d = 0;
        bool have = false;
        if (live && x < (int)(U.ep_off[e + 1] - U.ep_off[e])) {
            const int64_t sent = P.out_start[U.ep_off[e] + x];
            const int ub = in_upper(P, U, sent);
            if (ub > 0) {
                d = (long long)(sent - P.in_start[U.in_off + ub - 1]);
                have = !((double)d * U.tscale > large);
            }
        }
        // one atomic per wavefront and endpoint
        long long s = have ? d : 0;
        int c = have ? 1 : 0;
        for (int off = 32; off >= 1; off >>= 1) if (off < (int)blockDim.x) { s += __shfl_down(s, off); c += __shfl_down(c, off); }
        if ((threadIdx.x & 63) == 0 && c > 0) {
            atomicAdd((unsigned long long*)&B.delay_sum[(int64_t)Tl.unit * kMaxEp + e], (unsigned long long)s);
            atomicAdd(&B.delay_cnt[(int64_t)Tl.unit * kMaxEp + e], c);
        }
    }
}

__global__ void k_wap5_parents(Dev P, BaseDev B) {
    const TileDev Tl = P.tiles[blockIdx.x];
    const UnitDev& U = P.units[Tl.unit];
    const int x = Tl.first + threadIdx.x;
    if (x >= U.n_in) return;
    for (int e = 0; e < U.E; e++) {
        if (x >= (int)(U.ep_off[e + 1] - U.ep_off[e])) continue;
        B.call_request[U.ep_off[e] + x] = -1;
        const double limit = 4.0 * B.mean[(int64_t)Tl.unit * kMaxEp + e];   // MAGIC_DELAY * mean (wap5.py:286)
        const int64_t sent = P.out_start[U.ep_off[e] + x];
        const int hi = in_upper(P, U, sent);                                          // incoming spans before this call in the merged list
        const int lo = x > 0 ? in_upper(P, U, P.out_start[U.ep_off[e] + x - 1]) : 0;  // ... before the previous call
        if (hi <= lo) continue;                                                        // nothing unpicked in reach: "Spontaneous" at best
        const int64_t near = P.in_start[U.in_off + hi - 1];
        if ((double)(sent - near) * U.tscale > limit) continue;
        int j = lower_bound_i64(P.in_start + U.in_off, U.n_in, near);                 // the earliest position among equal starts ...
        if (j < lo) j = lo;                                                            // ... that is still a candidate
        B.parent[ie_index(U, e, j)] = x;
        B.call_request[U.ep_off[e] + x] = j;
        atomicAdd(&B.count[ie_index(U, e, j)], 1);
    }
}
// helpers/utils.py:68-73: a request with several options for an endpoint counts as wrong (-2), with none as ("NA", "NA")
__global__ void k_wap5_finish(Dev P, BaseDev B) {
    const TileDev Tl = P.tiles[blockIdx.x];
    const UnitDev& U = P.units[Tl.unit];
    const int i = Tl.first + threadIdx.x;
    if (i >= U.n_in) return;
    for (int e = 0; e < U.E; e++) {
        const int c = B.count[ie_index(U, e, i)];
        if (c != 1) B.parent[ie_index(U, e, i)] = c == 0 ? -1 : -2;
    }
}

// position -> span of the end-time ranking (k_rank_ends): the inverse of i -> i + delta[i]
__global__ void k_base_rank_inverse(Dev P, const int32_t* d_in, const int32_t* d_out, int32_t* inv_in, int32_t* inv_out) {
    const TileDev Tl = P.tiles[blockIdx.x];
    const UnitDev& U = P.units[Tl.unit];
    const int i = Tl.first + threadIdx.x;
    if (i >= U.n_in) return;
    inv_in[U.in_off + i + d_in[U.in_off + i]] = i;
    for (int e = 0; e < U.E; e++) inv_out[U.ep_off[e] + i + d_out[U.ep_off[e] + i]] = i;
}

__global__ void k_base_vpath(Dev P, BaseDev B) {
    const TileDev Tl = P.tiles[blockIdx.x];
    const UnitDev& U = P.units[Tl.unit];
    const int x = Tl.first + threadIdx.x;
    if (x >= U.n_in) return;
    for (int e = 0; e < U.E; e++) {
        const int64_t t = P.out_start[U.ep_off[e] + x];
        // the last setter before (t, call): kind order on equal times request(1) < call(2) < return(3) < response(4)
        int64_t best_t = INT64_MIN;
        int best_kind = 0, best_ord = -1, latest = -1;
        const int ub = in_upper(P, U, t);                                   // request starts at <= t come before the call
        if (ub > 0) { best_t = P.in_start[U.in_off + ub - 1]; best_kind = 1; latest = ub - 1; }
        const int rb = lower_bound_i64(P.in_end_sorted + U.in_off, U.n_in, t);   // responses at < t
        if (rb > 0) {
            const int64_t tr = P.in_end_sorted[U.in_off + rb - 1];
            if (tr > best_t || (tr == best_t && 4 > best_kind)) { best_t = tr; best_kind = 4; latest = -1; }
        }
        for (int f = 0; f < U.E; f++) {                                     // returns at < t, every endpoint
            const int64_t* se = P.out_end_sorted + U.ep_off[f];
            int q = lower_bound_i64(se, (int)(U.ep_off[f + 1] - U.ep_off[f]), t) - 1;   // (equal end times keep index order: the last one)
            while (q >= 0 && B.owner[U.ep_off[f] + B.out_rank_inv[U.ep_off[f] + q]] == kNoOwner) q--;   // a return nobody owns sets nothing
            if (q < 0) continue;
            const int64_t tr = se[q];
            // among equal (time, kind) the reference's list order decides: endpoints in partition-key order
            const bool later = tr > best_t || (tr == best_t && (3 > best_kind || (best_kind == 3 && (int)U.key_rank[f] > best_ord)));
            if (later) { best_t = tr; best_kind = 3; best_ord = (int)U.key_rank[f]; latest = B.owner[U.ep_off[f] + B.out_rank_inv[U.ep_off[f] + q]]; }
        }
        if (latest >= 0) atomicMax(&B.parent[ie_index(U, e, latest)], x);
    }
}

}  // namespace tw
