// tw_skip.h -- skip mode of the span->parent assignment (exps/exp2: some requests did not call an endpoint).
//
// Reference (paths under /root/reference/src/trace_reconstructor/ports/python/algorithms/):
//   TallySkipSpans / WaterFill        traceweaver_v3.py:853-989   host (traceweaver_amd/skipmode.py): a few dozen windows
//   BuildDistributions                traceweaver_v3.py:108-172   k_build_distributions: one thread per span of the merged sweep
//   FindAssignments, one iteration    traceweaver_v3.py:1141-1219 k_skip_walk
//   FetchSkipFromWindow               traceweaver_v3.py:820-842   round-robin draw per partial tuple that reaches the sentinel
//   DfsTraverseX with the sentinel    traceweaver_v3.py:292-351
//   ScoreAssignmentAsPerInvocationGraph(normalized=True), skipped-predecessor fallback   traceweaver_v1.py:259-361
//
// A skip span is "the request did not call this endpoint"; the skip spans of a 30-request time window form a pool
// per endpoint, and every partial tuple that reaches the end of an endpoint's list draws the least-used span of the
// pool -- which one it draws depends on how many draws every earlier request (both of its enumerations) made.  That
// counter chain, and the removal of chosen spans before the next window is enumerated, make the reference's loop
// sequential per service; skip-mode inputs are small (exp2: 1000 requests), so k_skip_walk keeps that order: one
// wavefront per service walks the requests, lane 0 enumerates (candidate lists of a request hold a handful of spans),
// all lanes solve the window's selection (select_window_coop, shared with the main path); services of a batch run
// side by side.  Skip spans appear in candidate tuples as idx = -(kSkipBase + window * kSkipStride + position in the
// pool): two candidates holding the same skip span at an endpoint conflict like any shared span
// (traceweaver_v3.py:1276-1281); skip spans are never consumed (traceweaver_v1.py:460-462).
#pragma once
#include "tw_kernels.h"

namespace tw {

constexpr int kSkipBase = 1024;
constexpr int kSkipStride = 128;
constexpr int kSkipCand = 64;    // spans inside one request per endpoint (more: TW_ERR_WINDOW_WIDTH)

struct SkipUnitDev {
    const int32_t* perm;      // per endpoint segment (ep_off): list position of the k-th span in start order (stable)
    const int64_t* tw_start;  // [n_tw] time windows in start order
    const int32_t* pool;      // [E][n_tw] skip spans per (endpoint, time window)
    const double* dist;       // [(E+1)][(E+1)][2] mean, std (NaN: no such pair); index 0 = the incoming endpoint
    long long* fetches;       // [E][n_tw] draws so far
    int32_t n_tw;
};

__device__ __forceinline__ bool is_skip_idx(int32_t x) { return x <= -kSkipBase; }

// BuildDistributions (traceweaver_v3.py:108-172): spans of the service merged in start order (stable: incoming spans,
// then the endpoints in order); every span looks back -- no further than the longest request -- for its nearest
// qualifying predecessor and contributes one delay sample to the pair (predecessor's endpoint, own endpoint).
// ep: 0 = incoming (server spans), 1 + e = outgoing endpoint e (client spans).  key_out = a * (E+1) + b or -1.
__global__ void k_build_distributions(const int64_t* start, const int64_t* dur, const uint8_t* ep, int64_t n, int64_t large_delay, int E,
                                      int32_t* key_out, int64_t* sample_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t st = start[i], du = dur[i];
    const int me = ep[i];
    int key = -1;
    int64_t sample = 0;
    for (int64_t j = i - 1; j >= 0; j--) {
        const int64_t ps = start[j];
        if ((st + du) - ps > large_delay) break;
        const int pe = ep[j];
        if (me != 0) {            // client span: the nearest server span, or a client span of an earlier endpoint that ended before
            if (pe == 0) { key = pe * (E + 1) + me; sample = st - ps; break; }
            if (ps + dur[j] < st && pe < me) { key = pe * (E + 1) + me; sample = st - (ps + dur[j]); break; }
        } else if (pe != 0 && ps + dur[j] < st + du) {   // server span: the nearest client span that ended before it does
            key = pe * (E + 1) + me; sample = (st + du) - (ps + dur[j]); break;
        }
    }
    key_out[i] = key;
    sample_out[i] = sample;
}

// Gaussian parameters that make every term finite (mean 0, std 1): the first enumeration of a skip-mode batch only
// serves the candidate sets of the windows.
__global__ void k_neutral_params(double* gparam, int64_t total) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    gparam[g * 4 + 0] = 0.0; gparam[g * 4 + 1] = 1.0; gparam[g * 4 + 2] = 0.0; gparam[g * 4 + 3] = 1.0;
}

struct SkipLds {
    int32_t cand[2][kMaxEp][kSkipCand];   // [enumeration][endpoint]: candidate spans of the request, in enumeration order
    int32_t ncand[2][kMaxEp];
    Cand<kMaxEp> heap[kTopK + 1];
    int32_t x[kMaxEp];                    // tuple under construction
    int32_t pos[kMaxEp];                  // next candidate position per level (ncand = the sentinel, ncand + 1 = done)
    int nheap, err, window_first;
};

struct SkipCtx {
    const Dev& P; const UnitDev& U; const SkipUnitDev& K; SkipLds& S;
    int64_t in_start, in_end;
    int wi;
    __device__ int64_t ostart(int e, int32_t x) const { return P.out_start[U.ep_off[e] + x]; }
    __device__ int64_t oend(int e, int32_t x) const { return P.out_end[U.ep_off[e] + x]; }
    // GetEpPairCost(normalized=True), traceweaver_v1.py:117-139: scipy.stats.norm.pdf
    __device__ double term(int a, int b, int64_t t1, int64_t t2) {
        const int E1 = U.E + 1;
        const double mean = K.dist[((int64_t)a * E1 + b) * 2];
        double sd = K.dist[((int64_t)a * E1 + b) * 2 + 1];
        if (mean != mean) { S.err = TW_ERR_SKIP_PARAMS; return 0.0; }
        if (sd < 1.0e-12) sd = 0.001;
        const double y = ((double)(t2 - t1) - mean) / sd;
        return (tw_exp(-(y * y) / 2.0) / kSqrt2Pi) / sd;
    }
    // ScoreAssignmentAsPerInvocationGraph, traceweaver_v1.py:259-361, normalized
    __device__ double score() {
        const int E = U.E;
        int last = -1, nmap = 0;
        int64_t last_end = 0;
        for (int e = 0; e < E; e++) {   // max() over the non-skipped spans keeps the first maximum
            if (is_skip_idx(S.x[e])) continue;
            const int64_t en = oend(e, S.x[e]);
            if (last < 0 || en > last_end) { last_end = en; last = e; }
        }
        if (last < 0) { S.err = TW_ERR_SKIP_REFERENCE_RAISES; return 0.0; }   // every endpoint skipped: hazard H7
        double cost = 0.0;
        for (int e = 0; e < E; e++) {
            if (is_skip_idx(S.x[e])) continue;
            const int64_t st = ostart(e, S.x[e]);
            for (int j = 0; j < U.npred[e]; j++) {
                if (!U.pred_prim[e][j]) continue;
                const int b = U.pred_list[e][j];
                if (is_skip_idx(S.x[b])) {
                    if (U.npred[b] == 0) { cost += term(0, 1 + e, in_start, st); nmap++; continue; }   // FindValidAncestor -> None
                    int lat = -1;
                    int64_t lat_end = 0;
                    for (int q = 0; q < U.npred[b]; q++) {
                        const int a = U.pred_list[b][q];
                        if (is_skip_idx(S.x[a])) continue;
                        const int64_t en = oend(a, S.x[a]);
                        if (lat < 0 || en > lat_end) { lat_end = en; lat = a; }
                    }
                    if (lat < 0) { S.err = TW_ERR_SKIP_REFERENCE_RAISES; return 0.0; }   // max() of a nested list, traceweaver_v1.py:338
                    cost += term(1 + lat, 1 + e, ostart(lat, S.x[lat]), st); nmap++;     // latest[1].start_mus, traceweaver_v1.py:340
                    continue;
                }
                cost += term(1 + b, 1 + e, oend(b, S.x[b]), st); nmap++;
            }
            if (U.npred[e] == 0) { cost += term(0, 1 + e, in_start, st); nmap++; }
            if (e == last) { cost += term(1 + e, 0, oend(e, S.x[e]), in_end); nmap++; }
        }
        return cost / (double)nmap;
    }
    // (score, [spans]) order; two different skip spans compare "None" < "None" (neither less); a skip span against a real
    // one raises in the reference (str < int)
    __device__ bool lt(const Cand<kMaxEp>& a, const Cand<kMaxEp>& b) {
        if (a.score != b.score) return a.score < b.score;
        for (int e = 0; e < U.E; e++)
            if (a.idx[e] != b.idx[e]) {
                const bool sa = is_skip_idx(a.idx[e]), sb = is_skip_idx(b.idx[e]);
                if (sa && sb) return false;
                if (sa || sb) { S.err = TW_ERR_SKIP_REFERENCE_RAISES; return false; }
                return ostart(e, a.idx[e]) < ostart(e, b.idx[e]);
            }
        return false;
    }
    __device__ void siftdown(int startpos, int pos) {
        const Cand<kMaxEp> item = S.heap[pos];
        while (pos > startpos) {
            const int parent = (pos - 1) >> 1;
            if (lt(item, S.heap[parent])) { S.heap[pos] = S.heap[parent]; pos = parent; continue; }
            break;
        }
        S.heap[pos] = item;
    }
    __device__ void siftup(int pos) {
        const int startpos = pos;
        const Cand<kMaxEp> item = S.heap[pos];
        int child = 2 * pos + 1;
        while (child < S.nheap) {
            const int right = child + 1;
            if (right < S.nheap && !lt(S.heap[child], S.heap[right])) child = right;
            S.heap[pos] = S.heap[child];
            pos = child;
            child = 2 * pos + 1;
        }
        S.heap[pos] = item;
        siftdown(startpos, pos);
    }
    __device__ void push(const Cand<kMaxEp>& c) {
        S.heap[S.nheap++] = c;
        siftdown(0, S.nheap - 1);
        if (S.nheap > kTopK) {
            const Cand<kMaxEp> last = S.heap[--S.nheap];
            if (S.nheap > 0) { S.heap[0] = last; siftup(0); }
        }
    }
    __device__ void reverse(int n) {
        for (int i = 0, j = n - 1; i < j; i++, j--) { const Cand<kMaxEp> t = S.heap[i]; S.heap[i] = S.heap[j]; S.heap[j] = t; }
    }
    __device__ void sort_desc() {   // CPython list.sort(reverse=True), n < 64
        const int n = S.nheap;
        if (n < 2) return;
        reverse(n);
        int run = 2;
        if (lt(S.heap[1], S.heap[0])) {
            for (int i = 2; i < n; i++, run++) if (!lt(S.heap[i], S.heap[i - 1])) break;
            reverse(run);
        } else {
            for (int i = 2; i < n; i++, run++) if (lt(S.heap[i], S.heap[i - 1])) break;
        }
        for (int start = run; start < n; start++) {
            int l = 0, r = start;
            const Cand<kMaxEp> pivot = S.heap[start];
            do {
                const int p = l + ((r - l) >> 1);
                if (lt(pivot, S.heap[p])) r = p; else l = p + 1;
            } while (l < r);
            for (int p = start; p > l; p--) S.heap[p] = S.heap[p - 1];
            S.heap[l] = pivot;
        }
        reverse(n);
    }
    // The spans of endpoint e inside the request, found through the start-ordered view of the list (the reference scans
    // the whole list, traceweaver_v3.py:316-320 with dynamism; containment selects these).  which = 0: the spans that are
    // still there, in list order (top_k, traceweaver_v3.py:1182); which = 1: all spans, in start order (top_k_2, :1185,
    // on the lists TallySkipSpans has sorted, :968-971).
    __device__ void gather(int which, int e) {
        const int n = (int)(U.ep_off[e + 1] - U.ep_off[e]);
        const int32_t* perm = K.perm + U.ep_off[e];
        int lo = 0, hi = n;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (ostart(e, perm[mid]) < in_start) lo = mid + 1; else hi = mid; }
        int c = 0;
        for (int p = lo; p < n; p++) {
            const int32_t x = perm[p];
            if (ostart(e, x) > in_end) break;
            if (oend(e, x) > in_end) continue;
            if (which == 0 && P.owner[U.ep_off[e] + x] == 0) continue;   // taken by an earlier window
            if (c >= kSkipCand) { S.err = TW_ERR_WINDOW_WIDTH; break; }
            S.cand[which][e][c++] = x;
        }
        if (which == 0)   // list order = position order
            for (int a = 1; a < c; a++) {
                const int32_t v = S.cand[0][e][a];
                int b = a - 1;
                while (b >= 0 && S.cand[0][e][b] > v) { S.cand[0][e][b + 1] = S.cand[0][e][b]; b--; }
                S.cand[0][e][b + 1] = v;
            }
        S.ncand[which][e] = c;
    }
    // DfsTraverseX (traceweaver_v3.py:292-351) with the sentinel branch; returns the number of tuples
    __device__ int64_t enumerate(int which) {
        const int E = U.E;
        int64_t leaves = 0;
        S.nheap = 0;
        int d = 0;
        S.pos[0] = 0;
        while (d >= 0 && S.err == 0) {
            if (d == E) {
                leaves++;
                Cand<kMaxEp> c;
                c.score = score();
                for (int q = 0; q < kMaxEp; q++) c.idx[q] = q < E ? S.x[q] : -1;
                push(c);
                d--;
                continue;
            }
            const int nc = S.ncand[which][d];
            bool found = false;
            while (S.pos[d] <= nc) {
                const int p = S.pos[d]++;
                if (p < nc) {
                    const int32_t x = S.cand[which][d][p];
                    const int64_t st = ostart(d, x);
                    bool ok = true;
                    for (int q = 0; q < d && ok; q++)
                        if (((U.pred_mask[d] >> q) & 1) && !is_skip_idx(S.x[q]) && oend(q, S.x[q]) > st) ok = false;   // a skipped predecessor orders nothing
                    if (!ok) continue;
                    S.x[d] = x;
                    found = true;
                    break;
                }
                // the sentinel: FetchSkipFromWindow, least used first == round robin over the window's pool
                const int32_t pool = K.pool[(int64_t)d * K.n_tw + wi];
                if (pool > 0) {
                    long long* f = &K.fetches[(int64_t)d * K.n_tw + wi];
                    const int at = (int)(*f % pool);
                    (*f)++;
                    S.x[d] = -(kSkipBase + wi * kSkipStride + at);
                    found = true;
                    break;
                }
            }
            if (!found) { d--; continue; }
            d++;
            if (d < E) S.pos[d] = 0;
        }
        if (S.err == 0) sort_desc();
        return leaves;
    }
};

// One wavefront per skip-mode unit (see the head of this file).  Windows (win_end, wid, w_last) exist already.
__global__ void __launch_bounds__(64) k_skip_walk(Dev P, const SkipUnitDev* skip) {
    if (*P.err != 0) return;
    __shared__ SelectLds L;
    __shared__ SkipLds S;
    const int unit = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
    const UnitDev& U = P.units[unit];
    const SkipUnitDev& K = skip[unit];
    const int E = U.E;
    TW_SEL_DECL();
    for (int q = t; q < SelectLds::kSlots; q += nt) L.memo[q].state = 0u;
    if (t == 0) { L.memo_gen = 0u; S.err = 0; S.window_first = 0; }
    group_sync();
    for (int i = 0; i < U.n_in; i++) {
        const int64_t g = U.in_off + i;
        if (t == 0) {
            SkipCtx C{P, U, K, S, P.in_start[g], P.in_end[g], -1};
            {   // FindWindow (traceweaver_v3.py:827-832): the window with the largest start <= the request's start, the first of equal starts
                int64_t best = INT64_MIN;
                for (int w = 0; w < K.n_tw; w++) if (K.tw_start[w] <= C.in_start && K.tw_start[w] > best) { best = K.tw_start[w]; C.wi = w; }
                if (C.wi < 0) S.err = TW_ERR_ARG;
            }
            for (int which = 0; which < 2 && S.err == 0; which++) {
                for (int e = 0; e < E; e++) C.gather(which, e);
                if (S.err != 0) break;
                const int64_t leaves = C.enumerate(which);
                if (S.err != 0) break;
                int32_t* out_n = which == 0 ? P.tkr_n : P.tk_n;
                int32_t* out_idx = which == 0 ? P.tkr_idx : P.tk_idx;
                double* out_score = which == 0 ? P.tkr_score : P.tk_score;
                out_n[g] = S.nheap;
                for (int k = 0; k < kTopK; k++) {
                    out_score[tks_index(U, k, i)] = k < S.nheap ? S.heap[k].score : dnan();
                    for (int e = 0; e < E; e++) out_idx[tk_index(U, k, e, i)] = k < S.nheap ? S.heap[k].idx[e] : -1;
                }
                if (which == 0) { P.leaves[g] = leaves; P.leaves_r[g] = leaves; P.rep[g] = 1; }   // per_span_candidates counts the top_k call only
            }
        }
        __threadfence();
        group_sync();
        if (S.err != 0) break;   // uniform: S.err is in LDS and stable here
        if (P.win_end[g]) {
            const int first = S.window_first, m = i - first + 1;
            if (m > kMaxWin) { if (t == 0) S.err = TW_ERR_WINDOW_SIZE; group_sync(); break; }
            select_window_coop(P, U, unit, first, m, L TW_SEL_PASS);
            __threadfence();
            group_sync();
            for (int k = t; k < m * E; k += nt) {   // AddAssignment(delete_out_spans=True): chosen real spans leave the lists
                const int b = k / E, e = k % E, c = P.chosen[U.in_off + first + b];
                if (c < 0) continue;
                const int32_t x = cand_idx(P, U, first + b, c, e);
                if (!is_skip_idx(x)) P.owner[U.ep_off[e] + x] = 0;
            }
            if (t == 0) S.window_first = i + 1;
            __threadfence();
            group_sync();
        }
    }
    if (t == 0 && S.err != 0) raise_err(P, S.err);
}

}  // namespace tw
