// tw_device.h -- device-side data structures and arithmetic of the span->parent assignment engine.
//
// Arithmetic contract (DESIGN.md "Scores"): every score is a chain of IEEE-754 binary64 operations
// in the order the reference evaluates them (traceweaver_v1.py:117-139,259-361), with no FMA
// contraction (the translation unit is built with -ffp-contract=off).  log / exp / log1p follow the
// published fdlibm algorithms (e_log.c, e_exp.c, s_log1p.c) so that the result is a pure function of
// the inputs on any IEEE machine; they agree with glibc/numpy to <= 1 ulp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "traceweaver_amd.h"

namespace tw {

constexpr int kMaxEp = TW_MAX_EP;
constexpr int kTopK = TW_TOPK;
constexpr int kMaxComp = TW_MAX_COMP;
constexpr int kMaxWin = TW_MAX_WINDOW;
constexpr int kCandWords = TW_CAND_WORDS;
constexpr int kEnumStretches = 8;   // most stretches a class' tile kernel is launched in (launch_enumerate: the wavefront kernel of a stretch runs beside the next stretch's tiles)
constexpr int kTile = 128;  // incoming spans per workgroup in the per-span kernels (measured: 128 beats 64 and 256)
constexpr int kCoop = 256;  // threads of the per-unit / per-row cooperative kernels
constexpr int32_t kNoOwner = 0x7f7f7f7f;

// One service unit as the kernels see it.
struct UnitDev {
    int64_t in_off;            // first incoming span in in_start/in_end
    int64_t ie_off;            // sum of n_in*E of earlier units: base of per-(span,endpoint) arrays
    int64_t ep_off[kMaxEp + 1];  // global offsets of the endpoint segments in out_start/out_end
    int32_t n_in;
    int32_t E;
    int32_t nblk;              // ceil(n_in / batch_size)
    int32_t nslot;             // E*E + 2E
    int64_t gp_off;            // base of this unit's Gaussian parameter table (in slots)
    int32_t slot_off;          // base of this unit's mixture table (in slots)
    int32_t tile_off;          // first tile (workgroup) of this unit
    int32_t ntile;
    uint8_t pred_mask[kMaxEp];            // bit p set <=> edge p->e (ordering constraint)
    uint8_t succ_mask[kMaxEp];            // bit f set <=> edge e->f
    uint8_t npred[kMaxEp];
    uint8_t pred_list[kMaxEp][kMaxEp];    // predecessors in networkx in_edges() order
    uint8_t pred_prim[kMaxEp][kMaxEp];    // 1 <=> that in-edge is primary (scored)
    uint8_t key_rank[kMaxEp];             // rank of the endpoint in the partition-key order (tw_batch.key_rank)
    double tscale;             // microseconds per timestamp unit (tw_batch.unit_time_scale; 1.0 for integer microseconds)
    int32_t float_time;        // timestamps are images of binary64 values: sums of timestamps accumulate in binary64
    int32_t part;              // tw_batch.unit_part: bit 0 the first request follows a cut, bit 1 a cut follows the last request
    int32_t skip;              // skip-mode unit: lists as handed over (not necessarily sorted), Python's bisect for the cut-offs
};

struct TileDev {
    int32_t unit;
    int32_t first;  // local index of the tile's first incoming span
};

struct PairVI {
    int64_t v;
    int32_t i;
};

// The tiles a per-span kernel of the window / selection stage is launched over: every tile of the batch (ids == nullptr: workgroup b
// serves tile b; the repair rounds, skip mode) or the tiles of one endpoint-count class (workgroup b serves tile ids[b]) -- units of
// different classes are independent, so a class' windows and selection run on the class' stream as soon as ITS enumeration is
// complete, beside the longer enumerations of the other classes.  `base` = tiles before the set in the class order (the set's share
// of the selection work lists starts at base * tile_spans), `slot` = its block of list counters and cursors (0 = all tiles, E = class E).
struct TileSet {
    const int32_t* ids;
    int32_t base, n, slot;
};

// All device pointers, passed to kernels by value.
struct Dev {
    const UnitDev* units;
    const TileDev* tiles;
    int32_t n_units, n_tiles;
    int32_t tile_spans;        // incoming spans per tile (workgroup size of the per-span kernels)
    int64_t n_in_total, n_out_total;
    int32_t batch_size, batch_mis;
    // spans (SoA, 16 B per span)
    const int64_t *in_start, *in_end, *out_start, *out_end;
    // sorted end times for the rank-aligned block statistics
    int64_t *in_end_sorted, *out_end_sorted;
    // Gaussian parameters [gp slots][4] = mean, std, log(std_used), std_used
    double* gparam;
    // mixtures [slots] / [slots][kMaxComp][4] = mean, prec_chol, log(prec_chol), log(weight)
    const int32_t* mix_n;
    const double* mix_c;
    // per incoming span
    int64_t* pm_val;    // inclusive prefix max of in_end inside the unit
    int32_t* pm_idx;    // its arg max (latest on ties)
    uint8_t* pc;        // PerfectCut(i)
    int32_t* seg;       // latest PerfectCut position <= i (0 if none)
    uint8_t* win_end;   // window closes after this span
    int32_t* wid;       // window id inside the unit
    int32_t* w_last;    // [in_off + w] last span of window w
    int32_t* unit_nwin; // [n_units]
    uint8_t* w_dirty;   // [in_off + w] window was re-solved because an earlier window took one of its candidate spans
    int32_t* w_conf;    // [in_off + w] the spans' best candidates clash: the window is on a selection work list
    int32_t* unit_ndirty;
    int32_t* tk_n;      // candidates found on all spans (top_k_2)
    int64_t* leaves;
    int64_t* leaves0;   // tuples of the first solve (on all spans) as counted in pass 1: the same in pass 2, whose pruned walks do not recount them
    int32_t* chosen;
    uint8_t* rep;       // 1 <=> candidate list was recomputed with consumed spans masked out
    int32_t* tkr_n;
    // per (unit, k, endpoint, span) / (unit, k, span)
    int32_t *tk_idx, *tkr_idx;
    double *tk_score, *tkr_score;
    // per (unit, endpoint, span)
    int32_t* c_lo;      // first candidate index (full-list cutoff)
    int32_t* c_hi;      // last candidate index
    uint64_t* c_bits;   // kCandWords words: spans that occur in >= 1 feasible tuple
    uint64_t* gone;     // kCandWords words: candidate spans taken by earlier windows when the span's current list was computed
    uint8_t* gone_valid;   // per incoming span: its words of `gone` have been written in this pass (else they read as zero)
    int64_t* leaves_r;  // per incoming span: tuples of the enumeration on the remaining spans (rep = 1)
    unsigned long long* frontier_big;  // pool of kFrontierBigSlots longer lists for the spans that outgrow their wavefront's buffers
    int32_t* frontier_big_busy;   // [frontier_big_slots] 1 <=> the list is in use (pool_acquire / pool_release)
    int32_t frontier_big_slots;
    int32_t* frontier_busy;       // [kFrontierSlots] ... of the wavefronts' own buffer pairs
    unsigned long long* frontier;  // scratch of k_enumerate_heavy: per wavefront two lists of kFrontierCap feasible prefixes
    int32_t* round_changed;  // spans whose set of taken candidate spans changed in the current repair round
    int32_t* parent;
    // per outgoing span
    int32_t* owner;     // smallest window id that speculatively chose the span
    // gap samples [gs_off + q*n_in + i]
    double* gaps;
    const int64_t* gs_off;  // [n_units]
    int64_t* unit_stats;    // [n_units][8]
    // Windows whose best candidates clash, listed by k_select_fast / k_detect_gone: four lists (0 windows of <= kBruteMax spans for
    // k_select_tiny; 1 long, 2 middle and 3 very long ones for the three instantiations of k_select_heavy), each cut into kSelSeg
    // segments by tile range with a counter of its own on a cache line of its own (same-address atomics serialise: one counter per
    // list kept k_select_fast at 1.4 ms for 150 k wavefronts).  heavy_count[((slot * 4 + list) * kSelSeg + segment) * kCtrStride] (slot: TileSet);
    // segment s owns the positions (set base + [first tile of s, first tile of s + 1)) * tile_spans of tiny_* (short windows from the front of the segment, very
    // long ones from its back) and of heavy_* (long ones from the front, middle ones from the back).
    int32_t* heavy_count;
    int32_t* heavy_next;    // [tile set slot][8] next unclaimed item of every selection list of the set; [0][4], [0][5]: see hard_unit
    int32_t *heavy_unit, *heavy_win;
    int32_t *tiny_unit, *tiny_win;
    int32_t *rheavy_unit, *rheavy_win, *rtiny_unit, *rtiny_win;   // ... of the repair rounds (slot 0: list_window)
    int32_t *hard_unit, *hard_win;   // windows k_select_heavy gave up on, for k_select_dp (count and cursor: heavy_next[4], [5])
    int32_t* heavy_in_count;  // [2][kMaxEp+1] incoming spans deferred to k_enumerate_heavy per endpoint count E: narrow, wide windows
    int32_t* heavy_in_next;   // [4][2][kMaxEp+1] work-list cursors of the wavefront kernels: per kind of launch (the class' lists, list parts, spans listed again, fallback), instantiation (narrow, wide) and class (enum_cursor)
    int32_t heavy_in_off[kMaxEp + 2];  // class offsets into heavy_in_unit / heavy_in_idx
    int32_t* enum_snap;         // [kMaxEp+1][kEnumStretches+1][4] list counters of class E (long, narrow, wide) after the j-th stretch of its tile kernel (k_enum_snapshot)
    int32_t* enum_stretch_next; // [kMaxEp+1][kEnumStretches][2] work-list cursor of the wavefront kernel's launch for a stretch (narrow, wide)
    int32_t *heavy_in_unit, *heavy_in_idx;
    int32_t* heavy_big_count;   // [kMaxEp+1] narrow spans with a long enumeration: served first
    int32_t *heavy_big_unit, *heavy_big_idx;   // class offsets heavy_big_off (room for the extra entries of split spans)
    int32_t heavy_big_off[kMaxEp + 2];
    // A very long enumeration is cut into parts by the position of the first endpoint's candidate (k_enumerate_tile lists one
    // entry per part); the parts leave their top-5 in a scratch slot each and k_merge_parts combines them.
    int32_t* heavy_big_part;   // [big list] number of parts | part << 8 | wide windows << 24   (1 = the whole enumeration)
    int32_t* heavy_big_slot;   // [big list] scratch slot of the part's result
    int32_t part_off[kMaxEp + 2];   // class offsets into the part scratch and the split-span records (2 slots per extra entry)
    int32_t split_twins;       // spans with twin candidates (two candidates of one endpoint that start together: Python's order of tuples may not
                               // decide): 2 (default) split, every part replays CPython's heap on its share and logs what entered it, k_merge_parts
                               // replays the logs; 1 split like the others, the span enumerated again as a whole when the order is not decided
                               // (tests: the merge's way back); 0 not split (TW_SPLIT_TWINS)
    int32_t* part_used;        // [kMaxEp+1] extra list entries handed out per class
    // deferred spans (deep call graphs; kListSplitFlag in tw_kernels.h): their listed prefixes, cut into list parts
    int32_t defer_min_e;       // classes from this many endpoints on defer their long spans (TW_DEFER_MIN_E, default 5; 9 = none)
    int32_t defer_cap;         // entries of defer_list
    unsigned long long* defer_list;   // arena of prefix lists (bump-allocated per pass)
    int32_t* defer_used;       // [1] entries handed out
    int32_t* defer_count;      // [kMaxEp+1] list parts appended behind the class' listed entries
    int32_t *part_lo, *part_hi;   // [slot] a list part's stretch of defer_list
    int32_t* part_lvl;            // [slot] ... whose entries are prefixes of the endpoints 0 .. part_lvl
    int32_t* any_wide;         // [kMaxEp+1] 1 <=> the class listed a span with wide windows in the first enumeration of this pass (the same spans in pass 2: the host
                               // launches the wide instantiations of pass 2 only for such classes)
    int32_t* redo_count;       // [kMaxEp+1] spans k_merge_parts lists again (their parts left the order of equal scores undecided): entries [0, redo_count) of the class' list of long enumerations
    int32_t* defer_refused;    // [kMaxEp+1] spans that asked for list parts / parts and were refused (budget of extra entries or arena exhausted): enumerated by one wavefront
    // what k_enumerate_lean leaves to k_enumerate_heavy (part = 4): entries in the format of the list of long enumerations, class offsets heavy_big_off
    int32_t lean_min_e;        // classes from this many endpoints on are enumerated by k_enumerate_lean (TW_LEAN_MIN_E, default 5; 9 = none)
    int32_t tile_max_deep;     // ... and take over from the tile kernel from this many tuples on (TW_TILE_MAX_DEEP <= kTileMax)
    int32_t lean_grid;         // ... which walks enumerations of up to this many grid points as a grid and lists the others (TW_LEAN_GRID)
    // pair-term tables that do not fit the wavefront's (small) LDS pool: slots of kPairSpill doubles in global memory, claimed by the
    // wavefronts that need one (pool_acquire), kept until the wavefront ends
    double* pair_pool;
    int32_t* pair_busy;
    int32_t* fb_count;         // [kMaxEp+1] entries of the current launch chain (reset with the work lists of a repair round)
    int32_t* fb_total;         // [kMaxEp+1] ... of the whole pass (tw_debug_worklists)
    int32_t *fb_unit, *fb_idx, *fb_part, *fb_slot;
    int32_t* split_count;      // [kMaxEp+1] split spans per class
    int32_t *split_unit, *split_idx, *split_slot, *split_parts;   // [part_off region] one record per split span
    int32_t* part_n;           // [slot] entries | ambiguous << 8
    int64_t* part_leaves;      // [slot]
    double* part_score;        // [slot][kTopK]
    int32_t* part_idx;         // [slot][kTopK][kMaxEp] span indices in the endpoint lists
    unsigned long long* part_bits;   // [slot][kMaxEp][kCandWords] candidate spans seen in a feasible tuple
    // log mode (split spans with twin candidates): the tuples a part pushed on its own heap while their score was not below the
    // heap's root -- a superset of the part's tuples that enter the heap of the whole enumeration -- in enumeration order
    int32_t* part_logn;        // [slot] entries logged (more than kPartLogCap: the log is incomplete)
    double* part_log_sc;       // [slot][kPartLogCap] score
    unsigned long long* part_log_ix;   // [slot][kPartLogCap] positions in the cut-off windows, 8 bits per endpoint
    int32_t* err;           // first error raised by a kernel (tw_status)
    unsigned long long* prof;  // [16] phase timers of -DTW_PROFILE builds
};

// ---------------------------------------------------------------------------------------------
// fdlibm-style elementary functions, strict IEEE double
__device__ __forceinline__ int32_t hi_word(double x) { return (int32_t)(__double_as_longlong(x) >> 32); }
__device__ __forceinline__ uint32_t lo_word(double x) { return (uint32_t)__double_as_longlong(x); }
__device__ __forceinline__ double with_hi(double x, int32_t hi) {
    return __longlong_as_double((long long)(((uint64_t)(uint32_t)hi << 32) | (uint64_t)lo_word(x)));
}
__device__ __forceinline__ double dnan() { return __longlong_as_double(0x7ff8000000000000LL); }
__device__ __forceinline__ double dinf() { return __longlong_as_double(0x7ff0000000000000LL); }

__device__ inline double tw_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 two54 = 1.80143985094819840000e+16, Lg1 = 6.666666666666735130e-01,
                 Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01,
                 Lg6 = 1.531383769920937332e-01, Lg7 = 1.479819860511658591e-01;
    int32_t k = 0, hx = hi_word(x);
    if (hx < 0x00100000) {
        if (((hx & 0x7fffffff) | lo_word(x)) == 0) return -dinf();
        if (hx < 0) return dnan();
        k -= 54;
        x *= two54;
        hx = hi_word(x);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int32_t i = (hx + 0x95f64) & 0x100000;
    x = with_hi(x, hx | (i ^ 0x3ff00000));
    k += (i >> 20);
    double f = x - 1.0, dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) {
        if (f == 0.0) return k == 0 ? 0.0 : dk * ln2_hi + dk * ln2_lo;
        double R = f * f * (0.5 - 0.33333333333333333 * f);
        return k == 0 ? f - R : dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    double s = f / (2.0 + f), z = s * s, w = z * z;
    i = hx - 0x6147a;
    int32_t j = 0x6b851 - hx;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    if ((i | j) > 0) {
        double hfsq = 0.5 * f * f;
        return k == 0 ? f - (hfsq - s * (hfsq + R)) : dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    return k == 0 ? f - s * (f - R) : dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

__device__ inline double tw_exp(double x) {
    const double o_threshold = 7.09782712893383973096e+02, u_threshold = -7.45133219101941108420e+02,
                 ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                 P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08,
                 twom1000 = 9.33263618503218878990e-302;
    double hi = 0.0, lo = 0.0;
    int32_t k = 0, hx = hi_word(x);
    const int32_t xsb = (hx >> 31) & 1;
    hx &= 0x7fffffff;
    if (hx >= 0x40862E42) {
        if (hx >= 0x7ff00000) {
            if (((hx & 0xfffff) | lo_word(x)) != 0) return x + x;
            return xsb == 0 ? x : 0.0;
        }
        if (x > o_threshold) return dinf();
        if (x < u_threshold) return 0.0;
    }
    if (hx > 0x3fd62e42) {
        if (hx < 0x3FF0A2B2) {
            if (xsb == 0) { hi = x - ln2HI; lo = ln2LO; k = 1; }
            else { hi = x + ln2HI; lo = -ln2LO; k = -1; }
        } else {
            k = (int32_t)(invln2 * x + (xsb == 0 ? 0.5 : -0.5));
            const double t = (double)k;
            hi = x - t * ln2HI;
            lo = t * ln2LO;
        }
        x = hi - lo;
    } else if (hx < 0x3e300000) {
        return 1.0 + x;
    }
    const double t = x * x;
    const double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    const double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    if (k >= -1021) return with_hi(y, hi_word(y) + (k << 20));
    return with_hi(y, hi_word(y) + ((k + 1000) << 20)) * twom1000;
}

// exp(x) for x <= 0 (or NaN): the arithmetic of tw_exp with its case distinctions written as selects, so that a wavefront whose
// lanes fall into different ranges runs one instruction stream (in the mixture terms of pass 2 every lane evaluates exp of another
// argument; the branches of tw_exp then cost the sum of their paths).  Bit-identical to tw_exp on that domain: the reduction
// k = (int)(invln2 * x - 0.5), hi = x - k * ln2HI, lo = k * ln2LO is what the k = -1 shortcut of e_exp.c computes too (-1 * ln2HI
// and -1 * ln2LO are exact, and invln2 * x - 0.5 truncates to -1 on all of its range), and with k = 0 it leaves x unchanged.
__device__ __forceinline__ double tw_exp_nonpos(double x) {
    const double u_threshold = -7.45133219101941108420e+02,
                 ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                 P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08,
                 twom1000 = 9.33263618503218878990e-302;
    const bool special = !(x >= u_threshold);   // below the underflow threshold, -inf or NaN
    const double xs = special ? 0.0 : x;
    const int32_t hx = hi_word(xs) & 0x7fffffff;
    const int32_t k = hx > 0x3fd62e42 ? (int32_t)(invln2 * xs - 0.5) : 0;
    const double t = (double)k;
    const double hi = xs - t * ln2HI, lo = t * ln2LO;
    const double xr = hi - lo;
    const double t2 = xr * xr;
    const double c = xr - t2 * (P1 + t2 * (P2 + t2 * (P3 + t2 * (P4 + t2 * P5))));
    const double q = (xr * c) / (k == 0 ? c - 2.0 : 2.0 - c);
    const double r0 = 1.0 - (q - xr);
    const double y = 1.0 - ((lo - q) - hi);
    const double ys = with_hi(y, hi_word(y) + ((k >= -1021 ? k : k + 1000) << 20));
    double r = k == 0 ? r0 : (k >= -1021 ? ys : ys * twom1000);
    r = hx < 0x3e300000 ? 1.0 + xs : r;
    return special ? (x != x ? x + x : 0.0) : r;
}

__device__ inline double tw_log1p(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 two54 = 1.80143985094819840000e+16, Lp1 = 6.666666666666735130e-01,
                 Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
                 Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01,
                 Lp6 = 1.531383769920937332e-01, Lp7 = 1.479819860511658591e-01;
    double f = 0.0, c = 0.0, u;
    int32_t k = 1, hx = hi_word(x), hu = 0;
    const int32_t ax = hx & 0x7fffffff;
    if (hx < 0x3FDA827A) {
        if (ax >= 0x3ff00000) return x == -1.0 ? -dinf() : dnan();
        if (ax < 0x3e200000) {
            if (two54 + x > 0.0 && ax < 0x3c900000) return x;
            return x - x * x * 0.5;
        }
        if (hx > 0 || hx <= ((int32_t)0xbfd2bec3)) { k = 0; f = x; hu = 1; }
    }
    if (hx >= 0x7ff00000) return x + x;
    if (k != 0) {
        if (hx < 0x43400000) {
            u = 1.0 + x;
            hu = hi_word(u);
            k = (hu >> 20) - 1023;
            c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);
            c /= u;
        } else {
            u = x;
            hu = hi_word(u);
            k = (hu >> 20) - 1023;
            c = 0;
        }
        hu &= 0x000fffff;
        if (hu < 0x6a09e) u = with_hi(u, hu | 0x3ff00000);
        else { k += 1; u = with_hi(u, hu | 0x3fe00000); hu = (0x00100000 - hu) >> 2; }
        f = u - 1.0;
    }
    const double hfsq = 0.5 * f * f;
    if (hu == 0) {
        if (f == 0.0) {
            if (k == 0) return 0.0;
            c += k * ln2_lo;
            return k * ln2_hi + c;
        }
        const double R = hfsq * (1.0 - 0.66666666666666666 * f);
        return k == 0 ? f - R : k * ln2_hi - ((R - (k * ln2_lo + c)) - f);
    }
    const double s = f / (2.0 + f), z = s * s;
    const double R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
    return k == 0 ? f - (hfsq - s * (hfsq + R)) : k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + c))) - f);
}

// numpy's add.reduce over a short contiguous double array: 0 + pairwise_sum (n <= 128 branch)
__device__ inline double np_sum(const double* a, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r[8];
    int i;
    for (int j = 0; j < 8; j++) r[j] = a[j];
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return 0.0 + res;
}

constexpr double kLogSqrt2Pi = 0x1.d67f1c864beb4p-1;  // np.log(np.sqrt(2*np.pi))
constexpr double kLog2Pi = 0x1.d67f1c864beb4p+0;      // np.log(2*np.pi)
constexpr double kSqrt2Pi = 0x1.40d931ff62706p+1;     // np.sqrt(2*np.pi)

__host__ __device__ inline int slot_root(int E, int e) { return e; }
__host__ __device__ inline int slot_prim(int E, int p, int e) { return E + p * E + e; }
__host__ __device__ inline int slot_close(int E, int e) { return E + E * E + e; }

// index helpers for the per-unit result arrays
__device__ __forceinline__ int64_t ie_index(const UnitDev& U, int e, int i) { return U.ie_off + (int64_t)e * U.n_in + i; }
__device__ __forceinline__ int64_t tk_index(const UnitDev& U, int k, int e, int i) {
    return (int64_t)kTopK * U.ie_off + ((int64_t)k * U.E + e) * U.n_in + i;
}
__device__ __forceinline__ int64_t tks_index(const UnitDev& U, int k, int i) {
    return (int64_t)kTopK * U.in_off + (int64_t)k * U.n_in + i;
}

// XCD-aware tile order: consecutive workgroups are dispatched round-robin over the 8 XCDs, so give
// every XCD one contiguous range of tiles (neighbouring tiles read neighbouring out-span ranges and
// then share that XCD's L2).  Bijective for any tile count.
__device__ __forceinline__ int xcd_tile(int b, int n) {
    const int q = n / 8, r = n % 8, x = b % 8, j = b / 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

}  // namespace tw
