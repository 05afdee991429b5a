// tw_engine.hip -- host side of libtwgpu.so: the C-ABI of include/traceweaver_amd.h on top of the
// kernels in tw_kernels.h.  One engine = one HIP device + one stream; inputs stay resident in HBM
// between the two passes (spans are read from host memory exactly once, in tw_load_batch).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <type_traits>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "traceweaver_amd.h"
#include "tw_kernels.h"
#include "tw_fit.h"
#include "tw_eval.h"
#include "tw_skip.h"
#include "tw_load.h"
#include "tw_baselines.h"

using namespace tw;

namespace {

constexpr int kMaxRepairRounds = 1 << 20;
enum { ST_EMPTY = 0, ST_LOADED = 1, ST_PASS1 = 2, ST_MIX = 3, ST_PASS2 = 4 };
enum { EV_BEGIN = 0, EV_PARAMS, EV_ENUM0, EV_ENUM1, EV_WIN, EV_SEL, EV_REPAIR, EV_END, EV_COUNT };

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// MT19937 as numpy's RandomState runs it (the published reference algorithm of Matsumoto & Nishimura; seeding of an integer
// seed = init_genrand, `random_sample` = (a >> 5, b >> 6) -> (a * 2^26 + b) / 2^53): `Mt19937(s).random_sample()` yields the
// doubles of `np.random.RandomState(s).random_sample()`.  Used for the k-means++ draws of the mixture refit (tw_fit.h).
struct Mt19937 {
    uint32_t mt[624];
    int idx = 624;
    explicit Mt19937(uint32_t seed = 0u) {
        mt[0] = seed;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    }
    uint32_t next() {
        if (idx >= 624) {
            for (int i = 0; i < 624; i++) {
                const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
                mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    double random_sample() {
        const uint32_t a = next() >> 5, b = next() >> 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
};

}  // namespace

struct tw_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t cls_stream[kMaxEp + 1] = {};   // one stream per endpoint count: the enumeration kernels of different classes overlap
    hipEvent_t cls_ev[kMaxEp + 2] = {};        // [0] fork, [E] class E's enumeration done
    // the window / selection stage of a class follows its enumeration on the class' stream (run_pass): the three instantiations of
    // k_select_heavy on streams of their own beside it, forked from and joined to the class' stream
    hipStream_t sel_stream[3] = {};
    // ... and a second set for the stages of the other classes that hold a large share of the batch's spans (their searches one after the
    // other on the class' stream were the longest chain of a pass on the nodejs shape: 4.7 + 8.7 + 1.8 ms for the three one-endpoint services)
    hipStream_t sel_stream2[3] = {};
    int64_t select_fork_min = 0;         // TW_SELECT_FORK_MIN: incoming spans of a class from which its stage forks too (0 = only the deepest class)
    hipEvent_t prep_ev = nullptr;              // the per-pass fills the selection stage needs (main stream, beside the enumerations)
    hipEvent_t post_fork[kMaxEp + 1] = {}, post_join[kMaxEp + 1][3] = {}, post_done[kMaxEp + 1] = {};
    hipEvent_t tile_ev[kMaxEp + 1] = {};       // class E's tile kernel done (launch_enumerate: gating of the classes launched after it)
    // a class of many tiles is launched in stretches: the wavefront kernel of a stretch's listed spans on the class' second stream
    // beside the tile kernel of the next stretch (launch_enumerate)
    hipStream_t cls_stream2[kMaxEp + 1] = {};
    hipEvent_t stretch_ev[kMaxEp + 1][kEnumStretches] = {}, stretch_done[kMaxEp + 1] = {};
    int enum_stretches = 1, stretch_min_tiles = 2048;   // TW_ENUM_STRETCHES (1 = one launch: the default, see launch_enumerate), TW_STRETCH_MIN_TILES (tiles per stretch at least)
    int heavy_grid = 4096, select_grid = 0;             // TW_HEAVY_GRID: most persistent wavefronts of k_enumerate_heavy per launch; TW_SELECT_GRID: of k_select_heavy (0 = by the layout's LDS, select_heavy_grid)
    int debug_lists = 0;                                // TW_DEBUG_LISTS=1: the list counters of the first enumeration are copied to the host per class (tw_debug_worklists)
    int32_t first_lists[3][kMaxEp + 1] = {};            // ... narrow, wide, long enumerations
    hipEvent_t gate_ev = nullptr;              // ... the last one recorded in the current launch_enumerate_all
    int tile_gate = 0;                         // TW_TILE_GATE (measured on the media shape: the tile kernels one after the other take 1.9 + 1.5 + 1.9 ms, beside one another 3.7 -- off)
    int stage_min_tiles = 1024;                // TW_STAGE_MIN_TILES: batches of fewer tiles per class on average join the classes after the enumeration (a stage is 17 small
                                               // launches per class: the 1 M-span Alibaba-shape slice, eight classes of 300 tiles, takes 7.8 ms per step staged and 6.4 joined)
    int32_t *rank_in = nullptr, *rank_out = nullptr;   // scratch of sort_ends: rank of a span's end time less its index
    bool class_params = false;                 // pass 1 of a staged batch: sort_ends / k_block_params per class on the class' stream
    int prio_min_e = 0, prio_high = 0;
    bool prio_any = false;
    int pipeline = 1;                          // TW_CLASS_PIPELINE=0: every class joins the engine's stream after its enumeration (measurements, tests)
    std::string err;
    int state = ST_EMPTY;
    int tile = kTile;   // incoming spans (threads) per workgroup of the per-span kernels
    int tile_threads = 2; // threads per incoming span of k_enumerate_tile (its items and tuples are spread over all of them)
    int coop = kCoop;   // threads of the per-unit cooperative kernels
    std::vector<UnitDev> units;
    std::vector<TileDev> tiles;
    std::vector<int64_t> gs_off_h;
    std::vector<void*> allocs;
    // the arrays of a batch live in one device allocation that is kept (and grown) across tw_load_batch calls:
    // ~60 hipMalloc / hipFree pairs per batch otherwise cost as much as the transfer itself
    void* arena = nullptr;
    size_t arena_cap = 0;
    bool arena_open = false;
    std::vector<std::pair<std::function<void(void*)>, size_t>> arena_req;
    Dev P{};
    int64_t n_ie = 0, n_gp = 0, n_slots = 0, n_gaps = 0;
    // scratch for scans / sort
    PairVI* agg_pair = nullptr;
    int32_t *agg_i32 = nullptr, *agg_i32b = nullptr;
    uint32_t *seg_in = nullptr, *seg_out = nullptr;
    int n_seg_out = 0;
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    double* mix_p_dev = nullptr;
    int32_t* mix_n_dev = nullptr;
    double* mix_c_dev = nullptr;
    double* gaps_sorted = nullptr;
    double* fit_models = nullptr;
    double* fit_uval = nullptr;
    int32_t *fit_ustart = nullptr, *fit_row_n = nullptr, *fit_row_uniq = nullptr;
    double* fit_centres = nullptr;
    double *fit_tape = nullptr, *fit_tape100 = nullptr;   // uniforms of the k-means++ seedings: model-selection fits / the refit (MT19937(100))
    int64_t* fit_tape_off = nullptr;
    int64_t fit_tape_cap = 0;
    bool fit_runs_pending = false;          // k_fit_runs is queued, its refusal flag not read yet (fit_prepare_launch / fit_prepare)
    bool fit_prepared = false;              // the gap rows of the resident pass-1 result are sorted and run-length compressed
    int tile_sub_max = 8;
    int32_t hard_pass1[kMaxEp + 1] = {};    // per class: windows its stage listed for k_select_dp in pass 1 (-1 not known)
    int8_t wide_pass1[kMaxEp + 1] = {};     // per class: -1 not known, 0 / 1 = the first enumeration of pass 1 listed no / some span with wide windows
    int32_t split_pass1[kMaxEp + 1] = {};   // per class: spans the first enumeration of pass 1 cut into parts (-1 not known): sizes the merge launch of pass 2 (a guess that only costs time when wrong)
    int lean_pool = 512;                    // doubles of LDS for the pair tables of k_enumerate_lean (TW_LEAN_POOL)
    bool pass1_done = false;                // tw_run_pass1 has run on the resident batch (tw_run_pass2 reads its cut-offs, windows, tuple counts)
    std::vector<int32_t> fit_max_n;         // per slot min(5, #unique), 0 = nothing to fit (host copy, tw_fit_rows)
    bool fit_max_n_valid = false;           // ... of the rows that are prepared now (cleared with fit_prepared)
    Mt19937 fit_rng;                        // stream of tw_fit_mixtures (tw_set_fit_seed; restarted by every tw_load_batch)
    uint32_t fit_seed = 0;
    int32_t* slot_unit = nullptr;
    int32_t* tile_ids = nullptr;            // tiles grouped by the unit's endpoint count
    // The small counters of a pass in one block, so that a pass (and a repair round) starts with one fill instead of a dozen:
    // [work-list counters .. frontier cursors] are what a repair round resets, the whole block what a pass resets.
    int32_t* ctr = nullptr;
    int64_t ctr_round_ints = 0, ctr_pass_ints = 0;
    int32_t tile_cls_off[kMaxEp + 2] = {};  // class E owns tile_ids[tile_cls_off[E] .. tile_cls_off[E+1])
    uint32_t *seg_gap = nullptr, *seg_gap_end = nullptr, *seg_gap_dst = nullptr;
    unsigned long long *comp_a = nullptr, *comp_b = nullptr;  // composite keys of sort_rows
    int64_t comp_cap = 0, n_gap_scored = 0;
    int32_t *truth = nullptr, *in_trace = nullptr;  // tw_set_truth
    uint8_t* trace_bad = nullptr;           // [2][n_traces]
    unsigned long long* eval_counts = nullptr;  // [n_units][4] + [2]
    int64_t n_traces = 0, trace_cap = 0;
    uint8_t* slot_scored = nullptr;
    int64_t n_gap_rows = 0;
    unsigned long long* key_acc = nullptr;  // [2] scratch of k_key_bits
    double fit_ms = 0.0;
    int rounds = 0;                         // repair rounds of the last pass
    bool skip_mode = false;                 // skip-mode batch (tw_batch.skip): one pass with skip spans
    SkipUnitDev* skip_units = nullptr;
    int32_t* skip_perm = nullptr; int64_t* skip_tw = nullptr; int32_t* skip_pool = nullptr; double* skip_dist = nullptr; long long* skip_fetch = nullptr;
    int64_t skip_fetch_n = 0;
    // tw_scale_load: the table as uploaded (kept so that every load level starts from it)
    int64_t *orig_is = nullptr, *orig_ie = nullptr, *orig_os = nullptr, *orig_oe = nullptr;
    int32_t *orig_truth = nullptr, *orig_trace = nullptr;
    bool scaled_upload = false;             // the batch was uploaded with unit_time_scale (already load-scaled by the caller)
    hipEvent_t ev[EV_COUNT] = {};
    double ms[6] = {0, 0, 0, 0, 0, 0};
    double host_ms[2] = {0, 0};             // host wall clock of the last pass: submitting the first enumeration / the whole tw_run_pass call
};

namespace {

int fail(tw_engine* e, int code, const std::string& msg) {
    e->err = msg;
    return code;
}

#define HIPCHK(call)                                                                               \
    do {                                                                                           \
        hipError_t _s = (call);                                                                    \
        if (_s != hipSuccess)                                                                      \
            return fail(e, TW_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(_s));      \
    } while (0)

template <class T>
int dev_alloc(tw_engine* e, T** p, int64_t count) {
    const size_t bytes = (size_t)std::max<int64_t>(count, 1) * sizeof(T);
    if (e->arena_open) {  // tw_load_batch: collected, placed by arena_commit
        *p = nullptr;
        e->arena_req.emplace_back([p](void* q) { *p = (T*)q; }, (bytes + 255) / 256 * 256);
        return TW_OK;
    }
    void* q = nullptr;
    HIPCHK(hipMalloc(&q, bytes));
    e->allocs.push_back(q);
    *p = (T*)q;
    return TW_OK;
}

int ensure_class_stream(tw_engine* e, int E) {
    if (e->cls_stream[E] != nullptr) return TW_OK;
    if (e->prio_min_e > 0 && E >= e->prio_min_e && e->prio_any) HIPCHK(hipStreamCreateWithPriority(&e->cls_stream[E], hipStreamDefault, e->prio_high));
    else HIPCHK(hipStreamCreate(&e->cls_stream[E]));
    return TW_OK;
}

int arena_commit(tw_engine* e) {
    e->arena_open = false;
    size_t total = 0;
    for (const auto& r : e->arena_req) total += r.second;
    if (total > e->arena_cap) {
        if (e->arena != nullptr) (void)hipFree(e->arena);
        e->arena = nullptr;
        e->arena_cap = 0;
        HIPCHK(hipMalloc(&e->arena, total));
        e->arena_cap = total;
    }
    size_t off = 0;
    for (const auto& r : e->arena_req) { r.first((char*)e->arena + off); off += r.second; }
    e->arena_req.clear();
    return TW_OK;
}

void free_all(tw_engine* e) {
    for (void* q : e->allocs) (void)hipFree(q);
    e->allocs.clear();
    e->orig_is = e->orig_ie = e->orig_os = e->orig_oe = nullptr; e->orig_truth = e->orig_trace = nullptr;
    e->truth = nullptr; e->in_trace = nullptr; e->trace_bad = nullptr; e->eval_counts = nullptr; e->n_traces = 0; e->trace_cap = 0;
    e->state = ST_EMPTY;
}

const char* kernel_error_text(int code) {
    switch (code) {
        case TW_ERR_WINDOW_WIDTH: return "an incoming span has more candidate spans at one endpoint than the candidate bitmap holds";
        case TW_ERR_WINDOW_SIZE: return "a window holds more than TW_MAX_WINDOW incoming spans";
        case TW_ERR_NAN_PARAMS: return "a parameter block holds a single sample (std = NaN); the reference aborts here (n_in % 100 == 1)";
        case TW_ERR_SKIP_PARAMS: return "skip mode: the scorer needs a (mean, std) pair that BuildDistributions did not produce (the reference raises KeyError)";
        case TW_ERR_SKIP_REFERENCE_RAISES: return "skip mode: the reference raises on this input (all endpoints skipped, or a score tie compared through a skip span)";
        case TW_ERR_ARG: return "skip mode: a request starts before the first time window";
        default: return "kernel reported an error";
    }
}

// (S: all tiles or the tiles of one endpoint-count class, TileSet in tw_device.h)
template <class Tr>
int run_scan(tw_engine* e, const TileSet& S, hipStream_t st, typename Tr::T* agg) {
    const Dev& P = e->P;
    hipLaunchKernelGGL((k_scan_local<Tr>), dim3(S.n), dim3(e->tile), 0, st, P, S, agg);
    hipLaunchKernelGGL((k_scan_spine<Tr>), dim3(P.n_units), dim3(e->coop), 0, st, P, S, agg);
    hipLaunchKernelGGL((k_scan_fix<Tr>), dim3(S.n), dim3(e->tile), 0, st, P, S, agg);
    HIPCHK(hipGetLastError());
    return TW_OK;
}

TileSet all_tiles_host(const tw_engine* e) { return TileSet{nullptr, 0, e->P.n_tiles, 0}; }
TileSet class_tiles(const tw_engine* e, int E) { return TileSet{e->tile_ids + e->tile_cls_off[E], e->tile_cls_off[E], e->tile_cls_off[E + 1] - e->tile_cls_off[E], E}; }

// (no-skip batches: every endpoint list holds one span per incoming span, all lists sorted by (start, end))
int sort_ends(tw_engine* e, const TileSet& S, hipStream_t st) {   // (the counters e->rank_in / rank_out zeroed by the caller)
    const Dev& P = e->P;
    const dim3 tiles(S.n), tb(e->tile);
    hipLaunchKernelGGL(k_rank_ends, tiles, tb, 0, st, P, S, e->rank_in, e->rank_out);
    hipLaunchKernelGGL(k_place_ends, tiles, tb, 0, st, P, S, (const int32_t*)e->rank_in, (const int32_t*)e->rank_out);
    HIPCHK(hipGetLastError());
    return TW_OK;
}

// mode 0: first solve on all spans (per-thread kernel, then the wavefront kernel for the spans it deferred);
// mode 1: the spans listed by k_detect_gone, without the candidate spans earlier windows took
// The classes (units of one endpoint count) are independent of one another and every class' kernels end in a long tail
// (work per span spans four orders of magnitude): each class runs on a stream of its own, forked from and joined to the
// engine's stream by events, so that the tails overlap.  The wide instantiation of a class owns the big-list pool slots
// together with the narrow one (one counter): no conflict, they only ever add.
// Extra work-list entries of a class for the parts of its split enumerations.  Up to four endpoints few spans are split (an eighth of
// the class + 64 was never short); in the deep call graphs most wavefront-enumerated spans are, and a budget that runs out leaves
// whichever spans come last unsplit -- single wavefronts then hold the class' kernel for milliseconds (round 5: 3 of its 4 ms).
// (Round 6, with the budget kept exactly -- bump_reserve in tw_kernels.h: the media shape at eight requests in flight asks for parts
// for 2 400 of its 20 000 four-endpoint spans and an eighth of the class serves 520 of them; serving all of them -- TW_PART_BUDGET_SHALLOW
// = 2 entries per span, like the deep classes -- is no faster: 10.5 vs 9.5 ms per step, k_merge_parts then replays 2 400 spans' logs.
// One-endpoint classes never split: no budget.)
int64_t part_extra(int cls, int64_t n_cls) {
    if (n_cls <= 0) return 0;
    static const int deep = env_int("TW_PART_BUDGET_DEEP", 2), shallow = env_int("TW_PART_BUDGET_SHALLOW", 0);
    if (cls <= 1) return 64;
    const int mult = cls >= 5 ? deep : shallow;
    return (mult > 0 ? n_cls * mult : n_cls / 8) + 64;
}

// `listed` (mode 1 only): the class' entries of heavy_in_count as the host read them with the round's change count -- a class, or an
// instantiation, with nothing listed is not launched at all.
// The whole chain of a class on the class' stream: tile kernel, wavefront kernels (k_enumerate_heavy<E>; from lean_min_e endpoints on
// k_enumerate_lean), k_merge_parts.  (k_enumerate_lean and k_merge_parts can serve several classes in one launch; measured on the
// Alibaba-shape slice, one launch per phase for the four deep classes: 2.5 ms per pass instead of 1.9 -- the phases of different classes
// no longer overlap, and every phase ends with its longest item.)
template <int E>
void launch_enumerate(tw_engine* e, int pass, int mode, const int32_t* listed) {
    const int nt = e->tile_cls_off[E + 1] - e->tile_cls_off[E];
    if (nt == 0) return;
    // (-1: not known.  The first enumeration of pass 2 lists the spans pass 1 listed: a class without wide windows then has none now)
    const int n_narrow = listed != nullptr ? listed[E] : -1, n_wide = listed != nullptr ? listed[kMaxEp + 1 + E] : ((pass == 2 && mode == 0 && e->wide_pass1[E] == 0) ? 0 : -1);
    if (n_narrow == 0 && n_wide == 0) return;
    hipStream_t st = e->cls_stream[E];
    (void)hipStreamWaitEvent(st, e->cls_ev[0], 0);
    // The tile kernel of a class that fills the GPU runs before those of the classes with fewer endpoints instead of beside them: its
    // chain (wavefront kernels, merge, selection: few wavefronts, long tails) is the longest of the pass and overlaps with the tile
    // kernels that follow, instead of starting when everything else has ended (TW_TILE_GATE: tiles from which a class gates, 0 = never)
    if (mode == 0 && e->gate_ev != nullptr) (void)hipStreamWaitEvent(st, e->gate_ev, 0);
    const Dev& P = e->P;
    const dim3 hb(std::min(e->coop, kHeavyThreads));
    const int pool = E > 4 ? 2048 : (E > 1 ? kPairPoolPerEp * E : 1);       // doubles of pair-term tables per wavefront of k_enumerate_heavy
    const size_t pool_bytes = sizeof(double) * (size_t)pool;
    // (k_enumerate_lean: root / closing term tables + an LDS pool of pair tables for the common case, larger pair tables in global memory)
    const size_t lean_bytes = sizeof(double) * (size_t)(2 * kMaxEp * kNarrow + e->lean_pool), lean_bytes_w = sizeof(double) * (size_t)(2 * kMaxEp * 64 * kCandWords + e->lean_pool);
    const bool lean = E >= P.lean_min_e && !e->skip_mode;   // the deep call graphs: k_enumerate_lean; what it hands on to k_enumerate_heavy is launched by launch_enumerate_all
    const int cap = P.heavy_in_off[E + 1] - P.heavy_in_off[E];
    auto grid_for = [&](int known, int most) { return std::max(std::min(((known >= 0 ? known : cap) + kWorkChunk - 1) / kWorkChunk, most), 1); };   // persistent wavefronts pulling spans from the work list
    auto narrow = [&](int part, int g) {
        if (lean) hipLaunchKernelGGL((k_enumerate_lean<kNarrow>), dim3(g), hb, lean_bytes, st, P, pass, mode, part, e->lean_pool, E, E);
        else hipLaunchKernelGGL((k_enumerate_heavy<E, kNarrow>), dim3(g), hb, pool_bytes, st, P, pass, mode, part, pool);
    };
    auto wide = [&](int part, int g) {
        if (lean) hipLaunchKernelGGL((k_enumerate_lean<64 * kCandWords>), dim3(g), hb, lean_bytes_w, st, P, pass, mode, part, e->lean_pool, E, E);
        else hipLaunchKernelGGL((k_enumerate_heavy<E, 64 * kCandWords>), dim3(g), hb, pool_bytes, st, P, pass, mode, part, pool);
    };
    bool stretched = false;
    if (mode == 0 && pass == 1 && e->class_params) {   // the class' sorted end times and block parameters (run_pass)
        const TileSet S = class_tiles(e, E);
        (void)sort_ends(e, S, st);
        const int64_t total = e->n_gp;
        hipLaunchKernelGGL(k_block_params, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, P, total, E);
    }
    if (mode == 0) {
        // cut-offs and work lists, the short enumerations: the tile kernel.  A class of few tiles: several workgroups per tile, until
        // the class fills the CUs twice over
        const dim3 tile_block(e->tile >= 64 ? e->tile * e->tile_threads : e->tile);
        int sub = 1;
        while (sub < e->tile_sub_max && nt * sub < 512 && (e->tile / (sub * 2)) * (sub * 2) == e->tile && e->tile / (sub * 2) >= 8) sub *= 2;
        // A class of many tiles in stretches: what a stretch's tiles listed for the wavefront kernel is enumerated on the class' second
        // stream beside the tile kernel of the next stretch (the wavefront kernel of the whole class behind the last tile ran 1.2-1.6 ms
        // at 1.5 wavefronts per SIMD on the media shape, a quarter of the class' chain).  Not for the classes whose wavefront kernel
        // appends list parts behind the listed entries (E >= defer_min_e) nor for k_enumerate_lean.
        const int ns = (!lean && !e->skip_mode && !(E >= 3 && E >= P.defer_min_e) && e->cls_stream2[E] != nullptr)
                           ? std::max(std::min(e->enum_stretches, nt / e->stretch_min_tiles), 1) : 1;
        stretched = ns > 1;
        for (int j = 0; j < ns; j++) {
            const int t0 = (int)((int64_t)nt * j / ns), t1 = (int)((int64_t)nt * (j + 1) / ns);
            const int32_t* ids = e->tile_ids + e->tile_cls_off[E] + t0;
            if (e->tile == kTile && e->tile / sub == kSubTileSpans)   // (the instantiation with tables for a sub-tile's 16 spans: a third of the LDS)
                hipLaunchKernelGGL((k_enumerate_tile<E, kSubTileSpans>), dim3((t1 - t0) * sub), tile_block, 0, st, P, pass, ids, t1 - t0, sub);
            else
                hipLaunchKernelGGL((k_enumerate_tile<E>), dim3((t1 - t0) * sub), tile_block, 0, st, P, pass, ids, t1 - t0, sub);
            if (ns > 1) {
                hipLaunchKernelGGL(k_enum_snapshot, dim3(1), dim3(1), 0, st, P, E, j + 1);
                if (j + 1 < ns) {   // (the last stretch's spans: on the class' stream itself)
                    hipStream_t s2 = e->cls_stream2[E];
                    (void)hipEventRecord(e->stretch_ev[E][j], st);
                    (void)hipStreamWaitEvent(s2, e->stretch_ev[E][j], 0);
                    hipLaunchKernelGGL((k_enumerate_heavy<E, kNarrow>), dim3(grid_for(-1, e->heavy_grid)), hb, pool_bytes, s2, P, pass, mode, (j + 1) << 8, pool);
                } else {
                    hipLaunchKernelGGL((k_enumerate_heavy<E, kNarrow>), dim3(grid_for(-1, e->heavy_grid)), hb, pool_bytes, st, P, pass, mode, (j + 1) << 8, pool);
                    (void)hipEventRecord(e->stretch_done[E], e->cls_stream2[E]);
                    (void)hipStreamWaitEvent(st, e->stretch_done[E], 0);
                }
            }
        }
        if (e->tile_gate > 0 && nt >= e->tile_gate) { (void)hipEventRecord(e->tile_ev[E], st); e->gate_ev = e->tile_ev[E]; }
    }
    // the wavefront kernels of the class: the spans the tile kernel handed over (mode 1: the spans k_detect_gone listed), the long
    // enumerations first; the list parts of the spans those launches deferred (kListSplitFlag); the parts combined, the few spans whose
    // order of equal scores the parts left undecided enumerated whole.  Every launch of the chain has a work-list cursor of its own
    // (enum_cursor): nothing is reset in between.
    if (n_narrow != 0 && !stretched) narrow(0, grid_for(n_narrow, e->heavy_grid));
    if (n_wide != 0) wide(0, grid_for(n_wide, 1024));   // (stretched: the wide instantiation once, over the whole lists)
    if (mode == 0 && E >= 3 && E >= P.defer_min_e) { narrow(3, grid_for(-1, 4096)); if (n_wide != 0) wide(3, grid_for(-1, 1024)); }
    if (mode == 0 && E > 1) {
        // (34 KB of LDS a workgroup: a thousand of them ask for all there is -- and wait for it while the selection kernels of other
        // classes hold the CUs' LDS: the launch for the two-endpoint class of the nodejs shape at 14.4 M spans, which has nothing to merge,
        // lasted 0.9-2.3 ms.  Pass 2 cuts the same spans as pass 1 -- the cuts follow the candidates' containment and the tuple counts --:
        // its launch is sized by pass 1's count; the workgroups stride over the records, so a wrong guess only costs time)
        const int nsplit = (pass == 2 && mode == 0) ? e->split_pass1[E] : -1;
        const int merge_grid = nsplit < 0 ? (cap >= (1 << 20) ? 1024 : 256) : std::max(std::min(nsplit, 1024), 8);
        hipLaunchKernelGGL(k_merge_parts, dim3(merge_grid), dim3(std::min(e->coop, 64)), 0, st, P, pass, E, E);
        narrow(1, nsplit == 0 ? 8 : 256); if (n_wide != 0) wide(1, nsplit == 0 ? 8 : 64);
    }
    if (e->debug_lists != 0 && mode == 0) {
        (void)hipMemcpyAsync(&e->first_lists[0][E], P.heavy_in_count + E, sizeof(int32_t), hipMemcpyDeviceToHost, st);
        (void)hipMemcpyAsync(&e->first_lists[1][E], P.heavy_in_count + kMaxEp + 1 + E, sizeof(int32_t), hipMemcpyDeviceToHost, st);
        (void)hipMemcpyAsync(&e->first_lists[2][E], P.heavy_big_count + E, sizeof(int32_t), hipMemcpyDeviceToHost, st);
    }
    (void)hipEventRecord(e->cls_ev[E], st);   // (who waits for it: launch_enumerate_all)
}

// What k_enumerate_lean handed on (whole spans only: P.fb_*), class by class, by k_enumerate_heavy (part 4).  Rare at scale, and a launch
// of that kernel -- 256 + 256 registers for eight endpoints: it needs a SIMD to itself -- waits for one even when its list is empty: the
// host reads the counts (one small copy, the enumeration has to be complete anyway) and launches only what has entries.
template <int E>
void launch_fallback(tw_engine* e, int pass, int mode, const int32_t* fb) {
    if (fb[E] <= 0) return;
    hipStream_t st = e->cls_stream[E];
    (void)hipStreamWaitEvent(st, e->cls_ev[0], 0);
    const dim3 hb(std::min(e->coop, kHeavyThreads));
    const int pool = E > 4 ? 2048 : (E > 1 ? kPairPoolPerEp * E : 1);
    const int grid = std::max(std::min((fb[E] + kWorkChunk - 1) / kWorkChunk, 1024), 1);
    hipLaunchKernelGGL((k_enumerate_heavy<E, kNarrow>), dim3(grid), hb, sizeof(double) * (size_t)pool, st, e->P, pass, mode, 4, pool);
    hipLaunchKernelGGL((k_enumerate_heavy<E, 64 * kCandWords>), dim3(grid), hb, sizeof(double) * (size_t)pool, st, e->P, pass, mode, 4, pool);
    (void)hipEventRecord(e->cls_ev[E], st);
}


// CreateWindows2 + PerfectCut (traceweaver_v3.py:1020-1078) over the tiles of S on stream st: needs the candidate sets of S's spans.
// (the running maximum of the request ends -- PerfectCut's prev_index -- depends on the spans alone: run_pass computes it over all
// tiles on the engine's stream beside the enumerations.  The rest in five launches, k_cut_scan .. k_index_fix in tw_kernels.h)
int launch_windows(tw_engine* e, const TileSet& S, hipStream_t st) {
    const Dev& P = e->P;
    const dim3 tiles(S.n), tb(e->tile);
    hipLaunchKernelGGL(k_cut_scan, tiles, tb, 0, st, P, S, e->agg_i32);
    hipLaunchKernelGGL((k_scan_spine<ScanSegStart>), dim3(P.n_units), dim3(e->coop), 0, st, P, S, e->agg_i32);
    hipLaunchKernelGGL(k_flags_scan, tiles, tb, 0, st, P, S, (const int32_t*)e->agg_i32, e->agg_i32b);
    hipLaunchKernelGGL((k_scan_spine<ScanWinId>), dim3(P.n_units), dim3(e->coop), 0, st, P, S, e->agg_i32b);
    hipLaunchKernelGGL(k_index_fix, tiles, tb, 0, st, P, S, (const int32_t*)e->agg_i32b);
    HIPCHK(hipGetLastError());
    return TW_OK;
}

// The listed windows of the tile set S: those of up to kBruteMax spans (nearly all of them) by k_select_tiny, whose workgroups hold 1 KB
// of LDS and fill the SIMDs, on stream st; the middle ones (up to kBigWindow - 1 spans: one-word masks, 7 KB) and the long ones (22 KB:
// seven wavefronts per CU, ends with its longest search) by the instantiations of k_select_heavy -- each on a stream of its own,
// forked from and joined to st by events, so that the many short windows run beside the long searches' tail instead of before it.
// What the searches give up on goes to one list for k_select_dp (the caller launches it when every set has joined).
// (fork = false: everything on st, one after the other -- the stages of the classes that end before the last one have the time, and the
// three selection streams stay free for the class whose stage ends the pass)
// the three instantiations of k_select_heavy with room for EMAX endpoints per candidate (SelectLdsT: what a workgroup holds in LDS decides
// how many windows the GPU searches at a time)
// Persistent workgroups of one wavefront each, as many as the CUs hold at a time by the layout's LDS (the window's data + the small tables
// of the level solver): 2 800 / 4 300 / 6 100 for the three layouts of a two-endpoint class.  (One figure for all three, 4 096, left the
// small layouts short of wavefronts and gave the large one a second round: nodejs shape at 14.4 M spans 40.9 -> 39.2 ms with 6 144, 41.4
// with 8 192.)  TW_SELECT_GRID > 0 is that one figure again.
template <class LDS>
unsigned select_heavy_grid(const tw_engine* e, dim3 most) {
    if (e->select_grid > 0) return std::min<unsigned>(most.x, (unsigned)e->select_grid);
    const size_t lds = sizeof(LDS) + sizeof(SelectDpT<LDS::kW, kDpCapSmall, kDpSlotsSmall>) + 512;
    const unsigned resident = 256u * (unsigned)std::max<size_t>((160u * 1024u) / lds, 1);
    return std::max(std::min(most.x, std::min(resident, 8192u)), 1u);
}
template <int EMAX>
void launch_select_heavy3(tw_engine* e, const TileSet& S, dim3 most, dim3 wave, hipStream_t s0, hipStream_t s1, hipStream_t s2) {
    hipLaunchKernelGGL((k_select_heavy<SelectLdsLvlE<EMAX>, 3>), dim3(select_heavy_grid<SelectLdsLvlE<EMAX>>(e, most)), wave, 0, s0, e->P, S);      // the longest searches first
    hipLaunchKernelGGL((k_select_heavy<SelectLdsBigLvlE<EMAX>, 1>), dim3(select_heavy_grid<SelectLdsBigLvlE<EMAX>>(e, most)), wave, 0, s1, e->P, S);
    hipLaunchKernelGGL((k_select_heavy<SelectLdsMidLvlE<EMAX>, 2>), dim3(select_heavy_grid<SelectLdsMidLvlE<EMAX>>(e, most)), wave, 0, s2, e->P, S);
}
void launch_select_heavy3(tw_engine* e, const TileSet& S, dim3 most, dim3 wave, hipStream_t s0, hipStream_t s1, hipStream_t s2) {
    int emax = S.slot;   // a class' tile set: its endpoint count; all tiles: the batch's largest
    if (emax == 0)
        for (int c = 1; c <= kMaxEp; c++) if (e->tile_cls_off[c + 1] > e->tile_cls_off[c]) emax = c;
    if (emax <= 2) launch_select_heavy3<2>(e, S, most, wave, s0, s1, s2);
    else if (emax <= 4) launch_select_heavy3<4>(e, S, most, wave, s0, s1, s2);
    else launch_select_heavy3<kMaxEp>(e, S, most, wave, s0, s1, s2);
}

void launch_select_listed(tw_engine* e, const TileSet& S, hipStream_t st, int64_t n_spans, bool fork = true, bool second = false) {
    const Dev& P = e->P;
    const dim3 wave(std::min(e->coop, 64));
    const dim3 grid((unsigned)std::min<int64_t>(n_spans / 2 + 1, 1 << 20));   // (what the set could need at most; launch_select_heavy3 sizes each layout's launch)
    if (!fork) {
        hipLaunchKernelGGL(k_select_tiny, dim3((unsigned)std::min<int64_t>(n_spans / 2 + 1, 8192)), wave, 0, st, P, S);
        launch_select_heavy3(e, S, grid, wave, st, st, st);
        return;
    }
    hipStream_t* ss = second ? e->sel_stream2 : e->sel_stream;
    (void)hipEventRecord(e->post_fork[S.slot], st);
    for (int j = 0; j < 3; j++) (void)hipStreamWaitEvent(ss[j], e->post_fork[S.slot], 0);
    launch_select_heavy3(e, S, grid, wave, ss[0], ss[1], ss[2]);
    for (int j = 0; j < 3; j++) (void)hipEventRecord(e->post_join[S.slot][j], ss[j]);
    hipLaunchKernelGGL(k_select_tiny, dim3((unsigned)std::min<int64_t>(n_spans / 2 + 1, 8192)), wave, 0, st, P, S);
    for (int j = 0; j < 3; j++) (void)hipStreamWaitEvent(st, e->post_join[S.slot][j], 0);
}

// the windows the searches gave up on (kDpNodes), level by level over 256 lanes each; nothing listed: the workgroups leave at once
// (`few`: the set listed no such window in pass 1 -- pass 2 rarely differs, and 512 workgroups of 70 KB of LDS take 0.1-0.2 ms to come and
// go with nothing to do; the workgroups are persistent, a small grid is only slower when the guess is wrong)
void launch_select_hard(tw_engine* e, const TileSet& S, hipStream_t st, bool few = false) {
    hipLaunchKernelGGL(k_select_dp, dim3(few ? 32 : (kDpCap <= 384 ? 512 : 256)), dim3(std::min(e->coop, kDpThreads)), 0, st, e->P, S);   // (one / two workgroups per CU by their LDS)
}

// What follows a class' first enumeration, on the class' stream: its windows (pass 1), the first selection of its windows, the first
// round of the span consumption (claim, detect) and -- in anticipation of that round finding nothing to repair, the usual case --
// the parents and gap samples of its units (run_pass runs both again over all tiles when a repair round changed something).  Units
// of different classes are independent (traceweaver_v3.py:1182-1193 runs per service): nothing here waits for another class.
int launch_class_stage(tw_engine* e, int pass, int E) {
    const TileSet S = class_tiles(e, E);
    if (S.n == 0) return TW_OK;
    hipStream_t st = e->cls_stream[E];
    (void)hipStreamWaitEvent(st, e->prep_ev, 0);
    if (pass == 1) { int rc = launch_windows(e, S, st); if (rc != TW_OK) return rc; }
    hipLaunchKernelGGL(k_select_fast, dim3(S.n), dim3(e->tile), 0, st, e->P, S);
    int deepest = 0;
    for (int c = 1; c <= kMaxEp; c++) if (e->tile_cls_off[c + 1] > e->tile_cls_off[c]) deepest = c;
    const int64_t n_cls = (int64_t)e->P.heavy_in_off[E + 1] - e->P.heavy_in_off[E];
    const bool second = E != deepest && e->select_fork_min > 0 && n_cls >= e->select_fork_min && e->sel_stream2[0] != nullptr;
    launch_select_listed(e, S, st, n_cls, E == deepest || second, second);
    launch_select_hard(e, S, st, pass == 2 && e->hard_pass1[E] == 0);
    const dim3 tiles(S.n), tb(e->tile);
    hipLaunchKernelGGL(k_claim, tiles, tb, 0, st, e->P, S, E);
    hipLaunchKernelGGL(k_detect_gone, tiles, tb, 0, st, e->P, S, 0, pass == 1 ? 2 : 1);
    (void)hipEventRecord(e->post_done[E], st);
    return TW_OK;
}

// mode 0 with `staged`: every class goes on to its windows and first selection on its own stream (launch_class_stage); the engine's
// stream waits for the enumerations (EV_ENUM1: the group's timer), then for the stages.
template <class Prep>
int launch_enumerate_all(tw_engine* e, int pass, int mode, const int32_t* listed, bool staged, Prep prep) {
    // (the work-list cursors were reset with the counter block of the pass / the repair round)
    (void)hipEventRecord(e->cls_ev[0], e->stream);
    e->gate_ev = nullptr;
    // the classes with the most endpoints first: their enumerations are the longest and end the group (each class runs on a
    // stream of its own; what is launched first is dispatched first)
    launch_enumerate<8>(e, pass, mode, listed); launch_enumerate<7>(e, pass, mode, listed); launch_enumerate<6>(e, pass, mode, listed); launch_enumerate<5>(e, pass, mode, listed);
    launch_enumerate<4>(e, pass, mode, listed); launch_enumerate<3>(e, pass, mode, listed); launch_enumerate<2>(e, pass, mode, listed); launch_enumerate<1>(e, pass, mode, listed);
    { int rc = prep(); if (rc != TW_OK) return rc; }   // (fills on the engine's stream, beside the enumerations)
    auto has = [&](int E) { return e->tile_cls_off[E + 1] > e->tile_cls_off[E]; };
    auto is_lean = [&](int E) { return E >= e->P.lean_min_e && !e->skip_mode; };
    bool any_lean = false;
    for (int E = 1; E <= kMaxEp; E++) any_lean |= has(E) && is_lean(E);
    // the classes that are complete with their chain go on at once -- those expected to end first first (the selection streams serve
    // the classes in this order): the fewest endpoints, or, when the tile kernels ran one after the other, the most
    const bool deep_first = e->gate_ev != nullptr;
    if (staged)
        for (int k = 1; k <= kMaxEp; k++) {
            const int E = deep_first ? kMaxEp + 1 - k : k;
            if (has(E) && !is_lean(E)) { int rc = launch_class_stage(e, pass, E); if (rc != TW_OK) return rc; }
        }
    if (any_lean) {
        // what k_enumerate_lean handed on: the host reads the counts when the lean classes' chains have ended
        int32_t fb[kMaxEp + 1] = {};
        int first_lean = 0;
        for (int E = kMaxEp; E >= 1; E--)
            if (has(E) && is_lean(E)) {
                if (listed == nullptr || listed[E] != 0 || listed[kMaxEp + 1 + E] != 0) HIPCHK(hipEventSynchronize(e->cls_ev[E]));
                first_lean = E;
            }
        HIPCHK(hipMemcpyAsync(fb, e->P.fb_count, sizeof(fb), hipMemcpyDeviceToHost, e->cls_stream[first_lean]));
        HIPCHK(hipStreamSynchronize(e->cls_stream[first_lean]));
        launch_fallback<8>(e, pass, mode, fb); launch_fallback<7>(e, pass, mode, fb); launch_fallback<6>(e, pass, mode, fb); launch_fallback<5>(e, pass, mode, fb);
        launch_fallback<4>(e, pass, mode, fb); launch_fallback<3>(e, pass, mode, fb); launch_fallback<2>(e, pass, mode, fb); launch_fallback<1>(e, pass, mode, fb);
        if (staged)
            for (int E = 1; E <= kMaxEp; E++)
                if (has(E) && is_lean(E)) { int rc = launch_class_stage(e, pass, E); if (rc != TW_OK) return rc; }
    }
    for (int E = 1; E <= kMaxEp; E++)
        if (has(E)) (void)hipStreamWaitEvent(e->stream, e->cls_ev[E], 0);
    HIPCHK(hipEventRecord(e->ev[EV_ENUM1], e->stream));
    if (staged) {
        HIPCHK(hipEventRecord(e->ev[EV_WIN], e->stream));   // (the windows are inside the class stages: what the timers show after this point is the part of the stages that outlasts the enumerations)
        for (int E = 1; E <= kMaxEp; E++)
            if (has(E)) (void)hipStreamWaitEvent(e->stream, e->post_done[E], 0);
    }
    return TW_OK;
}

// OR of (key ^ first key) and of the keys themselves over a set of index ranges (see k_key_bits)
int key_bits(tw_engine* e, const void* keys, const uint32_t* seg_begin, const uint32_t* seg_end, int nseg, int64_t total,
             unsigned long long out[2]) {
    out[0] = out[1] = 0;
    if (nseg <= 0 || total <= 0) return TW_OK;
    HIPCHK(hipMemsetAsync(e->key_acc, 0, 2 * sizeof(unsigned long long), e->stream));
    const int bx = (int)std::min<int64_t>(std::max<int64_t>(2048 / nseg, 1), total / nseg / 2048 + 1);  // ~2k workgroups at most
    hipLaunchKernelGGL(k_key_bits, dim3((unsigned)bx, (unsigned)std::min(nseg, 1024)), dim3(e->coop >= 64 ? 256 : e->coop), 0, e->stream,
                       (const unsigned long long*)keys, seg_begin, seg_end, nseg, e->key_acc);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, e->key_acc, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return TW_OK;
}

unsigned bit_length(unsigned long long x) {
    unsigned n = 0;
    while (x) { n++; x >>= 1; }
    return n;
}

int ensure_sort_tmp(tw_engine* e, size_t bytes) {
    if (bytes > e->sort_tmp_bytes) {
        void* q = nullptr;
        HIPCHK(hipMalloc(&q, bytes));
        e->allocs.push_back(q);
        e->sort_tmp = q;
        e->sort_tmp_bytes = bytes;
    }
    return TW_OK;
}

// Sorts every row [seg_begin[s], seg_end[s]) of 64-bit keys ascending into `out` (same positions).  Only the bits
// [begin_bit, end_bit) differ between keys and none of them is a sign bit that is set (callers check); `fixed` holds
// the common bits.  dst_off[s] = position of row s in the packed array (rows back to back), `packed` = its size.
// One device-wide radix sort over composite keys (see k_rows_pack); rows too many / keys too wide for 64 bits fall
// back to rocprim's segmented sort on the raw keys.
template <class Key>
int sort_rows(tw_engine* e, const Key* keys, Key* out, unsigned size, const uint32_t* seg_begin, const uint32_t* seg_end,
              const uint32_t* dst_off, int nseg, int64_t packed, unsigned begin_bit, unsigned end_bit, unsigned long long fixed) {
    if (nseg <= 0 || packed <= 0) return TW_OK;
    unsigned segbits = 0;
    while ((1u << segbits) < (unsigned)nseg) segbits++;
    const unsigned kb = end_bit - begin_bit;
    size_t bytes = 0;
    const bool mixed_signs = std::is_signed<Key>::value && end_bit >= 64;  // composite keys compare unsigned
    if (kb + segbits > 64 || kb == 0 || packed > e->comp_cap || mixed_signs) {
        HIPCHK(rocprim::segmented_radix_sort_keys(nullptr, bytes, keys, out, size, (unsigned)nseg, seg_begin, seg_end, begin_bit, end_bit, e->stream));
        int rc = ensure_sort_tmp(e, bytes);
        if (rc != TW_OK) return rc;
        bytes = e->sort_tmp_bytes;
        HIPCHK(rocprim::segmented_radix_sort_keys(e->sort_tmp, bytes, keys, out, size, (unsigned)nseg, seg_begin, seg_end, begin_bit, end_bit, e->stream));
        return TW_OK;
    }
    const int bx = (int)std::min<int64_t>(std::max<int64_t>(4096 / nseg, 1), packed / nseg / 1024 + 1);
    const dim3 grid((unsigned)bx, (unsigned)std::min(nseg, 1024)), block(e->coop >= 64 ? 256 : e->coop);
    hipLaunchKernelGGL(k_rows_pack, grid, block, 0, e->stream, (const unsigned long long*)keys, seg_begin, seg_end, dst_off, nseg,
                       (int)begin_bit, (int)kb, e->comp_a);
    HIPCHK(rocprim::radix_sort_keys(nullptr, bytes, e->comp_a, e->comp_b, (size_t)packed, 0u, kb + segbits, e->stream));
    int rc = ensure_sort_tmp(e, bytes);
    if (rc != TW_OK) return rc;
    bytes = e->sort_tmp_bytes;
    HIPCHK(rocprim::radix_sort_keys(e->sort_tmp, bytes, e->comp_a, e->comp_b, (size_t)packed, 0u, kb + segbits, e->stream));
    hipLaunchKernelGGL(k_rows_unpack, grid, block, 0, e->stream, (const unsigned long long*)e->comp_b, seg_begin, seg_end, dst_off, nseg,
                       (int)begin_bit, (int)kb, fixed, (unsigned long long*)out);
    HIPCHK(hipGetLastError());
    return TW_OK;
}

// (no-skip batches: every endpoint list holds one span per incoming span, all lists sorted by (start, end).  The counters
// borrow two per-span words that the pass writes only later: the window ids and the owner words)
int run_pass(tw_engine* e, int pass) {
    const Dev& P = e->P;
    const dim3 tiles(P.n_tiles), tb(e->tile);
    const auto host_t0 = std::chrono::steady_clock::now();
    HIPCHK(hipEventRecord(e->ev[EV_BEGIN], e->stream));
    HIPCHK(hipMemsetAsync(e->ctr, 0, sizeof(int32_t) * (size_t)e->ctr_pass_ints, e->stream));   // error flag, statistics, every work-list counter
#ifdef TW_PROFILE
    HIPCHK(hipMemsetAsync(P.prof + 5, 0, sizeof(unsigned long long) * 3, e->stream));    // the longest item / wavefront of THIS pass (the sums run on)
    HIPCHK(hipMemsetAsync(P.prof + 10, 0, sizeof(unsigned long long) * 3, e->stream));
#endif
    // Every class on its own stream: enumeration, then (staged) its windows and the first selection of its windows, while the longer
    // enumerations of other classes are still running.  TW_CLASS_PIPELINE=0 / skip mode / small batches: the classes join after the enumeration.
    int n_cls = 0;
    for (int E = 1; E <= kMaxEp; E++) n_cls += e->tile_cls_off[E + 1] > e->tile_cls_off[E] ? 1 : 0;
    const bool staged = !e->skip_mode && e->pipeline != 0 && P.n_tiles >= (int64_t)e->stage_min_tiles * std::max(n_cls, 1);
    // pass 1, ComputeEpPairDistParams3: sorted end times and block parameters -- staged: per class, at the head of the class' chain
    // (launch_enumerate: the class with the longest chain does not wait for the parameters of the others)
    e->class_params = false;
    if (pass == 1 && !e->skip_mode) {
        HIPCHK(hipMemsetAsync(e->rank_in, 0, sizeof(int32_t) * (size_t)std::max<int64_t>(P.n_in_total, 1), e->stream));
        HIPCHK(hipMemsetAsync(e->rank_out, 0, sizeof(int32_t) * (size_t)std::max<int64_t>(P.n_out_total, 1), e->stream));
        if (staged) e->class_params = true;
        else {
            int rc = sort_ends(e, all_tiles_host(e), e->stream);
            if (rc != TW_OK) return rc;
            const int64_t total = e->n_gp;
            hipLaunchKernelGGL(k_block_params, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, e->stream, P, total, 0);
        }
    }
    if (e->skip_mode) {
        // ComputeEpPairDistParams3 is not called when an endpoint is short of spans (traceweaver_v3.py:1177-1178); the first
        // enumeration only serves the candidate sets of the windows (DfsTraverse3), its scores are not used: neutral parameters
        const int64_t total = e->n_gp;
        hipLaunchKernelGGL(k_neutral_params, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, e->stream, P.gparam, total);
    }
    HIPCHK(hipEventRecord(e->ev[EV_PARAMS], e->stream));
    HIPCHK(hipEventRecord(e->ev[EV_ENUM0], e->stream));
    const auto host_t1 = std::chrono::steady_clock::now();
    {
        // what the selection stage and the repair rounds start from: filled on the engine's stream beside the enumerations
        int rc = launch_enumerate_all(e, pass, 0, nullptr, staged, [&]() -> int {
            if (pass == 1) { int rs = run_scan<ScanMaxEnd>(e, all_tiles_host(e), e->stream, e->agg_pair); if (rs != TW_OK) return rs; }
            if (e->skip_mode) { HIPCHK(hipEventRecord(e->prep_ev, e->stream)); return TW_OK; }
            HIPCHK(hipMemsetAsync(P.w_conf, 0, sizeof(int32_t) * (size_t)P.n_in_total, e->stream));
            HIPCHK(hipMemsetAsync(P.gone_valid, 0, (size_t)P.n_in_total, e->stream));   // (P.gone itself is not filled: k_detect_gone)
            HIPCHK(hipMemsetAsync(P.w_dirty, 0, (size_t)P.n_in_total, e->stream));
            HIPCHK(hipMemsetAsync(P.owner, 0x7f, sizeof(int32_t) * std::max<int64_t>(P.n_out_total, 1), e->stream));
            HIPCHK(hipEventRecord(e->prep_ev, e->stream));
            return TW_OK;
        });
        if (rc != TW_OK) return rc;
    }
    e->host_ms[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t1).count();
    if (pass == 1 && !staged) {
        int rc = launch_windows(e, all_tiles_host(e), e->stream);
        if (rc != TW_OK) return rc;
    }
    if (!staged) HIPCHK(hipEventRecord(e->ev[EV_WIN], e->stream));
    if (e->skip_mode) {   // the reference's loop in its own order, one wavefront per unit (tw_skip.h)
        HIPCHK(hipMemsetAsync(P.owner, 0x7f, sizeof(int32_t) * std::max<int64_t>(P.n_out_total, 1), e->stream));
        HIPCHK(hipMemsetAsync(e->skip_fetch, 0, sizeof(long long) * (size_t)std::max<int64_t>(e->skip_fetch_n, 1), e->stream));
        HIPCHK(hipMemsetAsync(P.w_dirty, 0, (size_t)P.n_in_total, e->stream));
        hipLaunchKernelGGL(k_skip_walk, dim3(P.n_units), dim3(std::min(e->coop, 64)), 0, e->stream, P, (const SkipUnitDev*)e->skip_units);
        HIPCHK(hipEventRecord(e->ev[EV_SEL], e->stream));
        HIPCHK(hipEventRecord(e->ev[EV_REPAIR], e->stream));
        hipLaunchKernelGGL(k_finalize, tiles, tb, 0, e->stream, P, all_tiles_host(e));
        HIPCHK(hipEventRecord(e->ev[EV_END], e->stream));
        HIPCHK(hipGetLastError());
        int32_t kerr0 = 0;
        HIPCHK(hipMemcpyAsync(&kerr0, P.err, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        if (kerr0 != 0) return fail(e, kerr0, kernel_error_text(kerr0));
        return TW_OK;
    }
    if (!staged) {
        hipLaunchKernelGGL(k_select_fast, tiles, tb, 0, e->stream, P, all_tiles_host(e));
        launch_select_listed(e, all_tiles_host(e), e->stream, P.n_in_total);
        launch_select_hard(e, all_tiles_host(e), e->stream);
    }
    HIPCHK(hipEventRecord(e->ev[EV_SEL], e->stream));
    // span consumption: rounds of claim / detect / re-enumerate / re-select until no span's set of taken candidates changes
    e->rounds = 0;
    for (int round = 0;; round++) {
        if (!(staged && round == 0)) {   // (staged: the first round ran class by class, launch_class_stage)
            if (round > 0) HIPCHK(hipMemsetAsync(P.owner, 0x7f, sizeof(int32_t) * std::max<int64_t>(P.n_out_total, 1), e->stream));
            hipLaunchKernelGGL(k_claim, tiles, tb, 0, e->stream, P, all_tiles_host(e), 0);
            HIPCHK(hipMemsetAsync(e->ctr, 0, sizeof(int32_t) * (size_t)e->ctr_round_ints, e->stream));   // work lists, round_changed, frontier cursors
            hipLaunchKernelGGL(k_detect_gone, tiles, tb, 0, e->stream, P, all_tiles_host(e), round, 0);
        }
        int32_t changed = 0, listed[2 * (kMaxEp + 1)] = {};
        HIPCHK(hipMemcpyAsync(&changed, P.round_changed, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(listed, P.heavy_in_count, sizeof(listed), hipMemcpyDeviceToHost, e->stream));   // (what k_detect_gone listed, per class and instantiation)
        HIPCHK(hipStreamSynchronize(e->stream));
        if (changed == 0) break;
        if (round >= kMaxRepairRounds) return fail(e, TW_ERR_DEVICE, "span consumption did not settle (more repair rounds than windows)");
        e->rounds = round + 1;
        { int rc = launch_enumerate_all(e, pass, 1, listed, false, []() { return (int)TW_OK; }); if (rc != TW_OK) return rc; }
        launch_select_listed(e, all_tiles_host(e), e->stream, P.n_in_total);
        launch_select_hard(e, all_tiles_host(e), e->stream);
    }
    HIPCHK(hipEventRecord(e->ev[EV_REPAIR], e->stream));
    if (!(staged && e->rounds == 0)) {   // (staged and nothing repaired: the classes' stages have written parents and gap samples)
        if (staged) hipLaunchKernelGGL(k_reset_stats, dim3((unsigned)((P.n_units + 255) / 256)), dim3(256), 0, e->stream, P);
        hipLaunchKernelGGL(k_finalize, tiles, tb, 0, e->stream, P, all_tiles_host(e));
        if (pass == 1) hipLaunchKernelGGL(k_gaps, tiles, tb, 0, e->stream, P, all_tiles_host(e));
    }
    HIPCHK(hipEventRecord(e->ev[EV_END], e->stream));
    HIPCHK(hipGetLastError());
    int32_t kerr = 0, aw[kMaxEp + 1] = {}, hn[kSelSlots * 8] = {}, sc[kMaxEp + 1] = {};
    HIPCHK(hipMemcpyAsync(&kerr, P.err, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    if (pass == 1) HIPCHK(hipMemcpyAsync(aw, P.any_wide, sizeof(aw), hipMemcpyDeviceToHost, e->stream));   // (which classes listed spans with wide windows: pass 2 lists the same spans)
    if (pass == 1) HIPCHK(hipMemcpyAsync(hn, P.heavy_next, sizeof(hn), hipMemcpyDeviceToHost, e->stream));
    if (pass == 1) HIPCHK(hipMemcpyAsync(sc, P.split_count, sizeof(sc), hipMemcpyDeviceToHost, e->stream));   // (the pass' total per class: not reset by the repair rounds)
    HIPCHK(hipStreamSynchronize(e->stream));
    if (pass == 1) for (int E = 1; E <= kMaxEp; E++) e->wide_pass1[E] = aw[E] != 0 ? 1 : 0;
    if (pass == 1) for (int E = 1; E <= kMaxEp; E++) e->split_pass1[E] = sc[E];
    // (a repair round's fill of the counter block takes the stages' counts with it: not known then)
    if (pass == 1) for (int E = 1; E <= kMaxEp; E++) e->hard_pass1[E] = (staged && e->rounds == 0) ? hn[E * 8 + kHardCount] : -1;
    float f = 0.f;
    HIPCHK(hipEventElapsedTime(&f, e->ev[EV_BEGIN], e->ev[EV_END])); e->ms[0] = f;
    HIPCHK(hipEventElapsedTime(&f, e->ev[EV_ENUM0], e->ev[EV_ENUM1])); e->ms[1] = f;
    HIPCHK(hipEventElapsedTime(&f, e->ev[EV_WIN], e->ev[EV_SEL])); e->ms[2] = f;
    HIPCHK(hipEventElapsedTime(&f, e->ev[EV_ENUM1], e->ev[EV_WIN])); e->ms[3] = f;
    HIPCHK(hipEventElapsedTime(&f, e->ev[EV_SEL], e->ev[EV_REPAIR])); e->ms[4] = f;
    HIPCHK(hipEventElapsedTime(&f, e->ev[EV_BEGIN], e->ev[EV_PARAMS])); e->ms[5] = f;
    e->host_ms[1] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
    if (kerr != 0) return fail(e, kerr, kernel_error_text(kerr));
    return TW_OK;
}

}  // namespace

namespace {
// One stable pass of the (start, end, trace order) sort: keys gathered through the current permutation, pairs sorted
// on the bits that differ at all.
template <class Src>
int load_sort_pass(tw_engine* e, const Src* src, int64_t n, unsigned bits, unsigned long long* ka, unsigned long long* kb, uint32_t** perm, uint32_t** perm_alt) {
    if (n <= 0 || bits == 0) return TW_OK;
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 8192), block = e->coop >= 64 ? 256 : e->coop;
    if (sizeof(Src) == 8) hipLaunchKernelGGL(k_load_keys64, dim3(grid), dim3(block), 0, e->stream, (const int64_t*)src, (const uint32_t*)*perm, n, ka);
    else hipLaunchKernelGGL(k_load_keys32, dim3(grid), dim3(block), 0, e->stream, (const int32_t*)src, (const uint32_t*)*perm, n, ka);
    size_t bytes = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, bytes, ka, kb, *perm, *perm_alt, (size_t)n, 0u, bits, e->stream));
    int rc = ensure_sort_tmp(e, bytes);
    if (rc != TW_OK) return rc;
    bytes = e->sort_tmp_bytes;
    HIPCHK(rocprim::radix_sort_pairs(e->sort_tmp, bytes, ka, kb, *perm, *perm_alt, (size_t)n, 0u, bits, e->stream));
    std::swap(*perm, *perm_alt);
    return TW_OK;
}

// number of low bits in which the int64 values of an array differ (64 if signs differ)
int differing_bits(tw_engine* e, const int64_t* v, int64_t n, unsigned* bits) {
    *bits = 0;
    if (n <= 0) return TW_OK;
    std::vector<uint32_t> seg = {0u, (uint32_t)n};
    uint32_t* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, sizeof(uint32_t) * 2));
    hipError_t st = hipMemcpy(d, seg.data(), sizeof(uint32_t) * 2, hipMemcpyHostToDevice);
    unsigned long long out[2] = {0, 0};
    int rc = st == hipSuccess ? key_bits(e, v, d, d + 1, 1, n, out) : TW_ERR_DEVICE;
    (void)hipFree(d);
    if (rc != TW_OK) return rc == TW_ERR_DEVICE ? fail(e, TW_ERR_DEVICE, "differing_bits: copy failed") : rc;
    *bits = std::min(64u, std::max(1u, bit_length(out[0])));
    return TW_OK;
}
}  // namespace

extern "C" int tw_scale_load(tw_engine* e, const int32_t* unit_factor, const int32_t* trace_rank, int32_t* in_perm, int32_t* out_perm, double* unit_time_scale) {
    if (e == nullptr || unit_factor == nullptr) return TW_ERR_ARG;
    if (e->state < ST_LOADED) return fail(e, TW_ERR_STATE, "tw_scale_load before tw_load_batch");
    if (e->skip_mode) return fail(e, TW_ERR_UNSUPPORTED, "load scaling of a skip-mode batch (skip mode takes integer microseconds)");
    if (e->scaled_upload) return fail(e, TW_ERR_UNSUPPORTED, "the batch was uploaded with unit_time_scale: it is load-scaled already");
    if (e->truth == nullptr) return fail(e, TW_ERR_STATE, "tw_scale_load needs the requests' own calls (tw_set_truth): helpers/transforms.py:25-29 pairs them by trace id");
    HIPCHK(hipSetDevice(e->device));
    Dev& P = e->P;
    const int64_t n_in = P.n_in_total, n_out = P.n_out_total;
    for (int u = 0; u < P.n_units; u++)
        if (unit_factor[u] < 1) return fail(e, TW_ERR_ARG, "tw_scale_load: load factors are integers >= 1 (executor.py:1089-1091)");
    int rc;
    const bool has_trace = e->n_traces > 0 && e->in_trace != nullptr;
    if (e->orig_is == nullptr) {   // first load level: keep the table as uploaded
        rc = dev_alloc(e, &e->orig_is, n_in); if (rc != TW_OK) return rc;
        rc = dev_alloc(e, &e->orig_ie, n_in); if (rc != TW_OK) return rc;
        rc = dev_alloc(e, &e->orig_os, n_out); if (rc != TW_OK) return rc;
        rc = dev_alloc(e, &e->orig_oe, n_out); if (rc != TW_OK) return rc;
        rc = dev_alloc(e, &e->orig_truth, e->n_ie); if (rc != TW_OK) return rc;
        rc = dev_alloc(e, &e->orig_trace, n_in); if (rc != TW_OK) return rc;
        HIPCHK(hipMemcpyAsync(e->orig_is, P.in_start, sizeof(int64_t) * n_in, hipMemcpyDeviceToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->orig_ie, P.in_end, sizeof(int64_t) * n_in, hipMemcpyDeviceToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->orig_os, P.out_start, sizeof(int64_t) * n_out, hipMemcpyDeviceToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->orig_oe, P.out_end, sizeof(int64_t) * n_out, hipMemcpyDeviceToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->orig_truth, e->truth, sizeof(int32_t) * e->n_ie, hipMemcpyDeviceToDevice, e->stream));
        if (has_trace) HIPCHK(hipMemcpyAsync(e->orig_trace, e->in_trace, sizeof(int32_t) * n_in, hipMemcpyDeviceToDevice, e->stream));
    }
    // scratch of this call (freed on return): binary64 / int64 images, tie ranks, row ids, sort keys and permutations
    const int64_t n_max = std::max(n_in, n_out);
    std::vector<void*> tmp;
    auto scratch = [&](size_t bytes) -> void* { void* q = nullptr; if (hipMalloc(&q, std::max<size_t>(bytes, 8)) != hipSuccess) return nullptr; tmp.push_back(q); return q; };
    struct Free { std::vector<void*>& v; ~Free() { for (void* q : v) (void)hipFree(q); } } free_tmp{tmp};
    LoadDev L{};
    L.units = P.units; L.tiles = P.tiles;
    L.is = e->orig_is; L.ie = e->orig_ie; L.os = e->orig_os; L.oe = e->orig_oe; L.truth = e->orig_truth; L.in_trace = e->orig_trace;
    L.x = (double*)scratch(8 * n_in); L.xe = (double*)scratch(8 * n_in); L.y = (double*)scratch(8 * n_out); L.ye = (double*)scratch(8 * n_out);
    L.rank_in = (int32_t*)scratch(4 * n_in); L.rank_out = (int32_t*)scratch(4 * n_out); L.row_in = (int32_t*)scratch(4 * n_in); L.row_out = (int32_t*)scratch(4 * n_out);
    L.owner_req = (int32_t*)scratch(4 * n_out); L.kmax = (int32_t*)scratch(4 * P.n_units); L.err = (int32_t*)scratch(4);
    int32_t* d_factor = (int32_t*)scratch(4 * P.n_units); int32_t* d_segbase = (int32_t*)scratch(4 * P.n_units);
    int32_t* d_rank = trace_rank != nullptr ? (int32_t*)scratch(4 * n_in) : nullptr;
    unsigned long long *ka = (unsigned long long*)scratch(8 * n_max), *kb = (unsigned long long*)scratch(8 * n_max);
    uint32_t *perm = (uint32_t*)scratch(4 * n_max), *perm_alt = (uint32_t*)scratch(4 * n_max);
    uint32_t* perm_in = (uint32_t*)scratch(4 * n_in);
    int32_t *pos_in = (int32_t*)scratch(4 * n_in), *pos_out = (int32_t*)scratch(4 * n_out);
    if (!L.x || !L.xe || !L.y || !L.ye || !L.rank_in || !L.rank_out || !L.row_in || !L.row_out || !L.owner_req || !L.kmax || !L.err || !d_factor ||
        !d_segbase || (trace_rank != nullptr && !d_rank) || !ka || !kb || !perm || !perm_alt || !perm_in || !pos_in || !pos_out)
        return fail(e, TW_ERR_DEVICE, "tw_scale_load: out of device memory");
    std::vector<int32_t> segbase_h((size_t)P.n_units);
    { int32_t acc = 0; for (int u = 0; u < P.n_units; u++) { segbase_h[(size_t)u] = acc; acc += e->units[(size_t)u].E; } }
    HIPCHK(hipMemcpyAsync(d_factor, unit_factor, sizeof(int32_t) * P.n_units, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(d_segbase, segbase_h.data(), sizeof(int32_t) * P.n_units, hipMemcpyHostToDevice, e->stream));
    if (d_rank != nullptr) HIPCHK(hipMemcpyAsync(d_rank, trace_rank, sizeof(int32_t) * n_in, hipMemcpyHostToDevice, e->stream));
    L.factor = d_factor; L.seg_base = d_segbase; L.trace_rank = d_rank;
    HIPCHK(hipMemsetAsync(L.kmax, 0, sizeof(int32_t) * P.n_units, e->stream));
    HIPCHK(hipMemsetAsync(L.err, 0, sizeof(int32_t), e->stream));
    HIPCHK(hipMemsetAsync(L.owner_req, 0xff, sizeof(int32_t) * n_out, e->stream));
    hipLaunchKernelGGL(k_load_scale_in, dim3(P.n_tiles), dim3(e->tile), 0, e->stream, L);
    hipLaunchKernelGGL(k_load_scale_out, dim3(P.n_tiles), dim3(e->tile), 0, e->stream, L);
    hipLaunchKernelGGL(k_load_to_int, dim3(P.n_tiles), dim3(e->tile), 0, e->stream, L);
    HIPCHK(hipGetLastError());
    std::vector<int32_t> kmax_h((size_t)P.n_units);
    int32_t err_h = 0;
    HIPCHK(hipMemcpyAsync(kmax_h.data(), L.kmax, sizeof(int32_t) * P.n_units, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(&err_h, L.err, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (err_h != 0)
        return fail(e, TW_ERR_ARG, "tw_scale_load: the requests' own calls are not one call per request at every endpoint (helpers/transforms.py:25-29 asserts it), "
                                   "or the scaled timestamps span too many binades for int64");
    // stable least-significant-key-first passes: trace order, end, start, list
    const unsigned grid_in = (unsigned)std::min<int64_t>((n_in + 255) / 256, 8192), grid_out = (unsigned)std::min<int64_t>((n_out + 255) / 256, 8192);
    const unsigned block = e->coop >= 64 ? 256 : e->coop;
    int64_t* d_is = const_cast<int64_t*>(P.in_start); int64_t* d_ie = const_cast<int64_t*>(P.in_end);
    int64_t* d_os = const_cast<int64_t*>(P.out_start); int64_t* d_oe = const_cast<int64_t*>(P.out_end);
    for (int side = 0; side < 2; side++) {
        const int64_t n = side == 0 ? n_in : n_out;
        const int64_t* vs = (const int64_t*)(side == 0 ? L.x : L.y); const int64_t* ve = (const int64_t*)(side == 0 ? L.xe : L.ye);
        const int32_t* rank = side == 0 ? L.rank_in : L.rank_out; const int32_t* row = side == 0 ? L.row_in : L.row_out;
        const int nrows = side == 0 ? P.n_units : e->n_seg_out;
        unsigned bs = 0, be = 0;
        rc = differing_bits(e, vs, n, &bs); if (rc != TW_OK) return rc;
        rc = differing_bits(e, ve, n, &be); if (rc != TW_OK) return rc;
        hipLaunchKernelGGL(k_load_iota, dim3(side == 0 ? grid_in : grid_out), dim3(block), 0, e->stream, perm, n);
        rc = load_sort_pass(e, rank, n, 32u, ka, kb, &perm, &perm_alt); if (rc != TW_OK) return rc;
        rc = load_sort_pass(e, ve, n, be, ka, kb, &perm, &perm_alt); if (rc != TW_OK) return rc;
        rc = load_sort_pass(e, vs, n, bs, ka, kb, &perm, &perm_alt); if (rc != TW_OK) return rc;
        rc = load_sort_pass(e, row, n, std::max(1u, bit_length((unsigned long long)std::max(nrows - 1, 1))), ka, kb, &perm, &perm_alt); if (rc != TW_OK) return rc;
        hipLaunchKernelGGL(k_load_place, dim3(side == 0 ? grid_in : grid_out), dim3(block), 0, e->stream, (const uint32_t*)perm, n, vs, ve,
                           side == 0 ? d_is : d_os, side == 0 ? d_ie : d_oe, side == 0 ? pos_in : pos_out);
        hipLaunchKernelGGL(k_load_local, dim3(side == 0 ? grid_in : grid_out), dim3(block), 0, e->stream, perm, row, (const uint32_t*)(side == 0 ? e->seg_in : e->seg_out), n);
        HIPCHK(hipGetLastError());
        int32_t* host = side == 0 ? in_perm : out_perm;
        if (host != nullptr) HIPCHK(hipMemcpyAsync(host, perm, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    hipLaunchKernelGGL(k_load_retruth, dim3(P.n_tiles), dim3(e->tile), 0, e->stream, L, (const int32_t*)pos_in, (const int32_t*)pos_out, e->truth,
                       has_trace ? e->in_trace : nullptr);
    HIPCHK(hipGetLastError());
    for (int u = 0; u < P.n_units; u++) {
        UnitDev& U = e->units[(size_t)u];
        U.tscale = std::ldexp(1.0, -kmax_h[(size_t)u]);
        U.float_time = 1;
        if (unit_time_scale != nullptr) unit_time_scale[u] = U.tscale;
    }
    HIPCHK(hipMemcpyAsync(const_cast<UnitDev*>(P.units), e->units.data(), sizeof(UnitDev) * e->units.size(), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->state = ST_LOADED; e->pass1_done = false;
    for (int E = 0; E <= kMaxEp; E++) { e->wide_pass1[E] = -1; e->hard_pass1[E] = -1; e->split_pass1[E] = -1; }
    return TW_OK;
}

extern "C" {

int tw_create(int device_id, tw_engine** out) {
    if (out == nullptr) return TW_ERR_ARG;
    *out = nullptr;
    tw_engine* e = new tw_engine();
    e->device = device_id;
    e->tile = std::min(std::max(env_int("TW_TILE", kTile), 1), kTile);
    e->tile_threads = std::min(std::max(env_int("TW_TILE_THREADS", 2), 1), 4);
    e->coop = std::min(std::max(env_int("TW_COOP_THREADS", kCoop), 1), kCoop);
    // The engine runs its endpoint-count classes and window classes on streams of their own (up to 8 + 3 beside its own).  The
    // runtime multiplexes a process' streams onto GPU_MAX_HW_QUEUES hardware queues, 4 by default: classes that share a queue run one
    // after the other (Alibaba-shape slice, eight classes: 9.5 / 11.2 ms per pass with 4 or 8 queues, 6.3 / 7.0 ms with 12 or 16;
    // three classes: the same either way).  The variable is read when the runtime initialises and belongs to the whole process
    // (torch, RCCL, other HIP libraries in it): the library does not touch it.  The applications of this repository export it
    // before anything initialises HIP (bench.py, python -m traceweaver_amd.executor); other hosts do the same, or set
    // TW_SET_HW_QUEUES=1 to let this call export it when it is still unset (effective if the engine is the process' first user of HIP).
    if (env_int("TW_SET_HW_QUEUES", 0) != 0) setenv("GPU_MAX_HW_QUEUES", "12", 0);
    hipError_t s = hipSetDevice(device_id);
    if (s == hipSuccess) s = hipStreamCreate(&e->stream);
    for (int i = 0; i < EV_COUNT && s == hipSuccess; i++) s = hipEventCreate(&e->ev[i]);
    // (Priorities for the class streams were measured: every class stream created with one -- the classes of four and more
    // endpoints highest -- takes the media shape's enumeration from 4.6 to 4.4 ms per launch set and the nodejs shape's from 0.9
    // to 1.2 ms: streams created with a priority, any, are mapped to the hardware queues differently and that shape's two classes
    // partly serialise.  Only the long classes prioritised: no change.  Plain streams.)
    // Round 6, with the classes' window / selection stages on their streams: what a deep class' tile kernel gains by going first is no
    // longer lost to a join -- its wavefront kernels and its stage then run beside the other classes' tile kernels.  TW_PRIO_MIN_E:
    // classes from this many endpoints on get the high priority (0 = plain streams everywhere).
    const int prio_min_e = env_int("TW_PRIO_MIN_E", 0);
    int prio_least = 0, prio_greatest = 0;
    if (s == hipSuccess) s = hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    // (the class streams are created when a batch first holds the class, tw_load_batch: the streams of a process share its hardware
    // queues, and a queue shared by two busy streams runs their kernels one after the other -- an engine that only ever sees three
    // classes owns seven streams, not twelve)
    e->prio_min_e = prio_min_e; e->prio_high = prio_greatest; e->prio_any = prio_greatest != prio_least;
    for (int i = 0; i <= kMaxEp + 1 && s == hipSuccess; i++) s = hipEventCreateWithFlags(&e->cls_ev[i], hipEventDisableTiming);
    if (s == hipSuccess) s = hipEventCreateWithFlags(&e->prep_ev, hipEventDisableTiming);
    for (int i = 0; i <= kMaxEp && s == hipSuccess; i++) {
        s = hipEventCreateWithFlags(&e->post_fork[i], hipEventDisableTiming);
        if (s == hipSuccess) s = hipEventCreateWithFlags(&e->tile_ev[i], hipEventDisableTiming);
        if (s == hipSuccess) s = hipEventCreateWithFlags(&e->post_done[i], hipEventDisableTiming);
        if (s == hipSuccess) s = hipEventCreateWithFlags(&e->stretch_done[i], hipEventDisableTiming);
        for (int j = 0; j < kEnumStretches && s == hipSuccess; j++) s = hipEventCreateWithFlags(&e->stretch_ev[i][j], hipEventDisableTiming);
        for (int j = 0; j < 3 && s == hipSuccess; j++) s = hipEventCreateWithFlags(&e->post_join[i][j], hipEventDisableTiming);
    }
    e->pipeline = env_int("TW_CLASS_PIPELINE", 1);
    e->tile_gate = env_int("TW_TILE_GATE", 0);
    e->enum_stretches = std::min(std::max(env_int("TW_ENUM_STRETCHES", 1), 1), kEnumStretches);
    e->heavy_grid = std::max(env_int("TW_HEAVY_GRID", 4096), 64);
    e->select_grid = std::max(env_int("TW_SELECT_GRID", 0), 0);
    e->debug_lists = env_int("TW_DEBUG_LISTS", 0);
    e->select_fork_min = (int64_t)env_int("TW_SELECT_FORK_MIN", 0);
    e->stretch_min_tiles = std::max(env_int("TW_STRETCH_MIN_TILES", 2048), 1);
    e->stage_min_tiles = env_int("TW_STAGE_MIN_TILES", 1024);
    e->tile_sub_max = std::max(env_int("TW_TILE_SUB", 8), 1);   // workgroups per tile for classes of few tiles (1 = never)
    e->lean_pool = std::min(std::max(env_int("TW_LEAN_POOL", 512), 1), 4096);
    if (s != hipSuccess) {
        fprintf(stderr, "tw_create: %s\n", hipGetErrorString(s));
        delete e;
        return TW_ERR_DEVICE;
    }
    *out = e;
    return TW_OK;
}

void tw_destroy(tw_engine* e) {
    if (e == nullptr) return;
    (void)hipSetDevice(e->device);
    free_all(e);
    if (e->arena != nullptr) (void)hipFree(e->arena);
    for (int i = 0; i < EV_COUNT; i++)
        if (e->ev[i]) (void)hipEventDestroy(e->ev[i]);
    for (int i = 0; i <= kMaxEp + 1; i++)
        if (e->cls_ev[i]) (void)hipEventDestroy(e->cls_ev[i]);
    for (int i = 1; i <= kMaxEp; i++)
        if (e->cls_stream[i]) (void)hipStreamDestroy(e->cls_stream[i]);
    for (int i = 0; i < 3; i++) {
        if (e->sel_stream[i]) (void)hipStreamDestroy(e->sel_stream[i]);
        if (e->sel_stream2[i]) (void)hipStreamDestroy(e->sel_stream2[i]);
    }
    if (e->prep_ev) (void)hipEventDestroy(e->prep_ev);
    for (int i = 0; i <= kMaxEp; i++) {
        if (e->post_fork[i]) (void)hipEventDestroy(e->post_fork[i]);
        if (e->tile_ev[i]) (void)hipEventDestroy(e->tile_ev[i]);
        if (e->post_done[i]) (void)hipEventDestroy(e->post_done[i]);
        if (e->stretch_done[i]) (void)hipEventDestroy(e->stretch_done[i]);
        for (int j = 0; j < kEnumStretches; j++)
            if (e->stretch_ev[i][j]) (void)hipEventDestroy(e->stretch_ev[i][j]);
        if (e->cls_stream2[i]) (void)hipStreamDestroy(e->cls_stream2[i]);
        for (int j = 0; j < 3; j++)
            if (e->post_join[i][j]) (void)hipEventDestroy(e->post_join[i][j]);
    }
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

const char* tw_last_error(const tw_engine* e) { return e ? e->err.c_str() : "null engine"; }

int tw_load_batch(tw_engine* e, const tw_batch* b, int spans_on_device) {
    if (e == nullptr || b == nullptr) return TW_ERR_ARG;
    HIPCHK(hipSetDevice(e->device));
    if (b->n_units <= 0) return fail(e, TW_ERR_ARG, "n_units must be positive");
    if (b->topk != TW_TOPK) return fail(e, TW_ERR_UNSUPPORTED, "topk must be 5 (traceweaver_v3.py:1109)");
    if (b->skip != nullptr && spans_on_device)   // the start-ordered views and the pool checks of a skip-mode batch are built on the host
        return fail(e, TW_ERR_UNSUPPORTED, "a skip-mode batch takes its span arrays from host memory (spans_on_device = 0)");
    if (b->batch_size <= 0 || b->batch_size_mis <= 0) return fail(e, TW_ERR_ARG, "batch sizes must be positive");
    free_all(e);
    e->fit_rng = Mt19937(e->fit_seed);   // a batch's refit does not depend on what the engine solved before
    e->fit_prepared = false; e->fit_runs_pending = false; e->fit_max_n_valid = false;
    e->units.assign((size_t)b->n_units, UnitDev{});
    e->tiles.clear();
    e->gs_off_h.assign((size_t)b->n_units, 0);
    int64_t ie = 0, gp = 0, slots = 0, gaps = 0, epi = 0, dagi = 0;
    std::vector<uint32_t> seg_in((size_t)b->n_units + 1), seg_out;
    for (int u = 0; u < b->n_units; u++) {
        UnitDev& U = e->units[(size_t)u];
        const int E = b->unit_E[u];
        const int64_t n = b->unit_in_off[u + 1] - b->unit_in_off[u];
        if (E < 1 || E > TW_MAX_EP) return fail(e, TW_ERR_UNSUPPORTED, "unit has E outside [1, TW_MAX_EP]");
        if (n < 2) return fail(e, TW_ERR_ARG, "a unit needs at least 2 incoming spans (the reference raises on max([]) for 1, traceweaver_v3.py:1119)");
        if (n > (1 << 26)) return fail(e, TW_ERR_ARG, "unit too large (more than 2^26 incoming spans: the selection work lists address a window's first span in 26 bits)");
        U.in_off = b->unit_in_off[u];
        U.tscale = b->unit_time_scale != nullptr ? b->unit_time_scale[u] : 1.0;
        U.float_time = b->unit_time_scale != nullptr ? 1 : 0;
        U.part = b->unit_part != nullptr ? (int32_t)b->unit_part[u] : 0;
        {
            int ex = 0;
            if (!(U.tscale > 0.0) || std::frexp(U.tscale, &ex) != 0.5) return fail(e, TW_ERR_ARG, "unit_time_scale must be a positive power of two");
        }
        U.n_in = (int32_t)n;
        U.E = E;
        U.ie_off = ie;
        U.nblk = (int32_t)((n + b->batch_size - 1) / b->batch_size);
        U.nslot = E * E + 2 * E;
        U.gp_off = gp;
        U.slot_off = (int32_t)slots;
        U.tile_off = (int32_t)e->tiles.size();
        U.ntile = (int32_t)((n + e->tile - 1) / e->tile);
        for (int t = 0; t < U.ntile; t++) e->tiles.push_back(TileDev{u, t * e->tile});
        seg_in[(size_t)u] = (uint32_t)U.in_off;
        const uint8_t* dag = b->dag + dagi;
        for (int k = 0; k <= E; k++) U.ep_off[k] = b->ep_off[epi + k];
        U.skip = b->skip != nullptr ? 1 : 0;
        for (int k = 0; k < E; k++) {
            const int64_t m = U.ep_off[k + 1] - U.ep_off[k];
            if (b->skip == nullptr && m != n)
                return fail(e, TW_ERR_UNSUPPORTED, "an endpoint's span count differs from the number of incoming spans: a skip-mode unit (traceweaver_v3.py:972,1155-1156) needs tw_batch.skip");
            if (b->skip != nullptr && (m > n || m < 0)) return fail(e, TW_ERR_ARG, "skip mode: an endpoint holds more spans than there are incoming spans");
            seg_out.push_back((uint32_t)U.ep_off[k]);
        }
        if (b->skip != nullptr) {
            if (b->unit_time_scale != nullptr) return fail(e, TW_ERR_UNSUPPORTED, "skip mode takes integer microseconds");
            const tw_skip_unit& K = b->skip[u];
            if (K.n_tw < 1 || !K.tw_start || !K.pool || !K.dist) return fail(e, TW_ERR_ARG, "skip mode: incomplete tw_skip_unit");
            if (K.n_tw > (0x7fffffff - TW_SKIP_BASE) / TW_SKIP_STRIDE) return fail(e, TW_ERR_ARG, "skip mode: too many time windows");
            for (int64_t q = 0; q < (int64_t)E * K.n_tw; q++)
                if (K.pool[q] < 0 || K.pool[q] > TW_SKIP_STRIDE) return fail(e, TW_ERR_ARG, "skip mode: a pool holds more than TW_SKIP_STRIDE skip spans");
        }
        for (int q = 0; q < E; q++) {
            uint8_t pm = 0, sm = 0;
            int np = 0;
            for (int p = 0; p < E; p++) {
                if (dag[p * E + q]) {
                    if (p >= q) return fail(e, TW_ERR_ARG, "endpoints are not in topological order of the DAG");
                    pm |= (uint8_t)(1u << p);
                    U.pred_list[q][np++] = (uint8_t)p;
                }
                if (dag[q * E + p]) sm |= (uint8_t)(1u << p);
            }
            U.pred_mask[q] = pm;
            U.key_rank[q] = (uint8_t)b->key_rank[epi + q];
            U.succ_mask[q] = sm;
            U.npred[q] = (uint8_t)np;
            // networkx in_edges(): predecessors in partition-key (insertion) order, executor.py:223-236
            std::stable_sort(U.pred_list[q], U.pred_list[q] + np, [&](uint8_t x, uint8_t y) {
                return b->key_rank[epi + x] < b->key_rank[epi + y];
            });
            for (int j = 0; j < np; j++) {  // primary <=> no 2-hop alternative (traceweaver_v1.py:245-254)
                const int p = U.pred_list[q][j];
                bool prim = true;
                for (int m = 0; m < E; m++)
                    if (m != p && m != q && dag[p * E + m] && dag[m * E + q]) prim = false;
                U.pred_prim[q][j] = prim ? 1 : 0;
            }
        }
        e->gs_off_h[(size_t)u] = gaps;
        ie += n * E;
        gp += (int64_t)U.nblk * U.nslot;
        slots += U.nslot;
        gaps += (int64_t)U.nslot * n;
        epi += E;
        dagi += E * E;
    }
    std::vector<int32_t> tile_ids_h;
    int32_t heavy_off_h[kMaxEp + 2] = {};
    for (int cls = 0; cls <= kMaxEp; cls++) {
        e->tile_cls_off[cls] = (int32_t)tile_ids_h.size();
        heavy_off_h[cls + 1] = heavy_off_h[cls];
        for (size_t tix = 0; tix < e->tiles.size(); tix++)
            if (e->units[(size_t)e->tiles[tix].unit].E == cls) tile_ids_h.push_back((int32_t)tix);
        for (int u = 0; u < b->n_units; u++)
            if (e->units[(size_t)u].E == cls) heavy_off_h[cls + 1] += e->units[(size_t)u].n_in;
    }
    e->tile_cls_off[kMaxEp + 1] = (int32_t)tile_ids_h.size();
    for (int cls = 1; cls <= kMaxEp; cls++)
        if (e->tile_cls_off[cls + 1] > e->tile_cls_off[cls]) {
            int rs = ensure_class_stream(e, cls); if (rs != TW_OK) return rs;
            if (e->enum_stretches > 1 && e->tile_cls_off[cls + 1] - e->tile_cls_off[cls] >= 2 * e->stretch_min_tiles && e->cls_stream2[cls] == nullptr)
                HIPCHK(hipStreamCreate(&e->cls_stream2[cls]));
        }
    for (int j = 0; j < 3; j++)
        if (e->sel_stream[j] == nullptr) HIPCHK(hipStreamCreate(&e->sel_stream[j]));
    {   // (the second set only for a batch that has such a class: every stream takes a share of the hardware queues)
        bool want = false;
        int deepest = 0;
        for (int cls = 1; cls <= kMaxEp; cls++) if (e->tile_cls_off[cls + 1] > e->tile_cls_off[cls]) deepest = cls;
        for (int cls = 1; cls < deepest; cls++) want |= e->select_fork_min > 0 && heavy_off_h[cls + 1] - heavy_off_h[cls] >= e->select_fork_min;
        for (int j = 0; j < 3 && want; j++)
            if (e->sel_stream2[j] == nullptr) HIPCHK(hipStreamCreate(&e->sel_stream2[j]));
    }
    const int64_t n_in_total = b->unit_in_off[b->n_units], n_out_total = b->ep_off[epi];
    if (n_in_total >= (1ll << 31) || n_out_total >= (1ll << 31)) return fail(e, TW_ERR_ARG, "batch exceeds 2^31 spans");
    if (gaps >= (1ll << 31)) return fail(e, TW_ERR_ARG, "batch too large: sum of nslot*n_in must stay below 2^31");
    std::vector<int32_t> slot_unit_h;
    std::vector<uint8_t> slot_scored_h;
    std::vector<uint32_t> seg_gap_h, seg_gap_end_h;  // gap rows of the scored slots only (the others stay NaN and are never fitted)
    for (int u = 0; u < b->n_units; u++) {
        const UnitDev& U = e->units[(size_t)u];
        const int E = U.E;
        for (int q = 0; q < U.nslot; q++) {
            bool scored;
            if (q < E) scored = U.npred[q] == 0;
            else if (q < E + E * E) {
                const int p = (q - E) / E, en = (q - E) % E;
                scored = false;
                for (int j = 0; j < U.npred[en]; j++) scored |= (U.pred_list[en][j] == p && U.pred_prim[en][j]);
            } else scored = true;
            slot_unit_h.push_back(u);
            slot_scored_h.push_back(scored ? 1 : 0);
            if (scored) {
                seg_gap_h.push_back((uint32_t)(e->gs_off_h[(size_t)u] + (int64_t)q * U.n_in));
                seg_gap_end_h.push_back((uint32_t)(e->gs_off_h[(size_t)u] + (int64_t)(q + 1) * U.n_in));
            }
        }
    }
    e->n_gap_rows = (int64_t)seg_gap_h.size();
    std::vector<uint32_t> seg_gap_dst_h(seg_gap_h.size());
    e->n_gap_scored = 0;
    for (size_t r = 0; r < seg_gap_h.size(); r++) { seg_gap_dst_h[r] = (uint32_t)e->n_gap_scored; e->n_gap_scored += seg_gap_end_h[r] - seg_gap_h[r]; }
    seg_in[(size_t)b->n_units] = (uint32_t)n_in_total;
    seg_out.push_back((uint32_t)n_out_total);
    e->n_seg_out = (int)seg_out.size() - 1;
    e->n_ie = ie; e->n_gp = gp; e->n_slots = slots; e->n_gaps = gaps;

    Dev& P = e->P;
    P = Dev{};
    P.n_units = b->n_units;
    P.n_tiles = (int32_t)e->tiles.size();
    P.tile_spans = e->tile;
    P.n_in_total = n_in_total;
    P.n_out_total = n_out_total;
    P.batch_size = b->batch_size;
    P.batch_mis = b->batch_size_mis;
    P.split_twins = env_int("TW_SPLIT_TWINS", 2);
    P.defer_min_e = env_int("TW_DEFER_MIN_E", 5);
    P.lean_min_e = env_int("TW_LEAN_MIN_E", 5);
    P.lean_grid = std::max(env_int("TW_LEAN_GRID", TW_LEAN_GRID), 1);
    P.tile_max_deep = std::min(std::max(env_int("TW_TILE_MAX_DEEP", kTileMax), 0), kTileMax);
    int rc;
#define ALLOC(ptr, count) do { rc = dev_alloc(e, &(ptr), (count)); if (rc != TW_OK) return rc; } while (0)
    e->arena_req.clear();
    e->arena_open = true;
    UnitDev* d_units; TileDev* d_tiles; int64_t *d_is, *d_ie, *d_os, *d_oe, *d_gs;
    ALLOC(d_units, P.n_units); ALLOC(d_tiles, P.n_tiles);
    ALLOC(d_is, n_in_total); ALLOC(d_ie, n_in_total); ALLOC(d_os, n_out_total); ALLOC(d_oe, n_out_total);
    ALLOC(d_gs, P.n_units);
    ALLOC(P.in_end_sorted, n_in_total); ALLOC(P.out_end_sorted, n_out_total);
    ALLOC(P.gparam, gp * 4);
    ALLOC(e->mix_n_dev, slots); ALLOC(e->mix_p_dev, slots * kMaxComp * 3); ALLOC(e->mix_c_dev, slots * kMaxComp * 4);
    ALLOC(P.pm_val, n_in_total); ALLOC(P.pm_idx, n_in_total); ALLOC(P.pc, n_in_total + 1); ALLOC(P.seg, n_in_total);
    ALLOC(P.win_end, n_in_total); ALLOC(P.wid, n_in_total); ALLOC(P.w_last, n_in_total);
    ALLOC(P.unit_nwin, P.n_units); ALLOC(P.w_dirty, n_in_total); ALLOC(P.w_conf, n_in_total);
    ALLOC(P.tk_n, n_in_total); ALLOC(P.leaves, n_in_total); ALLOC(P.leaves0, n_in_total); ALLOC(P.chosen, n_in_total); ALLOC(P.rep, n_in_total);
    ALLOC(P.tkr_n, n_in_total);
    ALLOC(P.tk_idx, ie * kTopK); ALLOC(P.tkr_idx, ie * kTopK);
    ALLOC(P.tk_score, n_in_total * kTopK); ALLOC(P.tkr_score, n_in_total * kTopK);
    ALLOC(P.c_lo, ie); ALLOC(P.c_hi, ie); ALLOC(P.c_bits, ie * kCandWords); ALLOC(P.parent, ie);
    ALLOC(P.gone, ie * kCandWords); ALLOC(P.gone_valid, n_in_total); ALLOC(P.leaves_r, n_in_total);
    ALLOC(P.frontier, (int64_t)kFrontierSlots * 2 * kFrontierCap); 
    // long tuple lists: a pool that grows with the batch (one list per 16 k incoming spans, 48 ... 512 of 32 MB each; recycled)
    P.frontier_big_slots = (int32_t)std::min<int64_t>(std::max<int64_t>(n_in_total / 16384, kFrontierBigSlots), std::max(kFrontierBigSlots, 512));
    ALLOC(P.frontier_big, (int64_t)P.frontier_big_slots * 2 * kFrontierBigCap);
    ALLOC(P.pair_pool, (int64_t)kPairSlots * kPairSpill);
    ALLOC(P.owner, n_out_total);
    ALLOC(P.gaps, gaps);
    ALLOC(e->tile_ids, (int64_t)tile_ids_h.size());
    ALLOC(P.heavy_in_unit, n_in_total); ALLOC(P.heavy_in_idx, n_in_total);
    {   // long enumerations: a list entry per span plus the extra entries of the split ones (an eighth of the class + 64), two
        // scratch slots per extra entry
        int64_t big_total = 0, slots = 0, want = 0;
        for (int cls = 0; cls <= kMaxEp; cls++) want += 2 * part_extra(cls, heavy_off_h[cls + 1] - heavy_off_h[cls]);
        // (a scratch slot is 4.4 KB, most of it the log of a part in log mode: at most TW_PART_SLOTS_MAX slots, 16 M = 70 GB, whatever the batch)
        static const int64_t slots_max = (int64_t)std::max(env_int("TW_PART_SLOTS_MAX", 16 << 20), 1024);
        for (int cls = 0; cls <= kMaxEp; cls++) {
            const int64_t n_cls = heavy_off_h[cls + 1] - heavy_off_h[cls];
            int64_t extra = part_extra(cls, n_cls);
            if (want > slots_max && extra > 64) extra = std::max<int64_t>((int64_t)((double)extra * (double)slots_max / (double)want), std::min<int64_t>(n_cls / 8 + 64, extra));
            P.heavy_big_off[cls] = (int32_t)big_total; P.part_off[cls] = (int32_t)slots;
            big_total += n_cls + extra; slots += 2 * extra;
        }
        P.heavy_big_off[kMaxEp + 1] = (int32_t)big_total; P.part_off[kMaxEp + 1] = (int32_t)slots;
        ALLOC(P.heavy_big_unit, big_total); ALLOC(P.heavy_big_idx, big_total);
        ALLOC(P.heavy_big_part, big_total); ALLOC(P.heavy_big_slot, big_total);
        ALLOC(P.fb_unit, big_total); ALLOC(P.fb_idx, big_total); ALLOC(P.fb_part, big_total); ALLOC(P.fb_slot, big_total);
        ALLOC(P.split_unit, slots); ALLOC(P.split_idx, slots); ALLOC(P.split_slot, slots); ALLOC(P.split_parts, slots);
        ALLOC(P.part_n, slots); ALLOC(P.part_leaves, slots); ALLOC(P.part_score, slots * kTopK); ALLOC(P.part_idx, slots * kTopK * kMaxEp);
        ALLOC(P.part_bits, slots * kMaxEp * kCandWords);
        ALLOC(P.part_logn, slots); ALLOC(P.part_log_sc, slots * kPartLogCap); ALLOC(P.part_log_ix, slots * kPartLogCap);
        ALLOC(P.part_lo, slots); ALLOC(P.part_hi, slots); ALLOC(P.part_lvl, slots);
        // the arena of the deferred spans' prefix lists: a few thousand entries per span that defers, recycled every pass
        int64_t deep = 0;
        for (int cls = std::max(P.defer_min_e, 3); cls <= kMaxEp; cls++) deep += heavy_off_h[cls + 1] - heavy_off_h[cls];
        P.defer_cap = (int32_t)(deep > 0 ? std::min<int64_t>(std::max<int64_t>(deep * kDeferListPerSpan, kDeferListMin), 1ll << 28) : 1);
        ALLOC(P.defer_list, P.defer_cap);
    }
    for (int cls = 0; cls <= kMaxEp + 1; cls++) P.heavy_in_off[cls] = heavy_off_h[cls];
    ALLOC(P.prof, 32); ALLOC(e->key_acc, 2);
    const int64_t sel_cap = (int64_t)P.n_tiles * e->tile + 1;   // every segment of the selection lists has room for all windows of its tiles
    {   // the counter block (every array on a 128-byte line of its own: their atomics come from different kernels)
        int64_t at = 0;
        auto take = [&](int64_t ints) { const int64_t o = at; at += (ints + 31) / 32 * 32; return o; };
        const int64_t o_hc = take(kSelSlots * 4 * kSelSeg * kCtrStride), o_hn = take(kSelSlots * 8), o_ic = take(2 * (kMaxEp + 1)), o_in = take(8 * (kMaxEp + 1)), o_bc = take(kMaxEp + 1),
                      o_rc = take(1), o_rd = take(kMaxEp + 1), o_fc = take(kMaxEp + 1);
        e->ctr_round_ints = at;
        const int64_t o_pu = take(kMaxEp + 1), o_sc = take(kMaxEp + 1), o_dc = take(kMaxEp + 1), o_du = take(1), o_err = take(1), o_nd = take(P.n_units), o_us = take((int64_t)P.n_units * 16);
        const int64_t o_dr = take(kMaxEp + 1), o_ft = take(kMaxEp + 1), o_aw = take(kMaxEp + 1);
        const int64_t o_es = take((kMaxEp + 1) * (kEnumStretches + 1) * 4), o_en = take((kMaxEp + 1) * kEnumStretches * 2);
        const int64_t o_fn = take(kFrontierSlots), o_fb = take(P.frontier_big_slots), o_pb = take(kPairSlots);   // flags of the tuple-list pools (given back by the kernels themselves)
        e->ctr_pass_ints = at;
        auto place = [=](void* q) {
            Dev& D = e->P;
            int32_t* c = e->ctr = (int32_t*)q;
            D.heavy_count = c + o_hc; D.heavy_next = c + o_hn; D.heavy_in_count = c + o_ic; D.heavy_in_next = c + o_in; D.heavy_big_count = c + o_bc;
            D.round_changed = c + o_rc; D.frontier_busy = c + o_fn; D.frontier_big_busy = c + o_fb;
            D.defer_count = c + o_dc; D.defer_used = c + o_du;
            D.part_used = c + o_pu; D.split_count = c + o_sc; D.err = c + o_err; D.unit_ndirty = c + o_nd;
            D.redo_count = c + o_rd; D.fb_count = c + o_fc; D.defer_refused = c + o_dr; D.fb_total = c + o_ft; D.pair_busy = c + o_pb; D.any_wide = c + o_aw;
            D.enum_snap = c + o_es; D.enum_stretch_next = c + o_en;
            D.unit_stats = (int64_t*)(c + o_us);
        };
        if (e->arena_open) e->arena_req.emplace_back(place, ((size_t)at * sizeof(int32_t) + 255) / 256 * 256);   // (placed by arena_commit, like the rest)
        else { ALLOC(e->ctr, at); place(e->ctr); }
    }
    ALLOC(P.heavy_unit, sel_cap); ALLOC(P.heavy_win, sel_cap);
    ALLOC(P.tiny_unit, sel_cap); ALLOC(P.tiny_win, sel_cap);
    ALLOC(P.rheavy_unit, sel_cap); ALLOC(P.rheavy_win, sel_cap); ALLOC(P.rtiny_unit, sel_cap); ALLOC(P.rtiny_win, sel_cap);
    ALLOC(P.hard_unit, sel_cap / (kBruteMax + 1) + 1 + kSelSlots); ALLOC(P.hard_win, sel_cap / (kBruteMax + 1) + 1 + kSelSlots);   // (a searched window holds more than kBruteMax spans)
    ALLOC(e->rank_in, n_in_total); ALLOC(e->rank_out, n_out_total);
    ALLOC(e->agg_pair, P.n_tiles); ALLOC(e->agg_i32, P.n_tiles); ALLOC(e->agg_i32b, P.n_tiles);
    ALLOC(e->seg_in, (int64_t)seg_in.size()); ALLOC(e->seg_out, (int64_t)seg_out.size());
    ALLOC(e->gaps_sorted, gaps); ALLOC(e->fit_models, slots * kMaxComp * kModelStride);
    ALLOC(e->fit_uval, gaps); ALLOC(e->fit_ustart, gaps); ALLOC(e->fit_row_n, slots); ALLOC(e->fit_row_uniq, slots);
    ALLOC(e->fit_tape, slots * kFitRowTape); ALLOC(e->fit_tape_off, slots); ALLOC(e->fit_tape100, kMaxComp * 13);
    e->fit_tape_cap = slots * kFitRowTape;
    ALLOC(e->fit_centres, slots * kMaxComp * (kMaxComp + 1));
    ALLOC(e->slot_unit, slots); ALLOC(e->slot_scored, slots);
    ALLOC(e->seg_gap, (int64_t)seg_gap_h.size()); ALLOC(e->seg_gap_end, (int64_t)seg_gap_end_h.size()); ALLOC(e->seg_gap_dst, (int64_t)seg_gap_dst_h.size());
    e->comp_cap = std::max(std::max(n_in_total, n_out_total), e->n_gap_scored);
    ALLOC(e->comp_a, e->comp_cap); ALLOC(e->comp_b, e->comp_cap);
    // skip mode: per unit the start-ordered view of every endpoint list (TallySkipSpans sorts the lists in place,
    // traceweaver_v3.py:968-971; the top_k_2 enumeration walks that order), time windows, pools, (mean, std) table, draw counters
    e->skip_mode = b->skip != nullptr;
    std::vector<int32_t> perm_h, pool_h;
    std::vector<int64_t> tw_h;
    std::vector<double> dist_h;
    std::vector<SkipUnitDev> skip_h;
    std::vector<int64_t> sk_tw_off, sk_pool_off, sk_dist_off;
    if (e->skip_mode) {
        perm_h.resize((size_t)n_out_total);
        for (int u = 0; u < b->n_units; u++) {
            const UnitDev& U = e->units[(size_t)u];
            const tw_skip_unit& K = b->skip[u];
            for (int k = 0; k < U.E; k++) {
                const int64_t a = U.ep_off[k], m = U.ep_off[k + 1] - a;
                std::vector<int32_t> idx((size_t)m);
                for (int64_t q = 0; q < m; q++) idx[(size_t)q] = (int32_t)q;
                std::stable_sort(idx.begin(), idx.end(), [&](int32_t x, int32_t y) { return b->out_start[a + x] < b->out_start[a + y]; });
                std::copy(idx.begin(), idx.end(), perm_h.begin() + a);
            }
            sk_tw_off.push_back((int64_t)tw_h.size()); sk_pool_off.push_back((int64_t)pool_h.size()); sk_dist_off.push_back((int64_t)dist_h.size());
            tw_h.insert(tw_h.end(), K.tw_start, K.tw_start + K.n_tw);
            pool_h.insert(pool_h.end(), K.pool, K.pool + (int64_t)U.E * K.n_tw);
            dist_h.insert(dist_h.end(), K.dist, K.dist + (int64_t)(U.E + 1) * (U.E + 1) * 2);
        }
        e->skip_fetch_n = (int64_t)pool_h.size();
        ALLOC(e->skip_units, b->n_units); ALLOC(e->skip_perm, n_out_total); ALLOC(e->skip_tw, (int64_t)tw_h.size());
        ALLOC(e->skip_pool, (int64_t)pool_h.size()); ALLOC(e->skip_dist, (int64_t)dist_h.size()); ALLOC(e->skip_fetch, e->skip_fetch_n);
    }
#undef ALLOC
    rc = arena_commit(e);
    if (rc != TW_OK) return rc;
    if (e->skip_mode) {
        for (int u = 0; u < b->n_units; u++) {
            SkipUnitDev K{};
            K.perm = e->skip_perm; K.tw_start = e->skip_tw + sk_tw_off[(size_t)u]; K.pool = e->skip_pool + sk_pool_off[(size_t)u];
            K.dist = e->skip_dist + sk_dist_off[(size_t)u]; K.fetches = e->skip_fetch + sk_pool_off[(size_t)u]; K.n_tw = b->skip[u].n_tw;
            skip_h.push_back(K);
        }
        HIPCHK(hipMemcpyAsync(e->skip_units, skip_h.data(), sizeof(SkipUnitDev) * skip_h.size(), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->skip_perm, perm_h.data(), sizeof(int32_t) * perm_h.size(), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->skip_tw, tw_h.data(), sizeof(int64_t) * tw_h.size(), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->skip_pool, pool_h.data(), sizeof(int32_t) * pool_h.size(), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->skip_dist, dist_h.data(), sizeof(double) * dist_h.size(), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));   // the staging vectors go out of scope
    }
    HIPCHK(hipMemsetAsync(P.prof, 0, sizeof(unsigned long long) * 32, e->stream));
    HIPCHK(hipMemsetAsync(P.prof + 10, 0xff, sizeof(unsigned long long), e->stream));
    P.units = d_units; P.tiles = d_tiles;
    P.in_start = d_is; P.in_end = d_ie; P.out_start = d_os; P.out_end = d_oe;
    P.gs_off = d_gs;
    P.mix_n = e->mix_n_dev; P.mix_c = e->mix_c_dev;
    e->sort_tmp = nullptr; e->sort_tmp_bytes = 0;
    const hipMemcpyKind kind = spans_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    HIPCHK(hipMemcpyAsync(d_is, b->in_start, sizeof(int64_t) * n_in_total, kind, e->stream));
    HIPCHK(hipMemcpyAsync(d_ie, b->in_end, sizeof(int64_t) * n_in_total, kind, e->stream));
    HIPCHK(hipMemcpyAsync(d_os, b->out_start, sizeof(int64_t) * n_out_total, kind, e->stream));
    HIPCHK(hipMemcpyAsync(d_oe, b->out_end, sizeof(int64_t) * n_out_total, kind, e->stream));
    HIPCHK(hipMemcpyAsync(d_units, e->units.data(), sizeof(UnitDev) * e->units.size(), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(d_tiles, e->tiles.data(), sizeof(TileDev) * e->tiles.size(), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(d_gs, e->gs_off_h.data(), sizeof(int64_t) * e->gs_off_h.size(), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->seg_in, seg_in.data(), sizeof(uint32_t) * seg_in.size(), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->seg_out, seg_out.data(), sizeof(uint32_t) * seg_out.size(), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->tile_ids, tile_ids_h.data(), sizeof(int32_t) * tile_ids_h.size(), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->slot_unit, slot_unit_h.data(), sizeof(int32_t) * slot_unit_h.size(), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->seg_gap, seg_gap_h.data(), sizeof(uint32_t) * seg_gap_h.size(), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->seg_gap_end, seg_gap_end_h.data(), sizeof(uint32_t) * seg_gap_end_h.size(), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->seg_gap_dst, seg_gap_dst_h.data(), sizeof(uint32_t) * seg_gap_dst_h.size(), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->slot_scored, slot_scored_h.data(), slot_scored_h.size(), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemsetAsync(P.gaps, 0xff, sizeof(double) * std::max<int64_t>(gaps, 1), e->stream));  // all-ones = NaN: rows of unscored slots
    HIPCHK(hipMemsetAsync(P.pc, 0, (size_t)n_in_total + 1, e->stream));
    // top-5 lists: all-ones = index -1 / score NaN; the enumeration kernels write only the entries a span has
    HIPCHK(hipMemsetAsync(P.tk_idx, 0xff, sizeof(int32_t) * (size_t)std::max<int64_t>(ie * kTopK, 1), e->stream));
    HIPCHK(hipMemsetAsync(P.tk_score, 0xff, sizeof(double) * (size_t)std::max<int64_t>(n_in_total * kTopK, 1), e->stream));
    HIPCHK(hipMemsetAsync(e->mix_n_dev, 0, sizeof(int32_t) * std::max<int64_t>(slots, 1), e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->scaled_upload = b->unit_time_scale != nullptr;
    e->state = ST_LOADED; e->pass1_done = false;
    for (int E = 0; E <= kMaxEp; E++) { e->wide_pass1[E] = -1; e->hard_pass1[E] = -1; e->split_pass1[E] = -1; }
    return TW_OK;
}

int tw_run_pass1(tw_engine* e) {
    if (e == nullptr) return TW_ERR_ARG;
    if (e->state < ST_LOADED) return fail(e, TW_ERR_STATE, "tw_run_pass1 before tw_load_batch");
    HIPCHK(hipSetDevice(e->device));
    const int rc = run_pass(e, 1);
    e->fit_prepared = false; e->fit_runs_pending = false; e->fit_max_n_valid = false;
    if (rc == TW_OK) { e->state = ST_PASS1; e->pass1_done = true; }
    return rc;
}

int tw_get_gaps(tw_engine* e, double* gaps) {
    if (e == nullptr || gaps == nullptr) return TW_ERR_ARG;
    if (e->state != ST_PASS1) return fail(e, TW_ERR_STATE, "tw_get_gaps is valid right after tw_run_pass1");
    if (e->skip_mode) return fail(e, TW_ERR_STATE, "a skip-mode batch runs one pass (traceweaver_v3.py:1155-1156): there is no refit");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(gaps, e->P.gaps, sizeof(double) * e->n_gaps, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return TW_OK;
}

int tw_set_gaps(tw_engine* e, const double* gaps) {
    if (e == nullptr || gaps == nullptr) return TW_ERR_ARG;
    if (e->state < ST_LOADED) return fail(e, TW_ERR_STATE, "tw_set_gaps before tw_load_batch");
    if (e->skip_mode) return fail(e, TW_ERR_STATE, "a skip-mode batch runs one pass (traceweaver_v3.py:1155-1156): there is no refit");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(e->P.gaps, gaps, sizeof(double) * e->n_gaps, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->fit_prepared = false; e->fit_runs_pending = false; e->fit_max_n_valid = false;
    e->state = ST_PASS1;
    return TW_OK;
}

int tw_device_buffers(tw_engine* e, tw_device_view* out) {
    if (e == nullptr || out == nullptr) return TW_ERR_ARG;
    if (e->state < ST_LOADED) return fail(e, TW_ERR_STATE, "tw_device_buffers before tw_load_batch");
    out->parent = e->P.parent; out->parent_count = e->n_ie;
    out->gaps = e->P.gaps; out->gaps_count = e->n_gaps;
    out->device = e->device;
    return TW_OK;
}

int tw_set_gaps_device(tw_engine* e, const double* dev_gaps) {
    if (e == nullptr || dev_gaps == nullptr) return TW_ERR_ARG;
    if (e->state < ST_LOADED) return fail(e, TW_ERR_STATE, "tw_set_gaps_device before tw_load_batch");
    if (e->skip_mode) return fail(e, TW_ERR_STATE, "a skip-mode batch runs one pass (traceweaver_v3.py:1155-1156): there is no refit");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(e->P.gaps, dev_gaps, sizeof(double) * e->n_gaps, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->fit_prepared = false; e->fit_runs_pending = false; e->fit_max_n_valid = false;
    e->state = ST_PASS1;
    return TW_OK;
}

int tw_set_mixtures(tw_engine* e, const int32_t* mix_n, const double* mix_p) {
    if (e == nullptr || mix_n == nullptr || mix_p == nullptr) return TW_ERR_ARG;
    if (e->state < ST_PASS1) return fail(e, TW_ERR_STATE, "tw_set_mixtures before tw_run_pass1");
    if (e->skip_mode) return fail(e, TW_ERR_STATE, "a skip-mode batch runs one pass (traceweaver_v3.py:1155-1156): there is no second pass");
    HIPCHK(hipSetDevice(e->device));
    for (int64_t q = 0; q < e->n_slots; q++)
        if (mix_n[q] < 0 || mix_n[q] > TW_MAX_COMP) return fail(e, TW_ERR_ARG, "mixture component count outside [0, TW_MAX_COMP]");
    HIPCHK(hipMemcpyAsync(e->mix_n_dev, mix_n, sizeof(int32_t) * e->n_slots, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->mix_p_dev, mix_p, sizeof(double) * e->n_slots * kMaxComp * 3, hipMemcpyHostToDevice, e->stream));
    const int64_t total = e->n_slots * kMaxComp;
    hipLaunchKernelGGL(k_mix_consts, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, e->stream, (const double*)e->mix_p_dev, (const int32_t*)e->mix_n_dev, (const int32_t*)e->slot_unit, e->P.units, e->mix_c_dev, total);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    e->state = ST_MIX;
    return TW_OK;
}

namespace {

FitDev fit_dev(tw_engine* e) {
    FitDev F{};
    F.units = e->P.units; F.n_units = e->P.n_units; F.n_slots = e->n_slots; F.gaps = e->P.gaps; F.sorted = e->gaps_sorted;
    F.gs_off = e->P.gs_off; F.slot_unit = e->slot_unit; F.slot_scored = e->slot_scored; F.models = e->fit_models; F.mix_n = e->mix_n_dev; F.mix_p = e->mix_p_dev;
    F.uval = e->fit_uval; F.ustart = e->fit_ustart; F.row_n = e->fit_row_n; F.row_uniq = e->fit_row_uniq;
    F.tape = e->fit_tape; F.tape_off = e->fit_tape_off; F.tape100 = e->fit_tape100; F.err = e->P.err; F.centres = e->fit_centres;
    static const int by_row = env_int("TW_FIT_BY_ROW", 0);
    F.by_row = by_row;
    return F;
}

// Turns every scored gap row into (value, multiplicity) runs; idempotent per pass-1 result.  Default route: k_fit_runs (a hash
// table per row in LDS, the row read once); rows it cannot take -- and TW_FIT_SORT=1 -- go the sort route below: composite-key
// radix sort of all rows + k_fit_compress.  Both leave the same arrays (uval, ustart, row_n, row_uniq).
// (in two halves: fit_prepare_launch queues k_fit_runs and returns -- the caller has host work of its own to do meanwhile, the
// uniforms of the k-means++ draws -- and fit_prepare waits for it, reads whether a row was refused and runs the sort route then)
int fit_prepare_launch(tw_engine* e) {
    e->fit_runs_pending = false;
    if (e->fit_prepared || env_int("TW_FIT_SORT", 0) != 0) return TW_OK;
    int32_t* flag = (int32_t*)e->key_acc;
    HIPCHK(hipMemsetAsync(flag, 0, sizeof(int32_t), e->stream));
#ifdef TW_HOST_EMULATION
    const int runs_threads = e->coop;
#else
    const int runs_threads = e->coop >= 64 ? kRunsThreads : e->coop;
#endif
    hipLaunchKernelGGL(k_fit_runs, dim3((unsigned)e->n_slots), dim3(runs_threads), 0, e->stream, fit_dev(e), flag);
    HIPCHK(hipGetLastError());
    e->fit_runs_pending = true;
    return TW_OK;
}
int fit_prepare(tw_engine* e) {
    if (e->fit_prepared) return TW_OK;
    if (env_int("TW_FIT_SORT", 0) == 0) {
        if (!e->fit_runs_pending) { int rl = fit_prepare_launch(e); if (rl != TW_OK) return rl; }
        e->fit_runs_pending = false;
        int32_t* flag = (int32_t*)e->key_acc;
        int32_t refused = 0;
        HIPCHK(hipMemcpyAsync(&refused, flag, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        if (!refused) {
            e->fit_prepared = true;
            e->fit_max_n_valid = false;
            return TW_OK;
        }
    }
    size_t bytes = 0;
    const unsigned size = (unsigned)e->n_gaps, nseg = (unsigned)e->n_gap_rows;
    // gap samples are non-negative integers (and NaN = 0x7ff8...0) stored as doubles: their low mantissa bits
    // are all zero, the sort starts at the lowest bit set anywhere.  A negative key (none can occur: children
    // lie inside their parent and follow their predecessors) would flip rocprim's key transform -> full width.
    unsigned long long kb[2];
    int rck = key_bits(e, e->P.gaps, e->seg_gap, e->seg_gap_end, (int)nseg, e->n_gaps, kb);
    if (rck != TW_OK) return rck;
    unsigned begin_bit = 0;
    if (kb[1] != 0 && !(kb[1] >> 63)) while (!((kb[1] >> begin_bit) & 1)) begin_bit++;
    (void)bytes;
    if (kb[1] >> 63) {  // a negative sample: leave the key transform to rocprim (never seen; see above)
        HIPCHK(rocprim::segmented_radix_sort_keys(nullptr, bytes, (const double*)e->P.gaps, e->gaps_sorted, size, nseg, e->seg_gap, e->seg_gap_end, 0, 64, e->stream));
        int rcs = ensure_sort_tmp(e, bytes);
        if (rcs != TW_OK) return rcs;
        bytes = e->sort_tmp_bytes;
        HIPCHK(rocprim::segmented_radix_sort_keys(e->sort_tmp, bytes, (const double*)e->P.gaps, e->gaps_sorted, size, nseg, e->seg_gap, e->seg_gap_end, 0, 64, e->stream));
    } else {  // non-negative doubles order like their bit patterns
        int rcs = sort_rows(e, (const unsigned long long*)e->P.gaps, (unsigned long long*)e->gaps_sorted, size, e->seg_gap, e->seg_gap_end,
                            e->seg_gap_dst, (int)nseg, e->n_gap_scored, begin_bit, 64, 0ull);
        if (rcs != TW_OK) return rcs;
    }
    FitDev F = fit_dev(e);
#ifdef TW_HOST_EMULATION   // (the lane-threaded emulation of the tests runs a host thread per lane: the cooperative size there)
    const int compress_threads = e->coop;
#else
    const int compress_threads = e->coop >= 64 ? kCompressThreads : e->coop;
#endif
    hipLaunchKernelGGL(k_fit_compress, dim3((unsigned)e->n_slots), dim3(compress_threads), 0, e->stream, F);
    HIPCHK(hipGetLastError());
    e->fit_prepared = true;
    e->fit_max_n_valid = false;   // (the host copy of the rows' distinct-value counts belongs to the rows prepared before)
    return TW_OK;
}

// The fits themselves; the tape and the slots' offsets are resident in e->fit_tape / e->fit_tape_off.
int fit_run(tw_engine* e) {
    static const std::vector<double> tape100 = [] {   // GaussianMixture(n, random_state=100): a fresh MT19937(100) per fit
        std::vector<double> t((size_t)kMaxComp * 13, 0.0);
        for (int k = 1; k <= kMaxComp; k++) {
            Mt19937 r(100u);
            for (int j = 0; j < fit_draws(k); j++) t[(size_t)(k - 1) * 13 + j] = r.random_sample();
        }
        return t;
    }();
    HIPCHK(hipMemcpyAsync(e->fit_tape100, tape100.data(), sizeof(double) * tape100.size(), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemsetAsync(e->P.err, 0, sizeof(int32_t), e->stream));
    FitDev F = fit_dev(e);
    const int threads = e->coop >= 64 ? kFitThreads : e->coop;
    // Model selection: the fits of one component count depend on nothing but their own k-means start -- a stream per count (the
    // class streams, idle between the passes), seed then EM, the largest count (the longest fits) first: the EM sweeps of one count
    // (f64 arithmetic) run beside the k-means++ walks of another (waiting on loads).  TW_FIT_STREAMS=0: two launches over all counts.
    static const int fit_streams = env_int("TW_FIT_STREAMS", 0);
    if (fit_streams != 0 && kMaxComp <= kMaxEp) {
        (void)hipEventRecord(e->cls_ev[0], e->stream);
        for (int k = kMaxComp; k >= 1; k--) {
            { int rs = ensure_class_stream(e, k); if (rs != TW_OK) return rs; }
            FitDev Fk = F;
            Fk.k_only = k;
            hipStream_t st = e->cls_stream[k];
            (void)hipStreamWaitEvent(st, e->cls_ev[0], 0);
            if (k >= 2) hipLaunchKernelGGL(k_fit_seed<false>, dim3((unsigned)e->n_slots), dim3(threads), 0, st, Fk);
            hipLaunchKernelGGL(k_fit_em<false>, dim3((unsigned)e->n_slots), dim3(threads), 0, st, Fk);
            (void)hipEventRecord(e->cls_ev[k], st);
            (void)hipStreamWaitEvent(e->stream, e->cls_ev[k], 0);
        }
    } else {
        hipLaunchKernelGGL(k_fit_seed<false>, dim3((unsigned)(e->n_slots * (kMaxComp - 1))), dim3(threads), 0, e->stream, F);
        hipLaunchKernelGGL(k_fit_em<false>, dim3((unsigned)(e->n_slots * kMaxComp)), dim3(threads), 0, e->stream, F);
    }
    hipLaunchKernelGGL(k_fit_select, dim3((unsigned)((e->n_slots + 63) / 64)), dim3(64), 0, e->stream, F);
    hipLaunchKernelGGL(k_fit_seed<true>, dim3((unsigned)e->n_slots), dim3(threads), 0, e->stream, F);
    hipLaunchKernelGGL(k_fit_em<true>, dim3((unsigned)e->n_slots), dim3(threads), 0, e->stream, F);
    const int64_t total = e->n_slots * kMaxComp;
    hipLaunchKernelGGL(k_mix_consts, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, e->stream, (const double*)e->mix_p_dev, (const int32_t*)e->mix_n_dev, (const int32_t*)e->slot_unit, e->P.units, e->mix_c_dev, total);
    HIPCHK(hipEventRecord(e->ev[EV_END], e->stream));
    HIPCHK(hipGetLastError());
    int32_t kerr = 0;
    HIPCHK(hipMemcpyAsync(&kerr, e->P.err, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    float f = 0.f;
    HIPCHK(hipEventElapsedTime(&f, e->ev[EV_BEGIN], e->ev[EV_END]));
    e->fit_ms = f;
    if (kerr != 0)
        return fail(e, kerr, "refit: every model-selection fit of an edge ended in a covariance <= 0 (scikit-learn's ValueError); the reference "
                             "raises there as well (np.argmin of an empty list, traceweaver_v3.py:777-780)");
    e->state = ST_MIX;
    return TW_OK;
}

int fit_check_state(tw_engine* e, const char* who) {
    if (e->state != ST_PASS1) return fail(e, TW_ERR_STATE, std::string(who) + " is valid right after tw_run_pass1");
    if (e->skip_mode) return fail(e, TW_ERR_STATE, "a skip-mode batch runs one pass (traceweaver_v3.py:1155-1156): there is no refit");
    return TW_OK;
}

}  // namespace

int tw_set_fit_seed(tw_engine* e, uint32_t seed) {
    if (e == nullptr) return TW_ERR_ARG;
    e->fit_seed = seed;
    e->fit_rng = Mt19937(seed);
    return TW_OK;
}

int tw_fit_rows(tw_engine* e, int32_t* max_n) {
    if (e == nullptr || max_n == nullptr) return TW_ERR_ARG;
    int rc = fit_check_state(e, "tw_fit_rows");
    if (rc != TW_OK) return rc;
    HIPCHK(hipSetDevice(e->device));
    rc = fit_prepare(e);
    if (rc != TW_OK) return rc;
    std::vector<int32_t> uniq((size_t)e->n_slots);
    HIPCHK(hipMemcpyAsync(uniq.data(), e->fit_row_uniq, sizeof(int32_t) * e->n_slots, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->fit_max_n.assign((size_t)e->n_slots, 0);
    for (int64_t q = 0; q < e->n_slots; q++) max_n[q] = e->fit_max_n[(size_t)q] = std::min<int32_t>(uniq[(size_t)q], kMaxComp);
    e->fit_max_n_valid = true;
    return TW_OK;
}

int tw_fit_mixtures_tape(tw_engine* e, const double* tape, int64_t tape_len, const int64_t* slot_off) {
    if (e == nullptr || tape == nullptr || slot_off == nullptr || tape_len < 0) return TW_ERR_ARG;
    int rc = fit_check_state(e, "tw_fit_mixtures_tape");
    if (rc != TW_OK) return rc;
    HIPCHK(hipSetDevice(e->device));
    if (!e->fit_prepared || !e->fit_max_n_valid || e->fit_max_n.size() != (size_t)e->n_slots) {
        std::vector<int32_t> tmp((size_t)e->n_slots);
        rc = tw_fit_rows(e, tmp.data());
        if (rc != TW_OK) return rc;
    }
    static const int need[kMaxComp + 1] = {0, 1, 4, 11, 21, 34};
    for (int64_t q = 0; q < e->n_slots; q++) {
        const int m = e->fit_max_n[(size_t)q];
        if (m > 0 && (slot_off[q] < 0 || slot_off[q] + need[m] > tape_len))
            return fail(e, TW_ERR_ARG, "tw_fit_mixtures_tape: the tape does not hold the draws of slot " + std::to_string(q));
    }
    HIPCHK(hipEventRecord(e->ev[EV_BEGIN], e->stream));
    if (tape_len > e->fit_tape_cap) {
        double* p = nullptr;
        rc = dev_alloc(e, &p, tape_len);
        if (rc != TW_OK) return rc;
        e->fit_tape = p;
        e->fit_tape_cap = tape_len;
    }
    HIPCHK(hipMemcpyAsync(e->fit_tape, tape, sizeof(double) * (size_t)std::max<int64_t>(tape_len, 0), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->fit_tape_off, slot_off, sizeof(int64_t) * e->n_slots, hipMemcpyHostToDevice, e->stream));
    return fit_run(e);
}

int tw_fit_mixtures_seeded(tw_engine* e, const uint32_t* unit_seed) {
    if (e == nullptr) return TW_ERR_ARG;
    int rc = fit_check_state(e, "tw_fit_mixtures");
    if (rc != TW_OK) return rc;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipEventRecord(e->ev[EV_BEGIN], e->stream));
    rc = fit_prepare_launch(e);   // (the rows become runs while the host draws the uniforms; fit_prepare below waits for both)
    if (rc != TW_OK) return rc;
    // one block of 34 uniforms per slot: no look at the rows needed, no host round trip.  With unit seeds every unit draws from
    // a stream of its own, so its fit does not depend on which other units share the batch (sharded runs);
    // otherwise all slots draw from the engine's stream in slot order.
    static thread_local std::vector<double> tape;
    static thread_local std::vector<int64_t> off;
    tape.resize((size_t)e->n_slots * kFitRowTape);
    off.resize((size_t)e->n_slots);
    for (int64_t q = 0; q < e->n_slots; q++) off[(size_t)q] = q * kFitRowTape;
    if (unit_seed == nullptr) {
        for (size_t i = 0; i < tape.size(); i++) tape[i] = e->fit_rng.random_sample();
    } else {
        for (size_t u = 0; u < e->units.size(); u++) {
            Mt19937 r(unit_seed[u]);
            const size_t lo = (size_t)e->units[u].slot_off * kFitRowTape, hi = lo + (size_t)e->units[u].nslot * kFitRowTape;
            for (size_t i = lo; i < hi; i++) tape[i] = r.random_sample();
        }
    }
    if ((int64_t)tape.size() > e->fit_tape_cap) return fail(e, TW_ERR_STATE, "tw_fit_mixtures: tape buffer smaller than the batch");
    HIPCHK(hipMemcpyAsync(e->fit_tape, tape.data(), sizeof(double) * tape.size(), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->fit_tape_off, off.data(), sizeof(int64_t) * off.size(), hipMemcpyHostToDevice, e->stream));
    if (e->fit_prepared) HIPCHK(hipStreamSynchronize(e->stream));   // (the host vectors are reused by the next call; else fit_prepare's wait covers it)
    rc = fit_prepare(e);
    if (rc != TW_OK) return rc;
    return fit_run(e);
}

int tw_fit_mixtures(tw_engine* e) { return tw_fit_mixtures_seeded(e, nullptr); }

int tw_get_mixtures(tw_engine* e, int32_t* mix_n, double* mix_p) {
    if (e == nullptr || mix_n == nullptr || mix_p == nullptr) return TW_ERR_ARG;
    if (e->state < ST_MIX) return fail(e, TW_ERR_STATE, "no mixtures resident (tw_fit_mixtures / tw_set_mixtures first)");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(mix_n, e->mix_n_dev, sizeof(int32_t) * e->n_slots, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(mix_p, e->mix_p_dev, sizeof(double) * e->n_slots * kMaxComp * 3, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return TW_OK;
}

int tw_run_pass2(tw_engine* e) {
    if (e == nullptr) return TW_ERR_ARG;
    if (e->state != ST_MIX && e->state != ST_PASS2) return fail(e, TW_ERR_STATE, "tw_run_pass2 needs tw_run_pass1 and tw_set_mixtures first");
    // (tw_set_gaps* + tw_set_mixtures reach ST_MIX on a batch that never ran pass 1 -- an engine that only serves the refit; the second
    // pass reads what the first left behind: cut-offs, window flags, tuple counts)
    if (!e->pass1_done) return fail(e, TW_ERR_STATE, "tw_run_pass2 on a batch whose first pass has not run (tw_run_pass1 first)");
    HIPCHK(hipSetDevice(e->device));
    const int rc = run_pass(e, 2);
    if (rc == TW_OK) e->state = ST_PASS2;
    return rc;
}

int tw_get_results(tw_engine* e, int pass, const tw_results* r) {
    if (e == nullptr || r == nullptr) return TW_ERR_ARG;
    const bool ok = (pass == 1 && (e->state == ST_PASS1 || e->state == ST_MIX)) || (pass == 2 && e->state == ST_PASS2);
    if (!ok) return fail(e, TW_ERR_STATE, "results of that pass are not resident (fetch pass-1 results before running pass 2)");
    HIPCHK(hipSetDevice(e->device));
    const Dev& P = e->P;
    const int64_t n = P.n_in_total;
#define D2H(dst, src, bytes) if ((dst) != nullptr) HIPCHK(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, e->stream))
    D2H(r->parent, P.parent, sizeof(int32_t) * e->n_ie);
    D2H(r->topk_idx, P.tk_idx, sizeof(int32_t) * e->n_ie * kTopK);
    D2H(r->topk_score, P.tk_score, sizeof(double) * n * kTopK);
    D2H(r->topk_n, P.tk_n, sizeof(int32_t) * n);
    D2H(r->chosen, P.chosen, sizeof(int32_t) * n);
    D2H(r->leaves, P.leaves, sizeof(int64_t) * n);
    D2H(r->window_end, P.win_end, sizeof(uint8_t) * n);
    D2H(r->unit_stats, P.unit_stats, sizeof(int64_t) * 8 * P.n_units);
#undef D2H
    HIPCHK(hipStreamSynchronize(e->stream));
    return TW_OK;
}

int tw_get_gauss_params(tw_engine* e, double* gauss) {
    if (e == nullptr || gauss == nullptr) return TW_ERR_ARG;
    if (e->state < ST_PASS1) return fail(e, TW_ERR_STATE, "tw_get_gauss_params before tw_run_pass1");
    HIPCHK(hipSetDevice(e->device));
    std::vector<double> tmp((size_t)e->n_gp * 4);
    HIPCHK(hipMemcpyAsync(tmp.data(), e->P.gparam, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int64_t q = 0; q < e->n_gp; q++)
        for (int k = 0; k < 3; k++) gauss[q * 3 + k] = tmp[(size_t)q * 4 + k];
    for (const UnitDev& U : e->units)  // the device keeps the means in timestamp units (k_block_params)
        for (int64_t q = U.gp_off; q < U.gp_off + (int64_t)U.nblk * U.nslot; q++) gauss[q * 3] *= U.tscale;
    return TW_OK;
}

int tw_get_timing(tw_engine* e, double* ms, int32_t n) {
    if (e == nullptr || ms == nullptr) return TW_ERR_ARG;
    for (int i = 0; i < n && i < 6; i++) ms[i] = e->ms[i];
    if (n > 6) ms[6] = e->fit_ms;
    if (n > 7) ms[7] = (double)e->rounds;
    if (n > 8) ms[8] = e->host_ms[0];
    if (n > 9) ms[9] = e->host_ms[1];
    return TW_OK;
}

/* Debug aid, not part of the public header: phase timers of -DTW_PROFILE builds (zeros otherwise). */
int tw_debug_profile_hist(tw_engine* e, unsigned long long* out16) {   // item durations of the profiled kernel, by powers of two (TW_ITEM_END)
    if (e == nullptr || out16 == nullptr || e->state < ST_LOADED) return TW_ERR_ARG;
    HIPCHK(hipMemcpyAsync(out16, e->P.prof + 16, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return TW_OK;
}

int tw_debug_profile(tw_engine* e, unsigned long long* out16) {
    if (e == nullptr || out16 == nullptr || e->state < ST_LOADED) return TW_ERR_ARG;
    HIPCHK(hipMemcpyAsync(out16, e->P.prof, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return TW_OK;
}

int tw_find_order(tw_engine* e, int32_t n_units, const int64_t* unit_in_off, const int32_t* unit_E, const int64_t* ep_off,
                  const int64_t* out_start, const int64_t* out_end, const int32_t* true_child, uint8_t* dag_out) {
    if (e == nullptr || n_units <= 0 || !unit_in_off || !unit_E || !ep_off || !out_start || !out_end || !true_child || !dag_out) return TW_ERR_ARG;
    HIPCHK(hipSetDevice(e->device));
    std::vector<int64_t> ep_base((size_t)n_units), ie_off((size_t)n_units);
    int64_t epi = 0, ie = 0, max_n = 0;
    for (int u = 0; u < n_units; u++) {
        if (unit_E[u] < 1 || unit_E[u] > TW_MAX_EP) return fail(e, TW_ERR_UNSUPPORTED, "unit has E outside [1, TW_MAX_EP]");
        const int64_t n = unit_in_off[u + 1] - unit_in_off[u];
        ep_base[(size_t)u] = epi; ie_off[(size_t)u] = ie;
        epi += unit_E[u]; ie += n * unit_E[u];
        max_n = std::max(max_n, n);
    }
    const int64_t n_out = ep_off[epi];
    std::vector<void*> tmp;
    auto up = [&](const void* src, size_t bytes, void** dst) -> hipError_t {
        hipError_t s = hipMalloc(dst, std::max<size_t>(bytes, 8));
        if (s != hipSuccess) return s;
        tmp.push_back(*dst);
        return hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, e->stream);
    };
    OrderDev O{};
    O.n_units = n_units;
    void *d_io, *d_E, *d_eb, *d_eo, *d_ie, *d_os, *d_oe, *d_tr, *d_v;
    hipError_t s = up(unit_in_off, sizeof(int64_t) * (size_t)(n_units + 1), &d_io);
    if (s == hipSuccess) s = up(unit_E, sizeof(int32_t) * (size_t)n_units, &d_E);
    if (s == hipSuccess) s = up(ep_base.data(), sizeof(int64_t) * (size_t)n_units, &d_eb);
    if (s == hipSuccess) s = up(ep_off, sizeof(int64_t) * (size_t)(epi + 1), &d_eo);
    if (s == hipSuccess) s = up(ie_off.data(), sizeof(int64_t) * (size_t)n_units, &d_ie);
    if (s == hipSuccess) s = up(out_start, sizeof(int64_t) * (size_t)n_out, &d_os);
    if (s == hipSuccess) s = up(out_end, sizeof(int64_t) * (size_t)n_out, &d_oe);
    if (s == hipSuccess) s = up(true_child, sizeof(int32_t) * (size_t)ie, &d_tr);
    if (s == hipSuccess) { s = hipMalloc(&d_v, sizeof(unsigned long long) * (size_t)n_units); if (s == hipSuccess) tmp.push_back(d_v); }
    if (s == hipSuccess) s = hipMemsetAsync(d_v, 0, sizeof(unsigned long long) * (size_t)n_units, e->stream);
    std::vector<unsigned long long> viol((size_t)n_units, 0ull);
    if (s == hipSuccess) {
        O.unit_in_off = (const int64_t*)d_io; O.unit_E = (const int32_t*)d_E; O.ep_base = (const int64_t*)d_eb; O.ep_off = (const int64_t*)d_eo;
        O.ie_off = (const int64_t*)d_ie; O.out_start = (const int64_t*)d_os; O.out_end = (const int64_t*)d_oe; O.truth = (const int32_t*)d_tr;
        O.viol = (unsigned long long*)d_v;
        const int threads = e->coop >= 64 ? 256 : e->coop;
        const unsigned bx = (unsigned)std::min<int64_t>((max_n + threads - 1) / threads + 1, 1024);
        for (int u0 = 0; u0 < n_units && s == hipSuccess; u0 += 32768) {  // grid.y is limited to 65535
            OrderDev Q = O;
            Q.unit_in_off += u0; Q.unit_E += u0; Q.ep_base += u0; Q.ie_off += u0; Q.viol += u0;
            hipLaunchKernelGGL(k_find_order, dim3(bx, (unsigned)std::min(n_units - u0, 32768)), dim3((unsigned)threads), 0, e->stream, Q);
            s = hipGetLastError();
        }
        if (s == hipSuccess) s = hipMemcpyAsync(viol.data(), d_v, sizeof(unsigned long long) * (size_t)n_units, hipMemcpyDeviceToHost, e->stream);
        if (s == hipSuccess) s = hipStreamSynchronize(e->stream);
    }
    for (void* q : tmp) (void)hipFree(q);
    if (s != hipSuccess) return fail(e, TW_ERR_DEVICE, std::string("tw_find_order: ") + hipGetErrorString(s));
    int64_t di = 0;
    for (int u = 0; u < n_units; u++) {
        const int E = unit_E[u];
        for (int a = 0; a < E; a++)
            for (int b = 0; b < E; b++) dag_out[di + a * E + b] = (a != b && !((viol[(size_t)u] >> (a * TW_MAX_EP + b)) & 1ull)) ? 1 : 0;
        di += E * E;
    }
    return TW_OK;
}

int tw_set_truth(tw_engine* e, const int32_t* true_child, const int32_t* in_trace, int64_t n_traces) {
    if (e == nullptr || true_child == nullptr) return TW_ERR_ARG;
    if (e->state < ST_LOADED) return fail(e, TW_ERR_STATE, "tw_set_truth before tw_load_batch");
    if (in_trace != nullptr && n_traces <= 0) return fail(e, TW_ERR_ARG, "in_trace given without n_traces");
    HIPCHK(hipSetDevice(e->device));
    // k_evaluate indexes trace_bad[] with in_trace and compares parents with true_child: both are checked here, once,
    // on the host (an out-of-range trace number would be an out-of-bounds store on the device)
    for (const UnitDev& U : e->units) {
        const int32_t* t = true_child + U.ie_off;
        for (int64_t k = 0; k < (int64_t)U.E * U.n_in; k++)
            if (t[k] < (e->skip_mode ? -2 : -1) || t[k] >= U.n_in) return fail(e, TW_ERR_ARG, "tw_set_truth: true_child outside [-1, n_in) (-2 = skipped, skip-mode batches only)");
    }
    if (in_trace != nullptr)
        for (int64_t k = 0; k < e->P.n_in_total; k++)
            if (in_trace[k] < -1 || in_trace[k] >= n_traces) return fail(e, TW_ERR_ARG, "tw_set_truth: in_trace outside [-1, n_traces)");
    int rc;
    if (e->truth == nullptr) {
        rc = dev_alloc(e, &e->truth, e->n_ie); if (rc != TW_OK) return rc;
        rc = dev_alloc(e, &e->eval_counts, (int64_t)e->P.n_units * 4 + 2); if (rc != TW_OK) return rc;
    }
    HIPCHK(hipMemcpyAsync(e->truth, true_child, sizeof(int32_t) * e->n_ie, hipMemcpyHostToDevice, e->stream));
    e->n_traces = 0;
    if (in_trace != nullptr) {  // (buffers are freed with the batch)
        if (e->in_trace == nullptr) { rc = dev_alloc(e, &e->in_trace, e->P.n_in_total); if (rc != TW_OK) return rc; }
        if (e->trace_bad == nullptr || n_traces > e->trace_cap) {
            rc = dev_alloc(e, &e->trace_bad, 2 * n_traces); if (rc != TW_OK) return rc;
            e->trace_cap = n_traces;
        }
        HIPCHK(hipMemcpyAsync(e->in_trace, in_trace, sizeof(int32_t) * e->P.n_in_total, hipMemcpyHostToDevice, e->stream));
        e->n_traces = n_traces;
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    return TW_OK;
}

int tw_evaluate(tw_engine* e, int64_t* per_unit, uint8_t* trace_flags, int64_t* e2e) {
    if (e == nullptr || per_unit == nullptr) return TW_ERR_ARG;
    if (e->state != ST_PASS1 && e->state != ST_MIX && e->state != ST_PASS2) return fail(e, TW_ERR_STATE, "tw_evaluate needs the results of a pass");
    if (e->truth == nullptr) return fail(e, TW_ERR_STATE, "tw_evaluate before tw_set_truth");
    if ((trace_flags != nullptr || e2e != nullptr) && e->n_traces == 0) return fail(e, TW_ERR_STATE, "per-trace accuracy needs in_trace (tw_set_truth)");
    HIPCHK(hipSetDevice(e->device));
    const Dev& P = e->P;
    const int64_t nc = (int64_t)P.n_units * 4 + 2;
    HIPCHK(hipMemsetAsync(e->eval_counts, 0, sizeof(unsigned long long) * nc, e->stream));
    if (e->n_traces > 0) HIPCHK(hipMemsetAsync(e->trace_bad, 0, (size_t)(2 * e->n_traces), e->stream));
    hipLaunchKernelGGL(k_evaluate, dim3(P.n_tiles), dim3(e->tile), 0, e->stream, P, (const int32_t*)e->truth,
                       (const int32_t*)(e->n_traces > 0 ? e->in_trace : nullptr), e->eval_counts, e->trace_bad,
                       e->n_traces > 0 ? e->trace_bad + e->n_traces : nullptr);
    if (e->n_traces > 0 && e2e != nullptr)
        hipLaunchKernelGGL(k_count_flags, dim3((unsigned)std::min<int64_t>(e->n_traces / 1024 + 1, 1024)), dim3(e->coop >= 64 ? 256 : e->coop), 0, e->stream,
                           (const uint8_t*)e->trace_bad, (const uint8_t*)(e->trace_bad + e->n_traces), e->n_traces, e->eval_counts + (nc - 2));
    HIPCHK(hipGetLastError());
    std::vector<unsigned long long> h((size_t)nc);
    HIPCHK(hipMemcpyAsync(h.data(), e->eval_counts, sizeof(unsigned long long) * nc, hipMemcpyDeviceToHost, e->stream));
    if (trace_flags != nullptr) HIPCHK(hipMemcpyAsync(trace_flags, e->trace_bad, (size_t)(2 * e->n_traces), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int u = 0; u < P.n_units; u++) {
        const int64_t n = e->units[(size_t)u].n_in;
        per_unit[u * 4 + 0] = n;
        per_unit[u * 4 + 1] = n - (int64_t)h[(size_t)u * 4 + 1];
        per_unit[u * 4 + 2] = n - (int64_t)h[(size_t)u * 4 + 2];
        per_unit[u * 4 + 3] = (int64_t)h[(size_t)u * 4 + 3];
    }
    if (e2e != nullptr) { e2e[0] = (int64_t)h[(size_t)nc - 2]; e2e[1] = (int64_t)h[(size_t)nc - 1]; }
    return TW_OK;
}

int tw_build_distributions(tw_engine* e, int64_t n, const int64_t* start, const int64_t* dur, const uint8_t* ep, int64_t large_delay,
                           int32_t E, int32_t* key_out, int64_t* sample_out) {
    if (e == nullptr || n <= 0 || !start || !dur || !ep || !key_out || !sample_out || E < 1 || E > TW_MAX_EP) return TW_ERR_ARG;
    HIPCHK(hipSetDevice(e->device));
    int64_t *d_s = nullptr, *d_d = nullptr, *d_v = nullptr;
    uint8_t* d_e = nullptr;
    int32_t* d_k = nullptr;
    hipError_t s = hipMalloc((void**)&d_s, sizeof(int64_t) * (size_t)n);
    if (s == hipSuccess) s = hipMalloc((void**)&d_d, sizeof(int64_t) * (size_t)n);
    if (s == hipSuccess) s = hipMalloc((void**)&d_v, sizeof(int64_t) * (size_t)n);
    if (s == hipSuccess) s = hipMalloc((void**)&d_e, (size_t)n);
    if (s == hipSuccess) s = hipMalloc((void**)&d_k, sizeof(int32_t) * (size_t)n);
    if (s == hipSuccess) s = hipMemcpyAsync(d_s, start, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, e->stream);
    if (s == hipSuccess) s = hipMemcpyAsync(d_d, dur, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, e->stream);
    if (s == hipSuccess) s = hipMemcpyAsync(d_e, ep, (size_t)n, hipMemcpyHostToDevice, e->stream);
    if (s == hipSuccess) {
        const int threads = e->coop >= 64 ? 256 : e->coop;
        hipLaunchKernelGGL(k_build_distributions, dim3((unsigned)((n + threads - 1) / threads)), dim3((unsigned)threads), 0, e->stream,
                           (const int64_t*)d_s, (const int64_t*)d_d, (const uint8_t*)d_e, n, large_delay, (int)E, d_k, d_v);
        s = hipGetLastError();
    }
    if (s == hipSuccess) s = hipMemcpyAsync(key_out, d_k, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, e->stream);
    if (s == hipSuccess) s = hipMemcpyAsync(sample_out, d_v, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, e->stream);
    if (s == hipSuccess) s = hipStreamSynchronize(e->stream);
    (void)hipFree(d_s); (void)hipFree(d_d); (void)hipFree(d_v); (void)hipFree(d_e); (void)hipFree(d_k);
    if (s != hipSuccess) return fail(e, TW_ERR_DEVICE, std::string("tw_build_distributions: ") + hipGetErrorString(s));
    return TW_OK;
}

int tw_measure_hbm_copy(tw_engine* e, int64_t bytes, int32_t iters, double* gbps) {
    if (e == nullptr || gbps == nullptr || bytes < (1 << 20) || iters < 1) return TW_ERR_ARG;
    HIPCHK(hipSetDevice(e->device));
    void *a = nullptr, *b = nullptr;
    const size_t n16 = (size_t)bytes / 16;
    hipError_t s = hipMalloc(&a, n16 * 16);
    if (s == hipSuccess) s = hipMalloc(&b, n16 * 16);
    if (s == hipSuccess) s = hipMemsetAsync(a, 1, n16 * 16, e->stream);
    *gbps = 0.0;
    // One 16-byte load and store per thread, one workgroup per 4 KB (consecutive workgroups walk consecutive DRAM pages): 6.2 TB/s
    // on MI355X, the guide's float4-copy figure.  The grid-stride form of the same copy (2048 persistent workgroups, four loads in
    // flight) reaches 4.4-4.7 TB/s -- what round 3 quoted as the ceiling; it is timed too, the better one is reported.
    for (int variant = 0; variant < 2 && s == hipSuccess; variant++) {
        const unsigned grid = variant == 0 ? (unsigned)((n16 + 255) / 256) : 2048u;
        auto launch = [&]() {
            if (variant == 0) hipLaunchKernelGGL((k_copy16<1, false>), dim3(grid), dim3(256), 0, e->stream, (const uint4*)a, (uint4*)b, (int64_t)n16);
            else hipLaunchKernelGGL((k_copy16<4, false>), dim3(grid), dim3(256), 0, e->stream, (const uint4*)a, (uint4*)b, (int64_t)n16);
        };
        launch();  // warm-up
        s = hipEventRecord(e->ev[EV_BEGIN], e->stream);
        for (int k = 0; k < iters; k++) launch();
        if (s == hipSuccess) s = hipEventRecord(e->ev[EV_END], e->stream);
        if (s == hipSuccess) s = hipStreamSynchronize(e->stream);
        float ms = 0.f;
        if (s == hipSuccess) s = hipEventElapsedTime(&ms, e->ev[EV_BEGIN], e->ev[EV_END]);
        if (s == hipSuccess) {
            const double r = 2.0 * (double)(n16 * 16) * iters / ((double)ms * 1e-3) / 1e9;
            if (env_int("TW_COPY_TRACE", 0)) fprintf(stderr, "tw_measure_hbm_copy: variant %d, %u workgroups: %.0f GB/s\n", variant, grid, r);
            if (r > *gbps) *gbps = r;
        }
    }
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (s != hipSuccess) return fail(e, TW_ERR_DEVICE, std::string("tw_measure_hbm_copy: ") + hipGetErrorString(s));
    return TW_OK;
}

int tw_host_alloc(int64_t bytes, void** out) {
    if (out == nullptr || bytes < 0) return TW_ERR_ARG;
    *out = nullptr;
    return hipHostMalloc(out, (size_t)std::max<int64_t>(bytes, 8), hipHostMallocDefault) == hipSuccess ? TW_OK : TW_ERR_DEVICE;
}

void tw_host_free(void* p) {
    if (p != nullptr) (void)hipHostFree(p);
}

/* Debug aid, not part of the public header: sizes of the work lists of the last pass --
 * out[0] = windows listed for k_select_heavy, out[1 + E] = spans listed for k_enumerate_heavy<E> (E <= kMaxEp),
 * out[2 + kMaxEp] = spans enumerated in parts, out[3 + kMaxEp] = of those, enumerated once more as a whole (out: 4 + kMaxEp ints). */
namespace {

struct BaseScratch {   // device scratch of one baseline call
    std::vector<void*> ptrs;
    ~BaseScratch() { for (void* p : ptrs) (void)hipFree(p); }
    void* raw(int64_t count, size_t elem, int fill) {
        void* p = nullptr;
        const size_t bytes = elem * (size_t)std::max<int64_t>(count, 1);
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        (void)hipMemset(p, fill, bytes);
        return p;
    }
};

int base_check(tw_engine* e, const char* who) {
    if (e->state < ST_LOADED) return fail(e, TW_ERR_STATE, std::string(who) + " before tw_load_batch");
    if (e->skip_mode) return fail(e, TW_ERR_UNSUPPORTED, std::string(who) + ": the lists of a skip-mode batch are not sorted; the host baselines serve it (traceweaver_amd/baselines.py)");
    return TW_OK;
}

}  // namespace

int tw_run_baseline(tw_engine* e, int kind, int32_t* parent_out) {
    if (e == nullptr || parent_out == nullptr) return TW_ERR_ARG;
    if (kind != TW_BASELINE_FCFS && kind != TW_BASELINE_VPATH) return fail(e, TW_ERR_ARG, "tw_run_baseline: kind is TW_BASELINE_FCFS or TW_BASELINE_VPATH (WAP5: tw_wap5_delays / tw_wap5_parents)");
    int rc = base_check(e, "tw_run_baseline");
    if (rc != TW_OK) return rc;
    if (kind == TW_BASELINE_VPATH && e->truth == nullptr) return fail(e, TW_ERR_STATE, "vPath follows a returning call to its request (vpath.py:42-46,81-84): tw_set_truth first");
    HIPCHK(hipSetDevice(e->device));
    const Dev& P = e->P;
    const dim3 tiles(P.n_tiles), tb(e->tile);
    BaseScratch S;
    BaseDev B{};
    B.truth = e->truth;
    B.parent = (int32_t*)S.raw(e->n_ie, sizeof(int32_t), 0);
    if (B.parent == nullptr) return fail(e, TW_ERR_DEVICE, "tw_run_baseline: out of device memory");
    if (kind == TW_BASELINE_FCFS) {
        hipLaunchKernelGGL(k_base_fcfs, tiles, tb, 0, e->stream, P, B);
    } else {
        B.count = (int32_t*)S.raw(e->n_ie, sizeof(int32_t), 0);
        B.unit_maxdur = (int64_t*)S.raw(P.n_units, sizeof(int64_t), 0);
        B.owner = (int32_t*)S.raw(P.n_out_total, sizeof(int32_t), 0x7f);
        int32_t *d_in = (int32_t*)S.raw(P.n_in_total, sizeof(int32_t), 0), *d_out = (int32_t*)S.raw(P.n_out_total, sizeof(int32_t), 0);
        int32_t *inv_in = (int32_t*)S.raw(P.n_in_total, sizeof(int32_t), 0), *inv_out = (int32_t*)S.raw(P.n_out_total, sizeof(int32_t), 0);
        if (B.count == nullptr || B.unit_maxdur == nullptr || B.owner == nullptr || d_in == nullptr || d_out == nullptr || inv_in == nullptr || inv_out == nullptr)
            return fail(e, TW_ERR_DEVICE, "tw_run_baseline: out of device memory");
        B.in_rank_inv = inv_in; B.out_rank_inv = inv_out;
        HIPCHK(hipDeviceSynchronize());   // (the fills above ran on the null stream)
        hipLaunchKernelGGL(k_rank_ends, tiles, tb, 0, e->stream, P, all_tiles_host(e), d_in, d_out);
        hipLaunchKernelGGL(k_place_ends, tiles, tb, 0, e->stream, P, all_tiles_host(e), (const int32_t*)d_in, (const int32_t*)d_out);
        hipLaunchKernelGGL(k_base_rank_inverse, tiles, tb, 0, e->stream, P, (const int32_t*)d_in, (const int32_t*)d_out, inv_in, inv_out);
        hipLaunchKernelGGL(k_base_prepare, tiles, tb, 0, e->stream, P, B);
        hipLaunchKernelGGL(k_base_vpath, tiles, tb, 0, e->stream, P, B);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(parent_out, B.parent, sizeof(int32_t) * e->n_ie, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return TW_OK;
}

int tw_wap5_delays(tw_engine* e, int64_t* delay_sum, int32_t* delay_cnt, int64_t* unit_maxdur) {
    if (e == nullptr || delay_sum == nullptr || delay_cnt == nullptr) return TW_ERR_ARG;
    int rc = base_check(e, "tw_wap5_delays");
    if (rc != TW_OK) return rc;
    HIPCHK(hipSetDevice(e->device));
    const Dev& P = e->P;
    const dim3 tiles(P.n_tiles), tb(e->tile);
    BaseScratch S;
    BaseDev B{};
    B.parent = (int32_t*)S.raw(e->n_ie, sizeof(int32_t), 0); B.count = (int32_t*)S.raw(e->n_ie, sizeof(int32_t), 0);
    B.unit_maxdur = (int64_t*)S.raw(P.n_units, sizeof(int64_t), 0);
    B.delay_sum = (long long*)S.raw((int64_t)P.n_units * kMaxEp, sizeof(long long), 0); B.delay_cnt = (int32_t*)S.raw((int64_t)P.n_units * kMaxEp, sizeof(int32_t), 0);
    if (B.parent == nullptr || B.count == nullptr || B.unit_maxdur == nullptr || B.delay_sum == nullptr || B.delay_cnt == nullptr)
        return fail(e, TW_ERR_DEVICE, "tw_wap5_delays: out of device memory");
    HIPCHK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_base_prepare, tiles, tb, 0, e->stream, P, B);
    hipLaunchKernelGGL(k_wap5_delays, tiles, tb, 0, e->stream, P, B);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(delay_sum, B.delay_sum, sizeof(int64_t) * P.n_units * kMaxEp, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(delay_cnt, B.delay_cnt, sizeof(int32_t) * P.n_units * kMaxEp, hipMemcpyDeviceToHost, e->stream));
    if (unit_maxdur != nullptr) HIPCHK(hipMemcpyAsync(unit_maxdur, B.unit_maxdur, sizeof(int64_t) * P.n_units, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return TW_OK;
}

int tw_wap5_parents(tw_engine* e, const double* mean, int32_t* parent_out, int32_t* call_request) {
    if (e == nullptr || mean == nullptr || parent_out == nullptr) return TW_ERR_ARG;
    int rc = base_check(e, "tw_wap5_parents");
    if (rc != TW_OK) return rc;
    HIPCHK(hipSetDevice(e->device));
    const Dev& P = e->P;
    const dim3 tiles(P.n_tiles), tb(e->tile);
    BaseScratch S;
    BaseDev B{};
    B.parent = (int32_t*)S.raw(e->n_ie, sizeof(int32_t), 0xff); B.count = (int32_t*)S.raw(e->n_ie, sizeof(int32_t), 0);
    double* mean_d = (double*)S.raw((int64_t)P.n_units * kMaxEp, sizeof(double), 0);
    B.call_request = (int32_t*)S.raw(P.n_out_total, sizeof(int32_t), 0xff);
    if (B.parent == nullptr || B.count == nullptr || mean_d == nullptr || B.call_request == nullptr) return fail(e, TW_ERR_DEVICE, "tw_wap5_parents: out of device memory");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyAsync(mean_d, mean, sizeof(double) * P.n_units * kMaxEp, hipMemcpyHostToDevice, e->stream));
    B.mean = mean_d;
    hipLaunchKernelGGL(k_wap5_parents, tiles, tb, 0, e->stream, P, B);
    hipLaunchKernelGGL(k_wap5_finish, tiles, tb, 0, e->stream, P, B);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(parent_out, B.parent, sizeof(int32_t) * e->n_ie, hipMemcpyDeviceToHost, e->stream));
    if (call_request != nullptr) HIPCHK(hipMemcpyAsync(call_request, B.call_request, sizeof(int32_t) * (size_t)P.n_out_total, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return TW_OK;
}

/* Debug aid, not part of the public header: the model-selection fits of the last refit, [n_slots][5][16] = BIC (inf = raised),
 * weights[5], means[5], precision_cholesky[5]. */
extern "C" int tw_debug_fit_models(tw_engine* e, double* out) {
    if (e == nullptr || out == nullptr) return TW_ERR_ARG;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(out, e->fit_models, sizeof(double) * e->n_slots * kMaxComp * kModelStride, hipMemcpyDeviceToHost, e->stream));

    HIPCHK(hipStreamSynchronize(e->stream));
    return TW_OK;
}

int tw_debug_worklists(tw_engine* e, int32_t* out) {
    if (e == nullptr || out == nullptr || e->state < ST_PASS1) return TW_ERR_ARG;
    static int32_t sel_all[4 * kSelSeg * kCtrStride];
    HIPCHK(hipMemcpyAsync(sel_all, e->P.heavy_count, sizeof(sel_all), hipMemcpyDeviceToHost, e->stream));
    int32_t both[2 * (kMaxEp + 1)];
    HIPCHK(hipMemcpyAsync(both, e->P.heavy_in_count, sizeof(both), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    out[0] = 0;
    for (int k = 0; k < 4 * kSelSeg; k++) out[0] += sel_all[k * kCtrStride];
    for (int k = 0; k <= kMaxEp; k++) out[1 + k] = both[k] + both[kMaxEp + 1 + k];
    if (e->debug_lists != 0)   // (the first enumeration's lists, as its chain left them: narrow + wide + long enumerations)
        for (int k = 1; k <= kMaxEp; k++) out[1 + k] = e->first_lists[0][k] + e->first_lists[1][k] + e->first_lists[2][k];
    int32_t split[kMaxEp + 1];   // [0] spans listed again after the merge, [E >= 2] split spans of the class
    HIPCHK(hipMemcpy(split, e->P.split_count, sizeof(split), hipMemcpyDeviceToHost));
    out[2 + kMaxEp] = 0; out[3 + kMaxEp] = split[0];
    for (int k = 2; k <= kMaxEp; k++) out[2 + kMaxEp] += split[k];
    // [12] items k_enumerate_lean handed to k_enumerate_heavy, [13] spans refused list parts / parts (budget or arena), [14] list parts
    int32_t more[3][kMaxEp + 1];
    HIPCHK(hipMemcpy(more[0], e->P.fb_total, sizeof(more[0]), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(more[1], e->P.defer_refused, sizeof(more[1]), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(more[2], e->P.defer_count, sizeof(more[2]), hipMemcpyDeviceToHost));
    for (int q = 0; q < 3; q++) { out[4 + kMaxEp + q] = 0; for (int k = 0; k <= kMaxEp; k++) out[4 + kMaxEp + q] += more[q][k]; }
    return TW_OK;
}

int tw_assign_service(tw_engine* e, int32_t n_in, const int64_t* in_start, const int64_t* in_end, int32_t E,
                      const int64_t* out_off, const int64_t* out_start, const int64_t* out_end, const uint8_t* dag,
                      const int32_t* key_rank, const int32_t* mix_n, const double* mix_p, const tw_results* r) {
    if (e == nullptr) return TW_ERR_ARG;
    const int64_t in_off[2] = {0, n_in};
    tw_batch b;
    b.n_units = 1; b.unit_in_off = in_off; b.unit_E = &E; b.ep_off = out_off; b.dag = dag; b.key_rank = key_rank;
    b.in_start = in_start; b.in_end = in_end; b.out_start = out_start; b.out_end = out_end;
    b.batch_size = 100; b.batch_size_mis = 30; b.topk = TW_TOPK; b.unit_time_scale = nullptr; b.skip = nullptr; b.unit_part = nullptr;
    int rc = tw_load_batch(e, &b, 0);
    if (rc == TW_OK) rc = tw_run_pass1(e);
    if (rc == TW_OK && mix_n != nullptr) {
        rc = tw_set_mixtures(e, mix_n, mix_p);
        if (rc == TW_OK) rc = tw_run_pass2(e);
    }
    if (rc == TW_OK && r != nullptr) rc = tw_get_results(e, mix_n != nullptr ? 2 : 1, r);
    return rc;
}

}  // extern "C"
