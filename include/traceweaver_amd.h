/*
 * traceweaver_amd.h -- C-ABI of the MI355X span->parent assignment engine (libtwgpu.so).
 *
 * The reference (Sachin-A/TraceWeaver) has no FFI; its drop-in boundary is the Python predictor
 * protocol (algorithms/README.md:12-73) as used by the executor:
 *     TraceWeaverV3(all_spans, all_processes).FindAssignments(method, process, in_span_partitions,
 *         out_span_partitions, parallel, instrumented_hops, true_assignments, invocation_graph)
 *     -> (all_assignments, all_topk_assignments, not_best_count, n_in, per_span_candidates,
 *         cnt_unassigned)                       traceweaver_v3.py:1087-1229, executor.py:1172-1175
 * traceweaver_amd/predictor.py mirrors that signature and drives the functions below through
 * ctypes.  Strings and (trace_id, span_id) keys never cross this ABI: the shim maps ids <-> indices.
 *
 * All pointers are plain host pointers unless a function says otherwise; the caller owns every
 * host buffer, the engine owns every device buffer.  Every function returns TW_OK (0) or a negative
 * tw_status; tw_last_error() gives the message.  One engine may be used from one thread at a time.
 *
 * Data model ("batch" = independent service units solved together; "unit" = one call of the
 * reference's FindAssignments):
 *   - incoming spans of a unit sorted by (start, end)                      executor.py:1112
 *   - outgoing endpoints in the topological order of the call-order DAG    traceweaver_v1.py:37-39
 *   - every endpoint's spans sorted by (start, end)
 *   - timestamps are int64 microseconds (Jaeger JSON startTime / startTime+duration)
 * A batch is either a no-skip batch (every endpoint of every unit holds exactly as many outgoing spans as the unit
 * has incoming spans, traceweaver_v3.py:972,1141-1158 "equal_eps": two passes with the mixture refit in between) or a
 * skip-mode batch (tw_batch.skip != NULL; exps/exp2: some requests did not call an endpoint: ONE pass with skip spans,
 * traceweaver_v3.py:820-989,1155-1156).
 */
#ifndef TRACEWEAVER_AMD_H
#define TRACEWEAVER_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TW_MAX_EP 8      /* outgoing endpoints per unit                                     */
#define TW_TOPK 5        /* candidates kept per incoming span        traceweaver_v3.py:1109 */
#define TW_MAX_COMP 5    /* mixture components per edge              traceweaver_v3.py:768  */
#define TW_MAX_WINDOW 32 /* incoming spans per window (reference cap: 31)    */
#define TW_CAND_WORDS 2  /* 64-bit words of the per-(span, endpoint) candidate bitmap       */
#define TW_SKIP_BASE 1024
#define TW_SKIP_STRIDE 128

typedef enum {
    TW_OK = 0,
    TW_ERR_ARG = -1,          /* bad argument / inconsistent sizes                                   */
    TW_ERR_UNSUPPORTED = -2,  /* skip mode (n_out != n_in), E > TW_MAX_EP, topk != TW_TOPK           */
    TW_ERR_DEVICE = -3,       /* HIP runtime error                                                   */
    TW_ERR_STATE = -4,        /* call order violated (e.g. pass 2 before pass 1)                     */
    TW_ERR_WINDOW_WIDTH = -5, /* an incoming span has > 64*TW_CAND_WORDS candidate spans at one      */
                              /* endpoint (the reference's enumeration is infeasible there too)      */
    TW_ERR_WINDOW_SIZE = -6,  /* a window holds more than TW_MAX_WINDOW incoming spans                */
    TW_ERR_NAN_PARAMS = -7,   /* a 100-span block has a single sample: std = NaN, the reference      */
                              /* aborts in the solver (SURVEY.md hazard H3)                          */
    TW_ERR_SKIP_PARAMS = -8,  /* skip mode: the scorer needs a (mean, std) pair BuildDistributions   */
                              /* did not produce (KeyError in the reference, traceweaver_v1.py:118)  */
    TW_ERR_SKIP_REFERENCE_RAISES = -9, /* skip mode: the reference raises on this input (a tuple that */
                              /* skips every endpoint, hazard H7; a score tie whose comparison       */
                              /* reaches a skip span; a skipped predecessor without a called one)    */
    TW_ERR_FIT = -10          /* refit: every model-selection fit of an edge raised in scikit-learn  */
                              /* (covariance <= 0), so the reference's np.argmin([]) raises too      */
                              /* (traceweaver_v3.py:777-780)                                         */
} tw_status;

typedef struct tw_engine tw_engine;

/* Replaces: TraceWeaverV3.__init__ (traceweaver_v3.py:30-54).  device_id = HIP device ordinal. */
int tw_create(int device_id, tw_engine **out);
void tw_destroy(tw_engine *e);
const char *tw_last_error(const tw_engine *e);

/* A batch of independent units in structure-of-arrays form.
 * Replaces: the in_span_partitions / out_span_partitions / invocation_graph arguments of
 * FindAssignments (traceweaver_v3.py:1087) for each unit.
 * Initialise the whole struct to zero (`tw_batch b = {0};` / memset) before filling it in: optional members are NULL = absent,
 * and members added by later versions of this header are appended at the end with zero meaning "as before" -- a caller that
 * leaves them uninitialised hands tw_load_batch a garbage pointer. */
typedef struct {
    int32_t n_units;
    const int64_t *unit_in_off; /* [n_units+1] offsets into in_start/in_end                        */
    const int32_t *unit_E;      /* [n_units]   outgoing endpoints per unit (1..TW_MAX_EP)           */
    const int64_t *ep_off;      /* [sum(E)+1]  offsets into out_start/out_end of every (unit,ep)    */
                                /*             segment, units in order, endpoints in topo order     */
    const uint8_t *dag;         /* concatenated E*E matrices, dag[p*E+e]=1 <=> edge p->e            */
                                /*             (executor.py:214-285 FindOrder, transitively closed) */
    const int32_t *key_rank;    /* [sum(E)]    rank of the endpoint in the partition-key order:     */
                                /*             networkx in_edges() iteration order of predecessors  */
    const int64_t *in_start, *in_end;   /* [unit_in_off[n_units]]                                   */
    const int64_t *out_start, *out_end; /* [ep_off[sum(E)]]                                         */
    int32_t batch_size;     /* 100  traceweaver_v3.py:1107 */
    int32_t batch_size_mis; /* 30   traceweaver_v3.py:1108 */
    int32_t topk;           /* 5    traceweaver_v3.py:1109 */
    /* Load-scaled units (helpers/transforms.py:10-40 repeat_change_spans, called at executor.py:1146-1148 when
     * --compress_factor > 1): the reference's timestamps are Python floats there (start / load_factor).  Such a
     * unit is handed over with its timestamps as exact integers in units of unit_time_scale[u] microseconds (a
     * power of two, e.g. 2^-3): comparisons are then the reference's float comparisons, differences are
     * (double)(t2 - t1) * scale exactly, and the sums of ComputeEpPairDistParams3 (traceweaver_v3.py:593,605)
     * accumulate in binary64 left to right as Python's sum() does over floats.  NULL = every unit holds plain
     * int64 microseconds (exact integer sums, as Python does over ints). */
    const double *unit_time_scale; /* [n_units] or NULL */
    /* Skip mode (NULL = a no-skip batch): one descriptor per unit.  The unit's lists are taken as they are -- after
     * helpers/transforms.py:153-238 (create_cache_hits) they are no longer sorted, and the reference bisects and walks
     * them as handed over (traceweaver_v3.py:1115 runs before the sort of :968-971); endpoints may hold fewer spans
     * than there are incoming spans.  Timestamps must be integer microseconds (unit_time_scale == NULL). */
    const struct tw_skip_unit *skip;
    /* Parts of a service (NULL = every unit is a whole service).  A service solved in parts -- consecutive stretches of its
     * requests, cut at PerfectCuts (traceweaver_v3.py:1024-1039), one unit each, e.g. on several GPUs -- must get the windows
     * of the unsplit run.  CreateWindows2 (:1056-1076) treats the service's first and last request specially: the first one
     * is counted twice towards the size cap (current_count starts at 1, a request after a cut starts at 0), the last one is
     * never tested for a PerfectCut.  unit_part[u] bit 0 (TW_PART_AFTER_CUT): the unit's first request follows a cut -- its
     * first window is counted like a window after a PerfectCut; bit 1 (TW_PART_BEFORE_CUT): a cut follows the unit's last
     * request -- that request is tested for a PerfectCut like any inner one. */
    const uint8_t *unit_part;   /* [n_units] or NULL */
} tw_batch;
#define TW_PART_AFTER_CUT 1
#define TW_PART_BEFORE_CUT 2

/* What TallySkipSpans (traceweaver_v3.py:853-989) and BuildDistributions (:108-172) hand the main loop. */
typedef struct tw_skip_unit {
    int32_t n_tw;            /* time windows of 30 requests (traceweaver_v3.py:976-987), in start order            */
    const int64_t *tw_start; /* [n_tw]                                                                              */
    const int32_t *pool;     /* [E][n_tw] skip spans per (endpoint, window) after water-filling, <= 128             */
    const double *dist;      /* [(E+1)][(E+1)][2] (mean, std) of the delay between two endpoints' spans, NaN where  */
                             /* no sample exists; index 0 = the incoming endpoint, 1 + e = outgoing endpoint e      */
} tw_skip_unit;

/* Copies the batch into HBM (span arrays: 16 B per span).  If `spans_on_device` is non-zero the
 * four span arrays are *device* pointers (already resident, e.g. produced by a device-side loader)
 * and are copied device-to-device; the small descriptor arrays are always host pointers.  A skip-mode batch
 * (tw_batch.skip) takes host span arrays only: TW_ERR_UNSUPPORTED otherwise. */
int tw_load_batch(tw_engine *e, const tw_batch *b, int spans_on_device);

/* Pass 1 (traceweaver_v3.py:1159-1219, iteration 0): windows (CreateWindows2, :1020-1078),
 * per-100-span Gaussian parameters (ComputeEpPairDistParams3, :580-646), candidate enumeration
 * + log-likelihood + top-5 (FindTopKAssignments, :180-465; ScoreAssignmentAsPerInvocationGraph,
 * traceweaver_v1.py:259-361), exact per-window selection (GetAssignmentsMIS, :1237-1281 +
 * Gurobi_MIS :1395-1419) and commit with span consumption (AddAssignment, traceweaver_v1.py:433-463). */
int tw_run_pass1(tw_engine *e);

/* Gap samples implied by the pass-1 assignment, one row of n_in doubles per scored slot of each
 * unit (ComputeEpPairDistParams5's `durations`, traceweaver_v3.py:717-762); entries of unassigned
 * incoming spans and rows of unscored slots are NaN.  Slot order per unit (nslot = E*E + 2E):
 *   [0,E) root(in->e) | E + p*E + e primary(p->e) | E + E*E + e closing(e->in).
 * `gaps` holds sum_u nslot_u * n_in_u doubles, unit after unit. */
int tw_get_gaps(tw_engine *e, double *gaps);

/* The other direction (layout of tw_get_gaps): gap samples for tw_fit_mixtures that were not produced by this engine's
 * pass 1 -- a service whose requests were solved in parts (on several GPUs) is refitted on the union of the parts'
 * samples, exactly as the reference fits one mixture per edge over the whole service (traceweaver_v3.py:1221-1222):
 * load a unit of the service's shape, hand over the gathered rows, tw_fit_mixtures, tw_get_mixtures
 * (traceweaver_amd/sharding.py).  The fit depends on the samples of a row *in request order* (k-means++ seeding): parts that are
 * consecutive stretches of the service's requests are concatenated in that order. */
int tw_set_gaps(tw_engine *e, const double *gaps);

/* The engine's own device buffers of the two exchange steps of a sharded run, so that a collective
 * (ncclAllGather = RCCL over xGMI) works on them in place -- no copy through host memory:
 *   parent  int32 [sum_u E_u * n_in_u]          parent indices of the last pass run, unit after unit, [E][n_in] per unit
 *                                               (the layout of tw_results.parent)
 *   gaps    double [sum_u nslot_u * n_in_u]     gap rows of the pass-1 assignment (the layout of tw_get_gaps)
 * Valid until the next tw_load_batch; every entry point returns with the engine's stream synchronised, so the buffers
 * are complete when the caller reads them.  tw_set_gaps_device is tw_set_gaps with a device pointer (the gathered rows
 * of a split service stay in HBM on their way into the refit). */
typedef struct {
    void *parent;
    int64_t parent_count;
    void *gaps;
    int64_t gaps_count;
    int32_t device;
} tw_device_view;
int tw_device_buffers(tw_engine *e, tw_device_view *out);
int tw_set_gaps_device(tw_engine *e, const double *dev_gaps);

/* Mixtures for pass 2 (the fitted sklearn GaussianMixture objects of traceweaver_v3.py:784-786):
 * mix_n[sum_u nslot_u] components per slot (0 = "(0,0)" fallback, traceweaver_v3.py:765-766),
 * mix_p[sum_u nslot_u][TW_MAX_COMP][3] = weight, mean, precision_cholesky. */
int tw_set_mixtures(tw_engine *e, const int32_t *mix_n, const double *mix_p);

/* Device-side alternative to tw_get_gaps + host fit + tw_set_mixtures: the reference's refit of every scored edge
 * (ComputeEpPairDistParams5, traceweaver_v3.py:764-786) on the GPU, from the pass-1 gap samples: for n = 1..min(5, #unique)
 * `GaussianMixture(n, covariance_type="diag").fit` -- k-means++ seeding on the samples in request order, Lloyd, EM (tol 1e-3,
 * <= 100 iterations, reg_covar 1e-6) --, the n of smallest BIC among the fits that did not raise, then the full-covariance
 * refit with `random_state=100`; see traceweaver_amd/csrc/tw_fit.h.  The only randomness are the uniforms the k-means++
 * seeding draws: 1, 3, 7, 10, 13 doubles for n = 1..5, n after n per edge (TW_FIT_ROW_TAPE = 34 for a row with >= 5 distinct
 * samples), independent of the data.
 *
 * tw_fit_mixtures draws them from the engine's own MT19937 (numpy's RandomState algorithm; tw_set_fit_seed reseeds it,
 * tw_create seeds it with 0, tw_load_batch restarts it; one 34-double block per slot and call): the reference's procedure on *a* random stream, as
 * the reference itself runs it unseeded (SURVEY.md hazard H9).  tw_fit_mixtures_seeded gives every unit a stream of its own,
 * MT19937(unit_seed[u]): a unit's fit is then the same whichever units share its batch (services sharded over GPUs).
 * tw_fit_mixtures_tape takes them from the caller: slot q reads tape[slot_off[q] ...], as many doubles as its fits draw
 * (tw_fit_rows tells: max_n[q] = min(5, #unique) of the row, 0 for a row without samples or an unscored slot; the fits of
 * a row draw 1, 4, 11, 21, 34 doubles for max_n = 1..5).  Handing over the doubles numpy's global RNG produces at that point
 * of a seeded reference run makes the refit the one of that run (traceweaver_amd/predictor.py). */
#define TW_FIT_ROW_TAPE 34
int tw_fit_mixtures(tw_engine *e);
int tw_fit_mixtures_seeded(tw_engine *e, const uint32_t *unit_seed);
int tw_set_fit_seed(tw_engine *e, uint32_t seed);
int tw_fit_rows(tw_engine *e, int32_t *max_n);
int tw_fit_mixtures_tape(tw_engine *e, const double *tape, int64_t tape_len, const int64_t *slot_off);

/* The mixture tables currently resident (layout of tw_set_mixtures). */
int tw_get_mixtures(tw_engine *e, int32_t *mix_n, double *mix_p);

/* Pass 2 (iteration 1): same as pass 1 with GaussianMixture.score terms
 * (traceweaver_v1.py:125-126), reusing the windows. */
int tw_run_pass2(tw_engine *e);

/* Results of pass `pass` (1 or 2), host buffers, any pointer may be NULL.  Per-unit arrays are
 * concatenated unit after unit; within a unit [E][n_in] / [TW_TOPK][E][n_in] / [TW_TOPK][n_in].
 *   parent      int32  chosen outgoing-span index per endpoint, -1 = ("NA","NA"), -2 = ("Skip","Skip")
 *   topk_idx    int32  top-5 tuples on *all* spans (top_k_2, traceweaver_v3.py:1185), -1 padded; skip mode: a skip span
 *                      is -(TW_SKIP_BASE + time window * TW_SKIP_STRIDE + position in the window's pool)
 *   topk_score  double their scores (NaN padded)
 *   topk_n      int32  [n_in] number of valid tuples
 *   chosen      int32  [n_in] rank of the chosen tuple in the span's own candidate list, -1 none
 *   leaves      int64  [n_in] enumerated tuples of the top_k call (per_span_candidates increment)
 *   window_end  uint8  [n_in] 1 where a window closes (traceweaver_v3.py:1192)
 *   unit_stats  int64  [n_units][8]: not_best_count, cnt_unassigned, n_windows, windows re-solved
 *                      because an earlier window consumed one of their candidates, windows whose
 *                      exact selection search hit its node budget (incumbent returned), search nodes /
 *                      states of the selection, windows solved by k_select_dp (a level outgrew the small
 *                      tables of k_select_heavy), components it handed on to the depth-first search */
typedef struct {
    int32_t *parent;
    int32_t *topk_idx;
    double *topk_score;
    int32_t *topk_n;
    int32_t *chosen;
    int64_t *leaves;
    uint8_t *window_end;
    int64_t *unit_stats;
} tw_results;
int tw_get_results(tw_engine *e, int pass, const tw_results *r);

/* Pass-1 Gaussian parameters: gauss[sum_u nblk_u * nslot_u][3] = mean, std, log(std_used);
 * NaN for unscored slots.  nblk_u = ceil(n_in_u / batch_size). */
int tw_get_gauss_params(tw_engine *e, double *gauss);

/* Timing of the last pass, measured with HIP events on the engine's stream:
 * ms[0] whole pass, ms[1] candidate-enumeration kernel, ms[2] selection kernel,
 * ms[3] window construction, ms[4] claim/detect/repair kernels, ms[5] parameter kernels (sort + block
 * sums), ms[6] last tw_fit_mixtures call, ms[7] (a count, not a time) rounds of the span-consumption fixed point,
 * ms[8] / ms[9] host wall clock the last pass spent submitting its first enumeration / in the whole call (a pass of a small
 * batch is bound by that, not by the kernels). */
int tw_get_timing(tw_engine *e, double *ms, int32_t n);

/* ---- neighbours of the hot path on the same device arrays -------------------------------
 *
 * Replaces: FindOrder (executor.py:214-285), the call-order DAG the scorer is given: dag[a*E+b] = 1 iff in every
 * request the true call to endpoint a ended no later than the true call to endpoint b started.  Works on units
 * *before* they are loaded (endpoints in any order, e.g. partition-key order): unit_in_off / unit_E / ep_off /
 * out_start / out_end as in tw_batch, true_child[sum_u E_u * n_in_u] = per unit [E][n_in] index of the true
 * outgoing span of request i at endpoint e in that endpoint's list (< 0 = unknown).  dag_out[sum_u E_u*E_u]. */
int tw_find_order(tw_engine *e, int32_t n_units, const int64_t *unit_in_off, const int32_t *unit_E,
                  const int64_t *ep_off, const int64_t *out_start, const int64_t *out_end,
                  const int32_t *true_child, uint8_t *dag_out);

/* Ground truth of the loaded batch for tw_evaluate: true_child in the layout of tw_results.parent;
 * in_trace[n_in_total] (may be NULL) = trace number in [0, n_traces) of every incoming span, for the per-trace
 * (end-to-end) accuracy.  Call after tw_load_batch. */
int tw_set_truth(tw_engine *e, const int32_t *true_child, const int32_t *in_trace, int64_t n_traces);

/* Replaces: repeat_change_spans (helpers/transforms.py:10-40) -- the load scaling `--compress_factor N` applies to every
 * service before the predictor runs (executor.py:1086-1097,1146-1148) -- on the RESIDENT span table, so that the load levels
 * of one corpus (exps/exp5/run_experiment.sh:60-156 runs six per call graph) share one upload.  Needs an integer-microsecond
 * batch (no unit_time_scale, no skip mode) and tw_set_truth (the requests' own calls pair the spans of a request as the
 * reference's sort by trace id does, helpers/transforms.py:25-29; every endpoint must hold exactly one call per request,
 * otherwise TW_ERR_ARG -- the reference asserts).  Per request with load factor f = unit_factor[u] >= 1 (executor.py:1089-1091):
 * x = in.start / f in binary64, out.start = x + (out.start - in.start), durations kept; every list is re-sorted by
 * (start, end), ties in the order of trace_rank[n_in_total] (rank of the request's trace id inside its unit; NULL = current
 * order); truth and trace numbers follow the spans.  The scaled binary64 timestamps are kept exactly, as int64 multiples of
 * unit_time_scale[u] = 2^-k (written if not NULL; semantics of tw_batch.unit_time_scale).  in_perm[n_in_total] /
 * out_perm[n_out_total] (may be NULL) = list-local old index of every new position.  Always scales the table as uploaded
 * (calls do not compound); afterwards the engine is in the state tw_load_batch leaves it in. */
int tw_scale_load(tw_engine *e, const int32_t *unit_factor, const int32_t *trace_rank, int32_t *in_perm, int32_t *out_perm,
                  double *unit_time_scale);

/* The reference's baseline predictors on the resident table (executor.py:888-900 indices 3, 4, 7 -- what
 * exps/exp1 and exps/exp5 run next to predictor 10), see traceweaver_amd/csrc/tw_baselines.h.  parent_out in the layout of
 * tw_results.parent.  No-skip batches only (the lists of a skip-mode batch are not sorted: TW_ERR_UNSUPPORTED).
 *   tw_run_baseline(TW_BASELINE_FCFS)   algorithms/fcfs.py:10-26: request i takes the i-th call of every endpoint.
 *   tw_run_baseline(TW_BASELINE_VPATH)  algorithms/vpath.py:48-89; needs tw_set_truth (a returning call is followed to the request
 *                                        it truly belongs to, as the reference does through the trace id).
 *   WAP5 (algorithms/wap5.py:257-351) in two steps, because its delay samples accumulate per callee *name* over the services of
 *   a run (the class keeps them; the names live on the host): tw_wap5_delays returns per (unit, endpoint) -- [n_units][TW_MAX_EP] --
 *   the sum (in the unit's timestamp units) and the number of the delay samples of BuildDistributions (:257-275) and the longest
 *   request per unit; the caller adds them up per name in service order, takes the means (statistics.mean: exact) and hands them,
 *   in microseconds, to tw_wap5_parents, which runs ScoreParents (:282-316).  -2 = several calls were given to the request:
 *   the reference's accuracy counts that as wrong (helpers/utils.py:68-73).  call_request (may be NULL; layout of out_start):
 *   the request every call was given to, -1 = none -- the reference's list-valued assignment is its inverse. */
#define TW_BASELINE_FCFS 4
#define TW_BASELINE_VPATH 7
int tw_run_baseline(tw_engine *e, int kind, int32_t *parent_out);
int tw_wap5_delays(tw_engine *e, int64_t *delay_sum, int32_t *delay_cnt, int64_t *unit_maxdur);
int tw_wap5_parents(tw_engine *e, const double *mean, int32_t *parent_out, int32_t *call_request);

/* Replaces: AccuracyForService / TopKAccuracyForService (helpers/utils.py:62-97) and AccuracyEndToEnd /
 * TopKAccuracyEndToEnd (helpers/utils.py:99-145) as reductions over the resident results of the last pass run.
 * per_unit[n_units][4] = requests, requests with every endpoint right, requests whose top-5 list holds the true
 * tuple, requests left unassigned.  trace_flags (may be NULL; needs in_trace): [2][n_traces] uint8, 1 = some span of
 * the trace is wrong (exact / top-5) -- combine across ranks with a MAX all-reduce; e2e[2] (may be NULL) = traces
 * right under the exact / the top-5 criterion on this engine alone. */
int tw_evaluate(tw_engine *e, int64_t *per_unit, uint8_t *trace_flags, int64_t *e2e);

/* Replaces: the sweep of BuildDistributions (traceweaver_v3.py:120-169).  The spans of one service merged in start
 * order (stable: incoming spans first, then the endpoints in order): start / dur [n], ep [n] (0 = incoming span,
 * 1 + e = span of outgoing endpoint e), large_delay = the longest incoming span.  Per span: key_out = a * (E + 1) + b
 * of the pair its delay sample belongs to (-1: no qualifying predecessor) and the sample in microseconds.  The
 * (mean, std) per pair are np.mean / np.std of the samples in this order (host, traceweaver_amd/skipmode.py); the
 * durations of the incoming spans form the pair (0, 0), which the scorer never reads. */
int tw_build_distributions(tw_engine *e, int64_t n, const int64_t *start, const int64_t *dur, const uint8_t *ep, int64_t large_delay,
                           int32_t E, int32_t *key_out, int64_t *sample_out);

/* Measurement aid for the roofline figures (bench.py): rate of a plain 16-B-per-lane streaming copy kernel over
 * `bytes` of HBM (read + written bytes per second, in GB/s), `iters` launches timed with HIP events. */
int tw_measure_hbm_copy(tw_engine *e, int64_t bytes, int32_t iters, double *gbps);

/* Page-locked host buffers for the arrays of a tw_batch / tw_results: tw_load_batch and tw_get_results move pinned
 * memory at the full host-link rate (pageable memory is staged by the runtime at a fraction of it). */
int tw_host_alloc(int64_t bytes, void **out);
void tw_host_free(void *p);

/* One-shot convenience for a single unit: load, pass 1, (optional) pass 2 with caller-supplied
 * mixtures, results of the last pass run.  mix_n == NULL => pass 1 only. */
int tw_assign_service(tw_engine *e, int32_t n_in, const int64_t *in_start, const int64_t *in_end,
                      int32_t E, const int64_t *out_off, const int64_t *out_start, const int64_t *out_end,
                      const uint8_t *dag, const int32_t *key_rank, const int32_t *mix_n,
                      const double *mix_p, const tw_results *r);

/* ---- Jaeger-JSON ingest -> structure of arrays (host side) ------------------------------------
 *
 * Replaces the reference's loader for corpora that need no span rewriting: GetAllTracesInDir / TimeOrder
 * (executor.py:287-339), ParseJsonTrace / ParseSpansJson / ParseProcessesJson(2) (executor.py:342-384,451-461,
 * 755-793), ProcessTraceData (executor.py:795-849), PartitionSpansByEndPoint (executor.py:1104-1113),
 * GetGroundTruth (helpers/utils.py:22-32), FindOrder + nx.topological_sort (executor.py:214-285,
 * traceweaver_v1.py:37-39), FixSpans / FixSpans2 (executor.py:505-537,542-645) and the rewrite of the Alibaba parser
 * output (executor.py:377-448).  Names are interned: every *_name / service / span_id field below is a string id for
 * tw_corpus_string().  String ids are handles, not indices of one dense table: service / operation / endpoint names
 * count up from 0 (in order of first use, traces in time order -- the same for any number of parser threads), trace
 * ids are 2^29 + trace number, span ids 2^30 + span-table row (unique per trace: stored, never looked up). */
typedef struct tw_corpus tw_corpus;
int tw_corpus_create(tw_corpus **out);
void tw_corpus_destroy(tw_corpus *c);
const char *tw_corpus_last_error(const tw_corpus *c);   /* first file that failed to parse, if any */

/* Parses one-trace-per-file Jaeger JSON ({"data":[{"traceID","spans":[...],"processes":{...}}]}, or the
 * requestType shape of alibaba-analysis/real-parser.py:308-359), n_threads parser threads (<= 0: up to 32, one per 64 files),
 * orders the traces by the start of their root span and adds those whose root operation is `first_span`
 * (NULL / "" = any; executor.py:757-763,841) until max_traces (> 0; the reference stops at 1001,
 * executor.py:873) traces are held.  Files that do not parse or break an assumption the reference asserts on
 * are counted, not fatal. */
int tw_corpus_add_files(tw_corpus *c, const char *const *paths, int32_t n_paths, const char *first_span,
                        int64_t max_traces, int32_t n_threads, int32_t fix);

/* `fix` = the span surgery the reference applies to some of its corpora before the walk (executor.py:776-779):
 *   TW_FIX_NONE          plain Jaeger exports (hotel; --fix 2..4)
 *   TW_FIX_CLIENT_TWINS  FixSpans (executor.py:505-537; nodejs, --fix 0): every hop is logged once, as a server span;
 *                        each gets a client twin "<id>_client" in the calling service, which is looked up in the
 *                        static service -> caller map given with tw_corpus_set_callers (executor.py:109-115)
 *   TW_FIX_REROOT        FixSpans2 (executor.py:542-645; media, --fix 1): the span named first_span becomes the root,
 *                        same-process children are dropped, every remaining hop gets a client twin in its parent's
 *                        process, spans are ordered by start time
 *   TW_FIX_RPC_TWINS     ParseSpansJson with first_span == None (executor.py:377-448; --fix 5, the output of
 *                        alibaba-analysis/real-parser.py): every call is logged twice under one rpc id, as a server record in
 *                        the callee and a client record in the caller; the client record becomes "<rpc id>.client" and
 *                        the server record is re-pointed at it; a service that calls itself gets a stand-in callee
 *                        "<callee>@<rpc id>-loop" (the reference draws a random name ending in "-loop"), one per rpc id
 *                        for the whole corpus, traces visited in time order; traces in which a child span is not
 *                        contained in its parent are dropped (counted as filtered) */
enum { TW_FIX_NONE = 0, TW_FIX_CLIENT_TWINS = 1, TW_FIX_REROOT = 2, TW_FIX_RPC_TWINS = 3 };
/* serviceLoopMap (executor.py:396): the service a "...-loop" stand-in was split from (its replica count is what the
 * load factor of executor.py:1092-1096 uses), NULL if `service` is not a stand-in. */
const char *tw_corpus_loop_origin(const tw_corpus *c, const char *service);
/* The static service -> caller map of TW_FIX_CLIENT_TWINS. */
int tw_corpus_set_callers(tw_corpus *c, const char *const *service, const char *const *caller, int32_t n);

/* out6 = spans held, traces held, files seen, files that failed to parse, traces filtered out, strings (names + trace
 * ids + span ids held).  tw_corpus_string returns NULL for an id that names nothing. */
int tw_corpus_counts(const tw_corpus *c, int64_t *out6);
const char *tw_corpus_string(const tw_corpus *c, int32_t id);
int tw_corpus_trace_names(const tw_corpus *c, int32_t *out);   /* string id of the traceID of trace 0 .. traces-1 */

/* The span table (one row per span reached from its trace's root, in walk order); caller-allocated columns of
 * tw_corpus_counts()[0] entries, any may be NULL.  parent = row of the referenced span (-1 for roots),
 * kind: 1 server (incoming), 2 client (outgoing). */
typedef struct {
    int32_t *trace, *span_id, *service, *op_name, *parent;
    int64_t *start, *duration;
    uint8_t *kind;
} tw_span_table;
int tw_corpus_span_table(const tw_corpus *c, tw_span_table *t);

/* Every service with a single caller whose endpoints each hold one call per request becomes a unit, in the
 * layout tw_load_batch takes (first nine fields = tw_batch's); pointers stay owned by the corpus and are valid
 * until the next call.  true_child / in_trace as in tw_find_order / tw_set_truth; in_row / out_row = span-table
 * row of every incoming / outgoing span (to translate results back to (trace id, span id)); skipped[4] =
 * services left out because they have several callers (executor.py:1126-1128), because an endpoint's span count
 * differs from the number of requests (skip mode) or E > TW_MAX_EP, because they hold < 2 requests, because the
 * call-order relation has a cycle. */
typedef struct {
    int32_t n_units;
    const int64_t *unit_in_off;
    const int32_t *unit_E;
    const int64_t *ep_off;
    const uint8_t *dag;
    const int32_t *key_rank;
    const int64_t *in_start, *in_end, *out_start, *out_end;
    const int32_t *true_child, *in_trace, *in_row, *out_row;
    const int32_t *unit_service, *ep_name, *in_ep_name;
    const int32_t *unit_order;   /* position of the unit's service among the services that make calls (the executor's process_id, executor.py:1079) */
    int64_t n_traces;
    int32_t skipped[4];
} tw_unit_set;
int tw_corpus_build_units(tw_corpus *c, tw_unit_set *out);

#ifdef __cplusplus
}
#endif
#endif /* TRACEWEAVER_AMD_H */
