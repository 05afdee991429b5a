// tw_ingest.cpp -- Jaeger-JSON ingest into the engine's structure-of-arrays form (host side of libtwgpu.so;
// SURVEY.md 8 f1).  Replaces the reference's Python loader:
//   GetAllTracesInDir / TimeOrder            executor.py:287-339   traces ordered by the start of their root span
//   ParseJsonTrace / ParseSpansJson          executor.py:342-384,755-793   one trace per file, {"data":[{traceID, spans, processes}]}
//   ParseProcessesJson / ParseProcessesJson2 executor.py:451-461   processID -> serviceName (or the id itself for requestType data)
//   ProcessTraceData                         executor.py:795-849   pre-order walk from the root, children by start time;
//                                                                  server spans = incoming, client spans = outgoing of their service
//   PartitionSpansByEndPoint                 executor.py:1104-1113 by caller ("client_<op>" for roots) / callee service, sorted (start, end)
//   GetGroundTruth                           helpers/utils.py:22-32 first outgoing span of the same trace per endpoint
//   FindOrder + nx.topological_sort          executor.py:214-285, traceweaver_v1.py:37-39
//   FixSpans / FixSpans2                     executor.py:505-537,542-645   the span surgery the nodejs / media corpora need
//   ParseSpansJson, first_span == None       executor.py:377-448   the rewrite of the Alibaba parser's output (--fix 5): client twins,
//                                                                  self-call stand-in services, containment filter
//
// Strings never reach the GPU: every name is interned, units carry string ids and span-table rows so that the
// caller can translate indices back to (trace id, span id) keys.
//
// Work per call of tw_corpus_add_files, one crew of parser threads throughout: (1) per trace, any thread: read, scan
// (keys and compared values are views of the file's text, nothing the loader does not read is copied), span surgery,
// tree walk, the trace's names numbered in the thread's own name table; (2) one thread: time order, trace limit, first
// row of every accepted trace, names interned in trace order (a few per trace; ids do not depend on the thread count);
// (3) per trace, the thread that parsed it: rows written, parse structures released (no two threads meet in one
// allocator arena); (4) one thread, streaming over the new rows: per-service row lists.  tw_corpus_build_units builds
// the services side by side.  Measured: profiles/r02b_ingest_sweep.json.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <limits>
#include <mutex>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <numeric>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include <cerrno>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "traceweaver_amd.h"

namespace {

// ------------------------------------------------------------------------------------------------
// JSON scanner (RFC 8259): no document tree -- the loader walks the text once, copies out the handful of fields it
// needs and skips everything else (tags it does not read, logs, warnings) without allocating.
struct Sv {   // a piece of text that lives elsewhere (the file's text, or a scratch string for the rare escaped value)
    const char* p = nullptr;
    size_t n = 0;
    template <size_t N> bool is(const char (&lit)[N]) const { return n == N - 1 && memcmp(p, lit, N - 1) == 0; }
    bool same(const Sv& o) const { return n == o.n && memcmp(p, o.p, n) == 0; }
    bool same(const std::string& o) const { return n == o.size() && memcmp(p, o.data(), n) == 0; }
    std::string text() const { return std::string(p, n); }
};

struct Scan {
    const char* p;
    const char* end;
    std::string err;
    bool fail(const char* m) { if (err.empty()) err = m; return false; }
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    bool eat(char c) { ws(); if (p < end && *p == c) { p++; return true; } return false; }
    bool expect(char c, const char* m) { return eat(c) ? true : fail(m); }
    static void utf8(std::string& out, unsigned cp) {
        if (cp < 0x80) out.push_back((char)cp);
        else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
        else { out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
    }
    bool hex4(unsigned& v) {
        if (end - p < 4) return fail("truncated \\u escape");
        v = 0;
        for (int k = 0; k < 4; k++) {
            const char c = *p++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
            else return fail("bad \\u escape");
        }
        return true;
    }
    // a string value as a view: into the text when it has no escapes (the usual case), into `scratch` otherwise
    bool strv(Sv& out, std::string& scratch) {
        ws();
        if (p >= end || *p != '"') return fail("expected string");
        const char* q = p + 1;
        while (q < end && *q != '"' && *q != '\\') q++;
        if (q >= end) return fail("unterminated string");
        if (*q == '"') { out.p = p + 1; out.n = (size_t)(q - p - 1); p = q + 1; return true; }
        if (!str(&scratch)) return false;
        out.p = scratch.data(); out.n = scratch.size();
        return true;
    }
    bool skip_str() {  // positioned at the opening quote
        const char* q = p + 1;
        while (q < end && *q != '"') q += (*q == '\\') ? 2 : 1;
        if (q >= end) return fail("unterminated string");
        p = q + 1;
        return true;
    }
    // a string value; `out` == nullptr just skips it
    bool str(std::string* out) {
        ws();
        if (p >= end || *p != '"') return fail("expected string");
        if (out == nullptr) {
            const char* q0 = p;
            if (!skip_str()) return false;
            // escapes are validated like everywhere else (a malformed one rejects the file): rescan when there is one
            if (memchr(q0, '\\', (size_t)(p - q0)) == nullptr) return true;
            p = q0;
        }
        p++;
        if (out) out->clear();
        while (true) {
            const char* q = p;
            while (q < end && *q != '"' && *q != '\\') q++;   // plain run
            if (q >= end) return fail("unterminated string");
            if (out) out->append(p, q);
            p = q;
            if (*p == '"') { p++; return true; }
            p++;  // backslash
            if (p >= end) return fail("truncated escape");
            const char c = *p++;
            char lit = 0;
            switch (c) {
                case '"': lit = '"'; break;
                case '\\': lit = '\\'; break;
                case '/': lit = '/'; break;
                case 'b': lit = '\b'; break;
                case 'f': lit = '\f'; break;
                case 'n': lit = '\n'; break;
                case 'r': lit = '\r'; break;
                case 't': lit = '\t'; break;
                case 'u': {
                    unsigned cp;
                    if (!hex4(cp)) return false;
                    if (cp >= 0xD800 && cp < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                        p += 2;
                        unsigned lo;
                        if (!hex4(lo)) return false;
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    if (out) utf8(*out, cp);
                    continue;
                }
                default: return fail("bad escape");
            }
            if (out) out->push_back(lit);
        }
    }
    // a number as int64 (exact when it is written as an integer, truncated toward zero otherwise)
    bool i64(int64_t& out) {
        ws();
        const char* q = p;
        bool integral = true;
        if (q < end && *q == '-') q++;
        if (q >= end || *q < '0' || *q > '9') return fail("expected number");
        const char* d0 = q;
        uint64_t acc = 0;
        while (q < end && *q >= '0' && *q <= '9') acc = acc * 10 + (uint64_t)(*q++ - '0');
        if (q < end && (*q == '.' || *q == 'e' || *q == 'E')) {
            integral = false;
            while (q < end && ((*q >= '0' && *q <= '9') || *q == '.' || *q == 'e' || *q == 'E' || *q == '+' || *q == '-')) q++;
        }
        if (integral && q - d0 <= 18) out = *p == '-' ? -(int64_t)acc : (int64_t)acc;   // no overflow below 10^18
        else {
            const std::string tok(p, q);
            out = (integral && tok.size() <= 19) ? strtoll(tok.c_str(), nullptr, 10) : (int64_t)strtod(tok.c_str(), nullptr);
        }
        p = q;
        return true;
    }
    bool skip(int depth = 0) {  // any value
        ws();
        if (p >= end) return fail("unexpected end of input");
        if (depth > 256) return fail("nesting too deep");
        const char c = *p;
        if (c == '"') return str(nullptr);
        if (c == '{' || c == '[') {
            const char close = c == '{' ? '}' : ']';
            p++;
            if (eat(close)) return true;
            while (true) {
                if (c == '{') { if (!str(nullptr) || !expect(':', "expected ':'")) return false; }
                if (!skip(depth + 1)) return false;
                if (eat(',')) continue;
                return expect(close, c == '{' ? "expected ',' or '}'" : "expected ',' or ']'");
            }
        }
        if (c == 't' && end - p >= 4 && !memcmp(p, "true", 4)) { p += 4; return true; }
        if (c == 'f' && end - p >= 5 && !memcmp(p, "false", 5)) { p += 5; return true; }
        if (c == 'n' && end - p >= 4 && !memcmp(p, "null", 4)) { p += 4; return true; }
        int64_t dummy;
        if (c == '-' || (c >= '0' && c <= '9')) return i64(dummy);
        return fail("unexpected character");
    }
    // for (auto key : object): calls f(key) positioned at the value; f must consume the value
    template <class F>
    bool object(F&& f) {
        if (!expect('{', "expected object")) return false;
        if (eat('}')) return true;
        std::string escaped;   // per nesting level: a key stays valid while its value is scanned
        Sv key;
        while (true) {
            if (!strv(key, escaped) || !expect(':', "expected ':'")) return false;
            if (!f(key)) return false;
            if (eat(',')) continue;
            return expect('}', "expected ',' or '}'");
        }
    }
    template <class F>
    bool array(F&& f) {  // f() consumes one element
        if (!expect('[', "expected array")) return false;
        if (eat(']')) return true;
        while (true) {
            if (!f()) return false;
            if (eat(',')) continue;
            return expect(']', "expected ',' or ']'");
        }
    }
    bool is_string() { ws(); return p < end && *p == '"'; }
};

// ------------------------------------------------------------------------------------------------
struct SpanTmp {
    std::string sid, op, pid;
    std::string caller, callee;  // upstream / downstream service of the call (alibaba-analysis/real-parser.py:318,320)
    bool has_caller = false, has_callee = false;
    std::vector<std::pair<std::string, std::string>> refs;  // (traceID, spanID)
    int64_t start = 0, dur = 0;
    int kind = 0;  // 1 server, 2 client
};
struct TraceTmp {
    bool ok = false;
    std::string error, trace_id;
    std::vector<SpanTmp> spans;
    std::vector<std::pair<std::string, std::string>> processes;  // pid -> service
    bool request_type = false;
    double root_start = 0.0;  // TimeOrder key (executor.py:287-317); +inf when the file has no root span
    bool has_root = false;
};

// ParseJsonTrace + ParseSpansJson (executor.py:342-384,755-793): one trace per file
void parse_trace(const char* buf, size_t len, TraceTmp& T) {
    Scan S{buf, buf + len, std::string()};
    int n_traces = 0;
    bool have_tid = false, have_spans = false, have_procs = false, bad_span = false, first_span = true;
    std::string why, esc_a, esc_b;
    std::vector<Sv> span_tids;  // the traceID key of the trace may come after its spans in the text (views: unescaped text only)
    std::vector<std::string> span_tids_esc;  // the rare escaped ones, by value
    const bool ok = S.object([&](const Sv& k0) {
        if (!k0.is("data")) return S.skip();
        return S.array([&]() {
            if (++n_traces > 1) return S.skip();
            return S.object([&](const Sv& k1) {
                if (k1.is("traceID")) { have_tid = S.is_string(); return have_tid ? S.str(&T.trace_id) : S.skip(); }
                if (k1.is("processes")) {
                    S.ws();
                    if (S.p >= S.end || *S.p != '{') return S.skip();
                    have_procs = true;
                    return S.object([&](const Sv& pid) {
                        std::string name;
                        bool named = false;
                        S.ws();
                        if (S.p >= S.end || *S.p != '{') { if (why.empty()) why = "process without serviceName"; bad_span = true; return S.skip(); }
                        if (!S.object([&](const Sv& k3) {
                                if (k3.is("serviceName") && S.is_string() && !named) { named = true; return S.str(&name); }
                                return S.skip();
                            })) return false;
                        if (!named) { if (why.empty()) why = "process without serviceName"; bad_span = true; }
                        T.processes.emplace_back(pid.text(), std::move(name));
                        return true;
                    });
                }
                if (!k1.is("spans")) return S.skip();
                S.ws();
                if (S.p >= S.end || *S.p != '[') return S.skip();
                have_spans = true;
                T.spans.reserve(16);
                return S.array([&]() {
                    T.spans.emplace_back();
                    SpanTmp& sp = T.spans.back();
                    Sv stid;
                    std::string req;
                    bool has_sid = false, has_pid = false, has_tid = false, has_start = false, has_dur = false, has_op = false, has_req = false;
                    S.ws();
                    if (S.p >= S.end || *S.p != '{') { bad_span = true; if (why.empty()) why = "span without spanID / processID"; T.spans.pop_back(); return S.skip(); }
                    if (!S.object([&](const Sv& k2) {
                            if (k2.n < 4) return S.skip();
                            switch (k2.p[0]) {   // the keys the loader reads; everything else is skipped unread
                            case 's':
                                if (k2.is("spanID") && S.is_string() && !has_sid) { has_sid = true; return S.str(&sp.sid); }
                                if (k2.is("startTime") && !has_start) {
                                    S.ws();
                                    if (!(S.p < S.end && (*S.p == '-' || (*S.p >= '0' && *S.p <= '9')))) return S.skip();
                                    has_start = true;
                                    return S.i64(sp.start);
                                }
                                break;
                            case 'p':
                                if (k2.is("processID") && S.is_string() && !has_pid) { has_pid = true; return S.str(&sp.pid); }
                                break;
                            case 't':
                                if (k2.is("traceID") && S.is_string() && !has_tid) {
                                    has_tid = true;
                                    if (!S.strv(stid, esc_a)) return false;
                                    if (stid.p == esc_a.data()) { span_tids_esc.push_back(esc_a); stid = Sv(); }
                                    return true;
                                }
                                if (k2.is("tags")) {
                                    S.ws();
                                    if (S.p >= S.end || *S.p != '[') return S.skip();
                                    return S.array([&]() {
                                        S.ws();
                                        if (S.p >= S.end || *S.p != '{') return S.skip();
                                        bool is_kind = false, have_val = false;
                                        Sv key_s, val_s;
                                        if (!S.object([&](const Sv& k3) {
                                                if (k3.is("key") && S.is_string()) { if (!S.strv(key_s, esc_a)) return false; is_kind = key_s.is("span.kind"); return true; }
                                                if (k3.is("value") && S.is_string()) { have_val = true; return S.strv(val_s, esc_b); }
                                                return S.skip();
                                            })) return false;
                                        if (is_kind && have_val) sp.kind = val_s.is("server") ? 1 : (val_s.is("client") ? 2 : 3);
                                        return true;
                                    });
                                }
                                break;
                            case 'o':
                                if (k2.is("operationName") && !has_op) { has_op = true; return S.is_string() ? S.str(&sp.op) : S.skip(); }
                                break;
                            case 'c':
                                if (k2.is("caller") && S.is_string() && !sp.has_caller) { sp.has_caller = true; return S.str(&sp.caller); }
                                if (k2.is("callee") && S.is_string() && !sp.has_callee) { sp.has_callee = true; return S.str(&sp.callee); }
                                break;
                            case 'd':
                                if (k2.is("duration") && !has_dur) {
                                    S.ws();
                                    if (!(S.p < S.end && (*S.p == '-' || (*S.p >= '0' && *S.p <= '9')))) return S.skip();
                                    has_dur = true;
                                    return S.i64(sp.dur);
                                }
                                break;
                            case 'r':
                                if (k2.is("requestType") && !has_req) { has_req = true; if (S.is_string()) return S.str(&req); req.clear(); return S.skip(); }
                                if (k2.is("references")) {
                                    S.ws();
                                    if (S.p >= S.end || *S.p != '[') return S.skip();
                                    return S.array([&]() {
                                        Sv rt, rs;
                                        bool hrt = false, hrs = false;
                                        S.ws();
                                        if (S.p >= S.end || *S.p != '{') { bad_span = true; if (why.empty()) why = "malformed reference"; return S.skip(); }
                                        if (!S.object([&](const Sv& k3) {
                                                if (k3.is("traceID") && S.is_string()) { hrt = true; return S.strv(rt, esc_a); }
                                                if (k3.is("spanID") && S.is_string()) { hrs = true; return S.strv(rs, esc_b); }
                                                return S.skip();
                                            })) return false;
                                        if (!hrt || !hrs) { bad_span = true; if (why.empty()) why = "malformed reference"; }
                                        sp.refs.emplace_back(rt.text(), rs.text());
                                        return true;
                                    });
                                }
                                break;
                            default: break;
                            }
                            return S.skip();
                        })) return false;
                    if (!has_sid || !has_pid) { bad_span = true; if (why.empty()) why = "span without spanID / processID"; }
                    else if (!has_tid) { bad_span = true; if (why.empty()) why = "different trace ids for spans in the same trace"; }  // executor.py:367-372
                    else if (!has_start || !has_dur) { bad_span = true; if (why.empty()) why = "span without startTime / duration"; }
                    if (has_req) sp.op = req;  // executor.py:358-361: requestType wins over operationName
                    if (first_span) { T.request_type = has_req; first_span = false; }   // executor.py:774
                    if (sp.refs.empty() && !T.has_root) { T.has_root = true; T.root_start = (double)sp.start; }  // first span without references
                    if (has_tid && stid.p != nullptr) span_tids.push_back(stid);
                    return true;
                });
            });
        });
    });
    if (!ok) { T.error = "JSON: " + S.err; return; }
    if (n_traces != 1) { T.error = "expected exactly one trace under \"data\""; return; }
    if (!have_tid || !have_spans || T.spans.empty()) { T.error = "trace without traceID / spans"; return; }
    if (bad_span) { T.error = why; return; }
    for (const Sv& t : span_tids)
        if (!t.same(T.trace_id)) { T.error = "different trace ids for spans in the same trace"; return; }  // executor.py:367-372
    for (const std::string& t : span_tids_esc)
        if (t != T.trace_id) { T.error = "different trace ids for spans in the same trace"; return; }
    if (T.request_type) {  // ParseProcessesJson2: the process id is the service name
        T.processes.clear();
        for (const SpanTmp& sp : T.spans) T.processes.emplace_back(sp.pid, sp.pid);
    } else if (!have_procs) { T.error = "trace without a process map"; return; }
    T.ok = true;
}

bool read_file(const char* path, std::string& out) {
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 0) { close(fd); return false; }
    out.resize((size_t)st.st_size);
    size_t got = 0;
    while (got < out.size()) {
        const ssize_t r = read(fd, &out[got], out.size() - got);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) break;
        got += (size_t)r;
    }
    close(fd);
    return got == out.size();
}

constexpr int32_t kSidBase = 1 << 30;   // string ids from here on name the span id of row (id - kSidBase)
constexpr int32_t kTraceBase = 1 << 29; // ... and from here to kSidBase the trace id of trace (id - kTraceBase)

struct SpanRow {
    int32_t trace, sid, service, op, parent, n_children, first_child_service;
    int64_t start, dur;
    uint8_t kind;
};

}  // namespace

struct tw_corpus {
    std::string err;
    std::deque<std::string> strings;           // interned names (trace ids, services, operations); references stay valid
    std::unordered_map<std::string_view, int32_t> string_ids;   // keys view `strings`
    std::vector<SpanRow> rows;                 // spans reached from the root of an accepted trace, pre-order
    std::vector<std::string> sids;             // span id of every row: string id kSidBase + row (unique per trace: stored, not looked up)
    std::vector<int32_t> trace_name;           // string id per accepted trace (kTraceBase + trace number)
    std::vector<std::string> trace_ids;        // trace id per accepted trace
    std::vector<int32_t> service_order;        // services in order of their first outgoing span
    std::unordered_map<int32_t, std::vector<int32_t>> in_rows, out_rows;  // per service, in walk order
    int64_t files_total = 0, files_rejected = 0, traces_filtered = 0;
    std::unordered_map<std::string, std::string> caller_of;  // service -> calling service (FixSpans' process_map_1)
    std::unordered_map<std::string, std::string> loop_by_rpc;  // selfLoopMap: rpc id -> stand-in service of a self-call (executor.py:386-399)
    std::unordered_map<std::string, std::string> loop_origin;  // serviceLoopMap: stand-in service -> the service that called itself
    std::unordered_map<std::string, int64_t> loop_seq;         // rpc id -> number (in time order, over all calls) of the trace that entered it
    int64_t traces_seen = 0;                                   // traces handed to the --fix 5 rewrite so far
    // last unit set built
    std::vector<int64_t> u_in_off, u_ep_off, u_in_start, u_in_end, u_out_start, u_out_end;
    std::vector<int32_t> u_E, u_key_rank, u_truth, u_in_trace, u_in_row, u_out_row, u_service, u_ep_name, u_in_ep, u_order;
    std::vector<uint8_t> u_dag;
    int32_t skipped_multi_in = 0, skipped_skip_mode = 0, skipped_small = 0, skipped_cyclic = 0;

    int32_t intern(std::string_view s) {
        auto it = string_ids.find(s);
        if (it != string_ids.end()) return it->second;
        const int32_t id = (int32_t)strings.size();
        strings.emplace_back(s);
        string_ids.emplace(std::string_view(strings.back()), id);
        return id;
    }
};

namespace {

// FixSpans (executor.py:505-537; nodejs corpora, --fix 0): the instrumentation logs every hop once, as a server span
// in the callee (the root as a client span).  Every server span gets a client twin "<sid>_client" with the same
// timestamps in the calling service (static service -> caller map, executor.py:109-115) and is re-pointed at it;
// client spans become server spans.  Twins are appended after the original spans (dict order matters: it breaks
// ties between equal timestamps further down).  Returns false where the reference would raise.
bool fix_client_twins(TraceTmp& T, const std::unordered_map<std::string, std::string>& caller_of) {
    std::unordered_map<std::string, std::string> proc, pid_of_service;
    for (const auto& kv : T.processes) proc[kv.first] = kv.second;
    for (const SpanTmp& s : T.spans) {
        auto it = proc.find(s.pid);
        if (it == proc.end()) return false;
        pid_of_service[it->second] = s.pid;  // process_map_2: the last pid seen per service
    }
    std::vector<SpanTmp> twins;
    for (SpanTmp& s : T.spans) {
        if (s.kind == 2) s.kind = 1;
        else if (s.kind == 1) {
            if (s.refs.empty()) return false;
            auto c = caller_of.find(proc[s.pid]);
            if (c == caller_of.end()) return false;
            auto p = pid_of_service.find(c->second);
            if (p == pid_of_service.end()) return false;
            SpanTmp twin = s;
            twin.sid = s.sid + "_client";
            twin.pid = p->second;
            twin.kind = 2;
            s.refs[0] = std::make_pair(twin.refs[0].first, twin.sid);
            twins.push_back(std::move(twin));
        }
    }
    for (SpanTmp& t : twins) T.spans.push_back(std::move(t));
    return true;
}

// FixSpans2 (executor.py:542-645; media corpora, --fix 1): the span named `root_op` ("ComposeReview") becomes the
// root (its ancestors are dropped, its span id becomes the trace id, its children are re-pointed), spans whose parent
// runs in the same process are dropped, every remaining span becomes a server span and -- unless it is the root --
// gets a client twin "<sid>_client" in its parent's process; finally the spans are ordered by start time (stable).
bool fix_reroot(TraceTmp& T, const std::string& root_op) {
    struct Node { SpanTmp s; std::string key; bool alive; };
    std::vector<Node> nodes;          // insertion order of the reference's dict `new_spans`
    std::unordered_map<std::string, int> at;  // key (span id at insertion) -> index of the live entry
    for (const SpanTmp& s : T.spans) {
        auto it = at.find(s.sid);
        if (it != at.end()) { nodes[(size_t)it->second].s = s; continue; }  // duplicate key: value replaced in place
        at[s.sid] = (int)nodes.size();
        nodes.push_back(Node{s, s.sid, true});
    }
    const int n0 = (int)nodes.size();
    // step 1: re-root
    for (int i = 0; i < n0; i++) {
        if (nodes[(size_t)i].s.op != root_op) continue;
        const std::string old_key = nodes[(size_t)i].key;
        if (nodes[(size_t)i].s.refs.empty()) return false;
        std::string up = nodes[(size_t)i].s.refs[0].second;          // DeleteAncestors
        while (true) {
            auto it = at.find(up);
            if (it == at.end() || !nodes[(size_t)it->second].alive) return false;
            Node& a = nodes[(size_t)it->second];
            // the reference walks the *original* spans' references and deletes from the copy
            a.alive = false;
            at.erase(it);
            if (a.s.refs.empty()) break;
            up = a.s.refs[0].second;
        }
        for (size_t j = 0; j < T.spans.size(); j++) {                // ChangeChildReferences (looks at the original spans)
            const SpanTmp& o = T.spans[j];
            if (!o.refs.empty() && o.refs[0].second == old_key && o.refs[0].first == T.trace_id) {
                auto it = at.find(o.sid);
                if (it == at.end()) return false;
                nodes[(size_t)it->second].s.refs[0] = std::make_pair(T.trace_id, T.trace_id);
            }
        }
        SpanTmp moved = nodes[(size_t)i].s;
        moved.sid = T.trace_id;
        moved.refs.clear();
        nodes[(size_t)i].alive = false;                              // del new_spans[span_id] ...
        at.erase(old_key);
        auto ex = at.find(T.trace_id);                               // ... new_spans[(trace_id, trace_id)] = span
        if (ex != at.end()) nodes[(size_t)ex->second].s = moved;
        else { at[T.trace_id] = (int)nodes.size(); nodes.push_back(Node{moved, T.trace_id, true}); }
    }
    // step 2: drop spans whose parent runs in the same process (parents looked up in the state after step 1)
    {
        std::vector<int> drop;
        for (size_t i = 0; i < nodes.size(); i++) {
            if (!nodes[i].alive || nodes[i].s.refs.empty()) continue;
            auto it = at.find(nodes[i].s.refs[0].second);
            if (it == at.end()) return false;                        // FindParentProcess would raise
            if (nodes[(size_t)it->second].s.pid == nodes[i].s.pid) drop.push_back((int)i);
        }
        for (int i : drop) { nodes[(size_t)i].alive = false; at.erase(nodes[(size_t)i].key); }
    }
    // step 3: everything is a server span; non-roots get a client twin in the parent's process
    std::vector<SpanTmp> out, twins;
    for (size_t i = 0; i < nodes.size(); i++) {
        if (!nodes[i].alive) continue;
        SpanTmp s = nodes[i].s;
        s.kind = 1;
        if (!s.refs.empty()) {
            auto it = at.find(s.refs[0].second);
            if (it == at.end()) return false;
            SpanTmp twin = s;
            twin.sid = s.sid + "_client";
            twin.pid = nodes[(size_t)it->second].s.pid;
            twin.kind = 2;
            s.refs[0] = std::make_pair(twin.refs[0].first, twin.sid);
            twins.push_back(std::move(twin));
        }
        // the dict key stays the id the span was inserted under; the walk below looks spans up by their sid field,
        // which differs from the key only for the re-rooted span (key == sid == trace id there)
        out.push_back(std::move(s));
    }
    for (SpanTmp& t : twins) out.push_back(std::move(t));
    std::stable_sort(out.begin(), out.end(), [](const SpanTmp& a, const SpanTmp& b) { return a.start < b.start; });
    T.spans = std::move(out);
    return true;
}

// ParseSpansJson with first_span == None (executor.py:377-448; the output of alibaba-analysis/real-parser.py, --fix 5):
// every call is logged twice under one rpc id -- a server record in the callee and a client record with the same
// timestamps in the caller.  The client record is renamed "<rpc id>.client" and the server record re-pointed at it;
// a service calling itself gets a stand-in callee "...-loop" (one per rpc id, for the whole corpus: the reference keeps
// the map across traces and visits them in time order), the client spans below such an rpc id move into the process of
// their parent span, and a trace in which a child is not contained in its parent is dropped.
// The reference draws the stand-in names at random (helpers/misc.py:17-19); here they are "<callee>@<rpc id>-loop".
//
// The corpus-wide map is the only thing that ties traces together, and a trace only ever sees the entries made by traces
// up to itself.  So the rewrite runs in three steps: (1) per trace, any thread: the self-calls it would enter
// (self_calls); (2) one thread, in time order: enter them, each with the number of its trace (a few entries per corpus);
// (3) per trace, any thread: the rewrite proper, reading the map as it stood when the trace was reached (entries with a
// number <= its own).  Same result as the sequential visit, with the per-span work spread over the parser threads.
std::string sanitized_rpc(const std::string& sid) {
    return sid.size() >= 7 && sid.compare(sid.size() - 7, 7, ".client") == 0 ? sid.substr(0, sid.size() - 7) : sid;
}

// step 1: (rpc id, callee) of every self-call in span order, up to the span at which the reference would raise
void self_calls(const TraceTmp& T, std::vector<std::pair<std::string, std::string>>& out) {
    out.clear();
    for (const SpanTmp& s : T.spans) {
        if (!s.has_caller || !s.has_callee) return;                    // span["caller"] would raise: the trace is dropped there
        if (s.caller == s.callee) out.emplace_back(s.sid, s.callee);   // (ids are still unsuffixed here: sanitized == sid)
    }
}

// step 2 (sequential, in time order)
void enter_self_calls(tw_corpus* c, const std::vector<std::pair<std::string, std::string>>& calls, int64_t seq) {
    for (const auto& rc : calls) {
        const std::string rpc = sanitized_rpc(rc.first);
        if (c->loop_by_rpc.find(rpc) != c->loop_by_rpc.end()) continue;
        const std::string name = rc.second + "@" + rpc + "-loop";
        c->loop_by_rpc.emplace(rpc, name);
        c->loop_seq[rpc] = seq;
        c->loop_origin[name] = rc.second;
    }
}

// step 3.  Returns false when the trace is dropped.
bool fix_rpc_twins(TraceTmp& T, const tw_corpus* c, int64_t seq) {
    auto visible = [&](const std::string& rpc) -> const std::string* {
        auto it = c->loop_by_rpc.find(rpc);
        if (it == c->loop_by_rpc.end()) return nullptr;
        auto sq = c->loop_seq.find(rpc);
        return sq != c->loop_seq.end() && sq->second <= seq ? &it->second : nullptr;
    };
    for (SpanTmp& s : T.spans) {                                       // step 1
        if (s.kind == 2) s.sid += ".client";
        if (s.kind == 1 && s.refs.size() == 1) s.refs[0].second = s.sid + ".client";
        if (!s.has_caller || !s.has_callee) return false;              // span["caller"] would raise
        if (s.caller == s.callee) {
            const std::string* name = visible(sanitized_rpc(s.sid));   // entered by this trace or an earlier one
            if (name == nullptr) return false;                         // (cannot happen: step 2 ran for this trace)
            s.callee = *name;
            if (s.kind == 1) s.pid = *name;
        }
    }
    const int n = (int)T.spans.size();
    std::unordered_map<std::string, int> by_sid;                       // the dict `spans`: later duplicates replace the value
    for (int i = 0; i < n; i++) by_sid[T.spans[(size_t)i].sid] = i;
    std::vector<std::vector<int>> children((size_t)n);
    int root = -1;
    for (int i = 0; i < n; i++) {                                      // steps 2-3 (dict order = first insertion of each key)
        const SpanTmp& s = T.spans[(size_t)i];
        if (by_sid[s.sid] != i) continue;
        if (s.refs.empty()) { if (root < 0) root = i; continue; }      // next(span for span in spans.values() if span.IsRoot())
        if (s.refs[0].first != T.trace_id) continue;
        auto it = by_sid.find(s.refs[0].second);
        if (it != by_sid.end()) children[(size_t)it->second].push_back(i);
    }
    if (root < 0) return true;                                         // nothing to check; the walk rejects the trace
    std::vector<char> seen((size_t)n, 0);
    std::function<bool(int)> contained = [&](int v) {                  // check_time_constraints
        if (seen[(size_t)v]) return false;
        seen[(size_t)v] = 1;
        const SpanTmp& p = T.spans[(size_t)v];
        for (int ch : children[(size_t)v]) {
            const SpanTmp& q = T.spans[(size_t)ch];
            if (!(p.start <= q.start && p.start + p.dur >= q.start + q.dur)) return false;
            if (!contained(ch)) return false;
        }
        return true;
    };
    if (!contained(root)) return false;
    std::function<void(int)> update = [&](int v) {                     // update_references
        for (int ch : children[(size_t)v]) {
            if (T.spans[(size_t)ch].kind == 2) T.spans[(size_t)ch].pid = T.spans[(size_t)v].pid;
            update(ch);
        }
    };
    std::function<void(int)> traverse = [&](int v) {                   // traverse_and_update
        if (visible(sanitized_rpc(T.spans[(size_t)v].sid)) != nullptr) update(v);
        for (int ch : children[(size_t)v]) traverse(ch);
    };
    traverse(root);
    T.processes.clear();                                               // ParseProcessesJson2 on the rewritten records
    for (const SpanTmp& s : T.spans) T.processes.emplace_back(s.pid, s.pid);
    return true;
}

// ProcessTraceData (executor.py:795-849) for one parsed trace, in two steps: the walk (any thread, touches only the
// trace) and the append to the corpus (one thread, in trace order).  walk_trace returns false (nothing is kept)
// when the trace breaks an assumption the reference asserts on or its root is not `first_span`.
// The distinct names (services, operations) one parser thread has met, in order of first use.  Traces refer to names by
// their number in the table of the thread that parsed them; the append gives every table entry its corpus string id once.
struct NameTable {
    std::deque<std::string> names;                            // references stay valid
    std::unordered_map<std::string_view, int32_t> ids;        // keys view `names`
    std::vector<int32_t> global;                              // corpus string id per entry (-1 = not interned yet)
    int32_t local(std::string_view s) {
        auto it = ids.find(s);
        if (it != ids.end()) return it->second;
        const int32_t id = (int32_t)names.size();
        names.emplace_back(s);
        ids.emplace(std::string_view(names.back()), id);
        return id;
    }
};

struct Walked {
    std::vector<int> order, parent, n_children;      // pre-order; per span (local index): parent, number of children
    std::vector<int> pos;                            // per span: its place in `order` (-1 = not reached)
    // what the append needs, gathered on the parser thread (numbers in that thread's NameTable): the distinct names of the
    // trace in order of first use and, per place in `order`, the span's service, operation and first child's service (-1 = none)
    std::vector<int32_t> names;
    std::vector<int32_t> svc_l, op_l, child_l;
};

// index of spans by span id without a hash table: positions ordered by (id, position); later duplicates win, like the
// dict of the reference
struct SidIndex {
    const std::vector<SpanTmp>* spans = nullptr;
    std::vector<int> ord;
    void build(const std::vector<SpanTmp>& sp) {
        spans = &sp;
        ord.resize(sp.size());
        std::iota(ord.begin(), ord.end(), 0);
        std::sort(ord.begin(), ord.end(), [&](int a, int b) { const int c = sp[(size_t)a].sid.compare(sp[(size_t)b].sid); return c != 0 ? c < 0 : a < b; });
    }
    int find(const std::string& sid) const {   // the last span with this id, -1 if none
        size_t lo = 0, hi = ord.size();
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if ((*spans)[(size_t)ord[mid]].sid.compare(sid) <= 0) lo = mid + 1; else hi = mid; }
        return lo > 0 && (*spans)[(size_t)ord[lo - 1]].sid == sid ? ord[lo - 1] : -1;
    }
};

// process id -> service name; later entries win.  A handful of processes per trace (one per span for requestType data)
struct ProcIndex {
    const std::vector<std::pair<std::string, std::string>>* procs = nullptr;
    std::vector<int> ord;
    void build(const std::vector<std::pair<std::string, std::string>>& pr) {
        procs = &pr;
        ord.clear();
        if (pr.size() <= 12) return;   // searched from the back
        ord.resize(pr.size());
        std::iota(ord.begin(), ord.end(), 0);
        std::sort(ord.begin(), ord.end(), [&](int a, int b) { const int c = pr[(size_t)a].first.compare(pr[(size_t)b].first); return c != 0 ? c < 0 : a < b; });
    }
    const std::string* find(const std::string& pid) const {
        const auto& pr = *procs;
        if (ord.empty()) {
            for (size_t k = pr.size(); k-- > 0;) if (pr[k].first == pid) return &pr[k].second;
            return nullptr;
        }
        size_t lo = 0, hi = ord.size();
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (pr[(size_t)ord[mid]].first.compare(pid) <= 0) lo = mid + 1; else hi = mid; }
        return lo > 0 && pr[(size_t)ord[lo - 1]].first == pid ? &pr[(size_t)ord[lo - 1]].second : nullptr;
    }
};

bool walk_trace(const TraceTmp& T, const std::string& first_span, Walked& W, NameTable& N) {
    const int n = (int)T.spans.size();
    // working arrays of the parser thread, kept between traces (no allocation per trace once they have grown)
    struct Scratch { SidIndex by_sid; ProcIndex proc; std::vector<int> ref_at, ref_to, child_off, child, fill, stack; std::vector<char> seen; };
    static thread_local Scratch X;
    SidIndex& by_sid = X.by_sid;
    by_sid.build(T.spans);
    ProcIndex& proc = X.proc;
    proc.build(T.processes);
    int root = -1;
    // children lists in one array: counted, placed, then ordered by start time (stable)
    std::vector<int>&ref_at = X.ref_at, &ref_to = X.ref_to, &child_off = X.child_off, &child = X.child, &fill = X.fill, &stack = X.stack;
    ref_at.clear(); ref_to.clear();
    child_off.assign((size_t)n + 1, 0);
    W.parent.assign((size_t)n, -1);
    for (int i = 0; i < n; i++) {
        const SpanTmp& s = T.spans[(size_t)i];
        if (by_sid.find(s.sid) != i) continue;  // shadowed duplicate
        if (s.refs.empty()) root = i;      // the last span without references (executor.py:823-825)
        bool first = true;
        for (const auto& r : s.refs) {
            const int to = r.first == T.trace_id ? by_sid.find(r.second) : -1;
            if (to < 0) return false;  // spans[par_id] would raise
            ref_at.push_back(i); ref_to.push_back(to);
            child_off[(size_t)to + 1]++;
            if (first) { W.parent[(size_t)i] = to; first = false; }
        }
    }
    if (root < 0) return false;
    if (!first_span.empty() && T.spans[(size_t)root].op != first_span) return false;  // executor.py:841
    for (int i = 0; i < n; i++) child_off[(size_t)i + 1] += child_off[(size_t)i];
    child.resize((size_t)child_off[(size_t)n]);
    fill.assign(child_off.begin(), child_off.end() - 1);
    for (size_t k = 0; k < ref_at.size(); k++) child[(size_t)fill[(size_t)ref_to[k]]++] = ref_at[k];
    for (int i = 0; i < n; i++) {
        const int a = child_off[(size_t)i], b = child_off[(size_t)i + 1];
        if (b - a > 1) std::stable_sort(child.begin() + a, child.begin() + b, [&](int x, int y) { return T.spans[(size_t)x].start < T.spans[(size_t)y].start; });
    }
    stack.assign(1, root);
    std::vector<char>& seen = X.seen;
    seen.assign((size_t)n, 0);
    W.order.clear();
    W.order.reserve((size_t)n); W.svc_l.reserve((size_t)n); W.op_l.reserve((size_t)n); W.child_l.reserve((size_t)n);
    W.n_children.assign((size_t)n, 0);
    W.pos.assign((size_t)n, -1);
    W.names.clear(); W.svc_l.clear(); W.op_l.clear(); W.child_l.clear();
    auto local = [&](const std::string& name) -> int32_t {
        const int32_t id = N.local(name);
        bool met = false;   // few distinct names per trace
        for (int32_t k : W.names) if (k == id) { met = true; break; }
        if (!met) W.names.push_back(id);
        return id;
    };
    while (!stack.empty()) {
        const int v = stack.back();
        stack.pop_back();
        if (seen[(size_t)v]) return false;  // a span referenced twice / a cycle
        seen[(size_t)v] = 1;
        W.pos[(size_t)v] = (int)W.order.size();
        W.order.push_back(v);
        const SpanTmp& s = T.spans[(size_t)v];
        if (s.kind != 1 && s.kind != 2) return false;                 // AddSpanToProcess asserts on other kinds
        const std::string* svc = proc.find(s.pid);
        if (svc == nullptr) return false;
        const int ca = child_off[(size_t)v], cb = child_off[(size_t)v + 1];
        if (s.kind == 2 && cb - ca != 1) return false;                // GetChildProcess asserts one child
        if (v != root && s.refs.size() != 1) return false;            // GetParentProcess asserts one reference
        W.n_children[(size_t)v] = cb - ca;
        W.svc_l.push_back(local(*svc));
        W.op_l.push_back(local(s.op));
        int32_t cl = -1;
        if (cb > ca) {
            const std::string* cs = proc.find(T.spans[(size_t)child[(size_t)ca]].pid);
            static const std::string none;
            cl = local(cs != nullptr ? *cs : none);
        }
        W.child_l.push_back(cl);
        for (int k = cb; k-- > ca;) stack.push_back(child[(size_t)k]);
    }
    return true;
}

// The append to the corpus in three steps (append_begin / append_rows / append_lists): names are interned by one thread in
// trace order (a few per trace: string ids do not depend on the number of parser threads), the rows are written by the
// parser threads (every trace knows its first row), the per-service row lists by one thread streaming over the new rows.
struct Slot { int32_t trace_no = -1, base = 0; };

void append_begin(tw_corpus* c, const Walked& W, NameTable& N, Slot& slot, int64_t& n_rows) {
    slot.trace_no = (int32_t)c->trace_name.size();
    c->trace_name.push_back(kTraceBase + slot.trace_no);
    slot.base = (int32_t)n_rows;
    if (N.global.size() < N.names.size()) N.global.resize(N.names.size(), -1);
    for (int32_t k : W.names)
        if (N.global[(size_t)k] < 0) N.global[(size_t)k] = c->intern(N.names[(size_t)k]);
    n_rows += (int64_t)W.order.size();
}

void append_rows(tw_corpus* c, TraceTmp& T, const Walked& W, const Slot& slot, const NameTable& N) {
    const int32_t* g = N.global.data();
    c->trace_ids[(size_t)slot.trace_no] = std::move(T.trace_id);
    for (size_t k = 0; k < W.order.size(); k++) {
        const int v = W.order[k];
        SpanTmp& s = T.spans[(size_t)v];
        SpanRow& r = c->rows[(size_t)slot.base + k];
        r.trace = slot.trace_no;
        r.sid = kSidBase + slot.base + (int32_t)k;
        c->sids[(size_t)slot.base + k] = std::move(s.sid);
        r.service = g[W.svc_l[k]];
        r.op = g[W.op_l[k]];
        r.parent = W.parent[(size_t)v] >= 0 ? slot.base + W.pos[(size_t)W.parent[(size_t)v]] : -1;
        r.n_children = W.n_children[(size_t)v];
        r.first_child_service = W.child_l[k] >= 0 ? g[W.child_l[k]] : -1;
        r.start = s.start;
        r.dur = s.dur;
        r.kind = (uint8_t)s.kind;
    }
}

void append_lists(tw_corpus* c, size_t first_row) {
    int32_t last_in = -1, last_out = -1;
    std::vector<int32_t>*in_list = nullptr, *out_list = nullptr;
    for (size_t row = first_row; row < c->rows.size(); row++) {
        const SpanRow& r = c->rows[row];
        if (r.kind == 2) {
            if (r.service != last_out || out_list == nullptr) {
                auto it = c->out_rows.find(r.service);
                if (it == c->out_rows.end()) { c->service_order.push_back(r.service); it = c->out_rows.emplace(r.service, std::vector<int32_t>()).first; }
                out_list = &it->second; last_out = r.service;
            }
            out_list->push_back((int32_t)row);
        } else {
            if (r.service != last_in || in_list == nullptr) { in_list = &c->in_rows[r.service]; last_in = r.service; }
            in_list->push_back((int32_t)row);
        }
    }
}

// One service -> one unit (or the reason it is left out).  Reads the corpus only: services are built side by side.
struct UnitOut {
    bool emitted = false;
    int skip = 0;   // 1 several callers, 2 skip mode / too many endpoints, 3 too small, 4 cyclic call order
    int32_t in_ep = -1;
    std::vector<int64_t> in_start, in_end, out_start, out_end, ep_size;
    std::vector<int32_t> in_trace, in_row, out_row, ep_name, key_rank, truth;
    std::vector<uint8_t> dag;
};

void build_service(const tw_corpus* c, int32_t svc, const std::unordered_map<int32_t, int32_t>& root_key, std::vector<int32_t>& first, UnitOut& U) {
    auto in_it = c->in_rows.find(svc);
    if (in_it == c->in_rows.end()) return;
    // PartitionSpansByEndPoint: keys in order of first appearance (a handful), each partition sorted by (start, end), stable
    struct Part { int32_t key; std::vector<int32_t> rows; };
    std::vector<Part> in_part, out_part;
    auto part_of = [](std::vector<Part>& parts, int32_t key) -> std::vector<int32_t>& {
        for (Part& q : parts) if (q.key == key) return q.rows;
        parts.push_back(Part{key, {}});
        return parts.back().rows;
    };
    for (int32_t row : in_it->second) {
        const SpanRow& r = c->rows[(size_t)row];
        const int32_t key = r.parent < 0 ? root_key.at(r.op) : c->rows[(size_t)r.parent].service;
        part_of(in_part, key).push_back(row);
    }
    auto out_it = c->out_rows.find(svc);
    if (out_it != c->out_rows.end())
        for (int32_t row : out_it->second) part_of(out_part, c->rows[(size_t)row].first_child_service).push_back(row);
    if (in_part.size() != 1) { U.skip = 1; return; }  // executor.py:1126-1128
    struct Timed { int64_t start, end; int32_t row; };
    auto by_time = [&](std::vector<int32_t>& rows) {
        std::vector<Timed> t(rows.size());
        for (size_t i = 0; i < rows.size(); i++) { const SpanRow& r = c->rows[(size_t)rows[i]]; t[i] = Timed{r.start, r.start + r.dur, rows[i]}; }
        std::stable_sort(t.begin(), t.end(), [](const Timed& a, const Timed& b) { return a.start != b.start ? a.start < b.start : a.end < b.end; });
        for (size_t i = 0; i < rows.size(); i++) rows[i] = t[i].row;
        return t;
    };
    std::vector<int32_t>& ins = in_part[0].rows;
    const std::vector<Timed> in_t = by_time(ins);
    const int64_t n = (int64_t)ins.size();
    const int E = (int)out_part.size();
    bool equal = true;
    std::vector<std::vector<Timed>> out_t((size_t)E);
    for (int a = 0; a < E; a++) equal = equal && (int64_t)out_part[(size_t)a].rows.size() == n;
    if (!equal || E > TW_MAX_EP) { U.skip = 2; return; }   // skip mode / too many endpoints: not accelerated
    if (n < 2) { U.skip = 3; return; }
    for (int a = 0; a < E; a++) out_t[(size_t)a] = by_time(out_part[(size_t)a].rows);
    // GetGroundTruth: first outgoing span of the same trace per endpoint (key order)
    std::vector<std::vector<int32_t>> truth((size_t)E, std::vector<int32_t>((size_t)n, -1));
    for (int a = 0; a < E; a++) {
        const std::vector<int32_t>& part = out_part[(size_t)a].rows;
        for (int32_t j = (int32_t)part.size(); j-- > 0;) first[(size_t)c->rows[(size_t)part[(size_t)j]].trace] = j;
        for (int64_t i = 0; i < n; i++) truth[(size_t)a][(size_t)i] = first[(size_t)c->rows[(size_t)ins[(size_t)i]].trace];
        for (int32_t row : part) first[(size_t)c->rows[(size_t)row].trace] = -1;
    }
    // FindOrder: a -> b iff a ended no later than b started in every request
    std::vector<uint8_t> rel((size_t)(E * E), 1);
    for (int a = 0; a < E; a++) rel[(size_t)(a * E + a)] = 0;
    for (int64_t i = 0; i < n; i++)
        for (int a = 0; a < E; a++) {
            const int32_t xa = truth[(size_t)a][(size_t)i];
            if (xa < 0) continue;
            const int64_t a_end = out_t[(size_t)a][(size_t)xa].end;
            for (int b = 0; b < E; b++) {
                const int32_t xb = truth[(size_t)b][(size_t)i];
                if (a == b || xb < 0) continue;
                if (a_end > out_t[(size_t)b][(size_t)xb].start) rel[(size_t)(a * E + b)] = 0;
            }
        }
    // nx.topological_sort: generations of zero in-degree nodes, ties in insertion (= key) order
    std::vector<int> indeg((size_t)E, 0), topo;
    for (int a = 0; a < E; a++) for (int b = 0; b < E; b++) indeg[(size_t)b] += rel[(size_t)(a * E + b)];
    std::vector<int> ready;
    for (int a = 0; a < E; a++) if (indeg[(size_t)a] == 0) ready.push_back(a);
    for (size_t h = 0; h < ready.size(); h++) {
        const int a = ready[h];
        topo.push_back(a);
        for (int b = 0; b < E; b++) if (rel[(size_t)(a * E + b)] && --indeg[(size_t)b] == 0) ready.push_back(b);
    }
    if ((int)topo.size() != E) { U.skip = 4; return; }
    U.emitted = true;
    U.in_ep = in_part[0].key;
    U.in_start.resize((size_t)n); U.in_end.resize((size_t)n); U.in_trace.resize((size_t)n); U.in_row.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        U.in_start[(size_t)i] = in_t[(size_t)i].start; U.in_end[(size_t)i] = in_t[(size_t)i].end;
        U.in_trace[(size_t)i] = c->rows[(size_t)ins[(size_t)i]].trace; U.in_row[(size_t)i] = ins[(size_t)i];
    }
    for (int k = 0; k < E; k++) {
        const int a = topo[(size_t)k];
        for (const Timed& t : out_t[(size_t)a]) { U.out_start.push_back(t.start); U.out_end.push_back(t.end); U.out_row.push_back(t.row); }
        U.ep_size.push_back((int64_t)out_t[(size_t)a].size());
        U.ep_name.push_back(out_part[(size_t)a].key);
        U.key_rank.push_back(a);
        U.truth.insert(U.truth.end(), truth[(size_t)a].begin(), truth[(size_t)a].end());
    }
    for (int k = 0; k < E; k++) for (int l = 0; l < E; l++) U.dag.push_back(rel[(size_t)(topo[(size_t)k] * E + topo[(size_t)l])]);
}

// nt - 1 helper threads and the caller run f(worker number) side by side, one call of run() after the other
class Crew {
public:
    explicit Crew(int nt) : nt_(nt) {
        for (int t = 1; t < nt; t++) helpers_.emplace_back([this, t]() { loop(t); });
    }
    ~Crew() {
        { std::lock_guard<std::mutex> g(m_); quit_ = true; }
        go_.notify_all();
        for (auto& th : helpers_) th.join();
    }
    void run(const std::function<void(int)>& f) {
        { std::lock_guard<std::mutex> g(m_); job_ = &f; round_++; pending_ = nt_ - 1; }
        go_.notify_all();
        f(0);
        std::unique_lock<std::mutex> g(m_);
        idle_.wait(g, [&]() { return pending_ == 0; });
        job_ = nullptr;
    }
private:
    void loop(int t) {
        int seen = 0;
        while (true) {
            const std::function<void(int)>* f = nullptr;
            {
                std::unique_lock<std::mutex> g(m_);
                go_.wait(g, [&]() { return quit_ || round_ != seen; });
                if (quit_) return;
                seen = round_;
                f = job_;
            }
            (*f)(t);
            { std::lock_guard<std::mutex> g(m_); pending_--; }
            idle_.notify_one();
        }
    }
    int nt_, round_ = 0, pending_ = 0;
    bool quit_ = false;
    const std::function<void(int)>* job_ = nullptr;
    std::mutex m_;
    std::condition_variable go_, idle_;
    std::vector<std::thread> helpers_;
};

}  // namespace

extern "C" {

int tw_corpus_create(tw_corpus** out) {
    if (out == nullptr) return TW_ERR_ARG;
    *out = new tw_corpus();
    return TW_OK;
}
void tw_corpus_destroy(tw_corpus* c) { delete c; }
const char* tw_corpus_last_error(const tw_corpus* c) { return c ? c->err.c_str() : "null corpus"; }

int tw_corpus_set_callers(tw_corpus* c, const char* const* service, const char* const* caller, int32_t n) {
    if (c == nullptr || n < 0 || (n > 0 && (service == nullptr || caller == nullptr))) return TW_ERR_ARG;
    c->caller_of.clear();
    for (int i = 0; i < n; i++) c->caller_of[service[i]] = caller[i];
    return TW_OK;
}

int tw_corpus_add_files(tw_corpus* c, const char* const* paths, int32_t n_paths, const char* first_span, int64_t max_traces,
                        int32_t n_threads, int32_t fix) {
    if (c == nullptr || (n_paths > 0 && paths == nullptr) || n_paths < 0 || fix < 0 || fix > 3) return TW_ERR_ARG;
    std::vector<TraceTmp> parsed((size_t)n_paths);
    std::vector<Walked> walked((size_t)n_paths);
    std::vector<char> usable((size_t)n_paths, 0);
    std::vector<std::vector<std::pair<std::string, std::string>>> loops(fix == TW_FIX_RPC_TWINS ? (size_t)n_paths : 0);
    const std::string fs = first_span ? first_span : "";
    // <= 0: up to 32 parser threads, one per ~64 files
    const int want = n_threads <= 0 ? std::min(std::min((int)std::thread::hardware_concurrency(), 32), n_paths / 64 + 1) : n_threads;
    const int nt = std::max(1, std::min(want, std::max(n_paths, 1)));
    const bool timing = getenv("TW_INGEST_TIMING") != nullptr;   // debug aid: phase times on stderr
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
    // One crew of parser threads for the whole call.  A trace stays with the thread that parsed it: the later steps (rewrite,
    // rows, release) touch memory that thread allocated, so no two threads meet in one allocator arena.
    Crew crew(nt);
    std::vector<std::vector<int>> mine((size_t)nt);
    std::vector<NameTable> tables((size_t)nt);
    std::vector<int> worker_of((size_t)n_paths, 0);
    std::vector<double> key((size_t)n_paths);   // TimeOrder key per file: start of the root span, +inf without one
    std::atomic<int> next(0);
    crew.run([&](int t) {  // everything that touches one trace only: read, parse, span surgery, walk
        std::string buf;
        std::vector<int>& own = mine[(size_t)t];
        for (int i = next.fetch_add(1); i < n_paths; i = next.fetch_add(1)) {
            own.push_back(i);
            worker_of[(size_t)i] = t;
            TraceTmp& T = parsed[(size_t)i];
            key[(size_t)i] = std::numeric_limits<double>::infinity();
            if (!read_file(paths[i], buf)) { T.error = "cannot read file"; continue; }
            parse_trace(buf.data(), buf.size(), T);
            if (!T.ok) continue;
            if (T.has_root) key[(size_t)i] = T.root_start;
            bool ok = true;
            if (fix == TW_FIX_CLIENT_TWINS) ok = fix_client_twins(T, c->caller_of);
            else if (fix == TW_FIX_REROOT) ok = fix_reroot(T, fs);
            else if (fix == TW_FIX_RPC_TWINS) { self_calls(T, loops[(size_t)i]); continue; }   // the rewrite reads the corpus-wide self-call map: below
            usable[(size_t)i] = ok && walk_trace(T, fs, walked[(size_t)i], tables[(size_t)t]);
        }
    });
    if (timing) fprintf(stderr, "tw_corpus_add_files: %d files, %d threads: read+parse+walk %.3f s", n_paths, nt, since());
    // TimeOrder (executor.py:314-318): by the start of the root span, files without one last; ties keep the given order
    std::vector<int> idx((size_t)n_paths);
    for (int i = 0; i < n_paths; i++) idx[(size_t)i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return key[(size_t)a] < key[(size_t)b]; });
    int64_t accepted = (int64_t)c->trace_name.size();
    for (int i = 0; i < n_paths; i++) {  // parse failures are reported whether or not the trace limit is reached first
        c->files_total++;
        if (parsed[(size_t)i].ok) continue;
        c->files_rejected++;
        if (c->err.empty()) c->err = std::string(paths[i]) + ": " + parsed[(size_t)i].error;
    }
    std::vector<int64_t> seq_of((size_t)n_paths, -1);
    if (fix == TW_FIX_RPC_TWINS) {
        // the self-call map in time order (a few entries), then the rewrite + walk of every trace on the parser threads
        for (int i : idx) {
            if (!parsed[(size_t)i].ok) continue;
            seq_of[(size_t)i] = c->traces_seen++;
            enter_self_calls(c, loops[(size_t)i], seq_of[(size_t)i]);
        }
        crew.run([&](int t) {
            for (int i : mine[(size_t)t]) {
                TraceTmp& T = parsed[(size_t)i];
                if (T.ok) usable[(size_t)i] = fix_rpc_twins(T, c, seq_of[(size_t)i]) && walk_trace(T, fs, walked[(size_t)i], tables[(size_t)t]);
            }
        });
        if (timing) fprintf(stderr, ", rewrite+walk done at %.3f s", since());
    }
    int64_t last_seq = -1;
    bool stopped = false;
    std::vector<Slot> slots((size_t)n_paths);
    const size_t first_row = c->rows.size();
    int64_t n_rows = (int64_t)first_row;
    for (int i : idx) {
        TraceTmp& T = parsed[(size_t)i];
        if (!T.ok) continue;
        last_seq = seq_of[(size_t)i];
        if (usable[(size_t)i]) { append_begin(c, walked[(size_t)i], tables[(size_t)worker_of[(size_t)i]], slots[(size_t)i], n_rows); accepted++; } else c->traces_filtered++;
        if (max_traces > 0 && accepted >= max_traces) { stopped = true; break; }  // executor.py:873 stops after 1001 accepted traces
    }
    if (n_rows >= (int64_t)kSidBase || (int64_t)c->trace_name.size() >= (int64_t)(kSidBase - kTraceBase) || (int64_t)c->strings.size() >= (int64_t)kTraceBase) {
        c->err = "more spans, traces or names than the string ids of one corpus hold (2^30 / 2^29 / 2^29)";
        return TW_ERR_UNSUPPORTED;
    }
    c->rows.resize((size_t)n_rows);
    c->sids.resize((size_t)n_rows);
    c->trace_ids.resize(c->trace_name.size());
    if (timing) fprintf(stderr, ", order+names done at %.3f s", since());
    crew.run([&](int t) {   // rows of the accepted traces; then the parsed structures are released by the thread that built them
        for (int i : mine[(size_t)t]) {
            if (slots[(size_t)i].trace_no >= 0) append_rows(c, parsed[(size_t)i], walked[(size_t)i], slots[(size_t)i], tables[(size_t)t]);
            std::vector<SpanTmp>().swap(parsed[(size_t)i].spans);
            std::vector<std::pair<std::string, std::string>>().swap(parsed[(size_t)i].processes);
            Walked empty;
            std::swap(walked[(size_t)i], empty);
            if (!loops.empty()) std::vector<std::pair<std::string, std::string>>().swap(loops[(size_t)i]);
        }
    });
    append_lists(c, first_row);
    if (fix == TW_FIX_RPC_TWINS && stopped) {   // the reference never reached the later traces: their self-calls are not in its map
        for (auto it = c->loop_seq.begin(); it != c->loop_seq.end();) {
            if (it->second <= last_seq) { ++it; continue; }
            auto nm = c->loop_by_rpc.find(it->first);
            if (nm != c->loop_by_rpc.end()) { c->loop_origin.erase(nm->second); c->loop_by_rpc.erase(nm); }
            it = c->loop_seq.erase(it);
        }
        c->traces_seen = last_seq + 1;
    }
    if (timing) fprintf(stderr, ", rows+lists done at %.3f s\n", since());
    return TW_OK;
}

int tw_corpus_counts(const tw_corpus* c, int64_t* out6) {
    if (c == nullptr || out6 == nullptr) return TW_ERR_ARG;
    out6[0] = (int64_t)c->rows.size(); out6[1] = (int64_t)c->trace_name.size(); out6[2] = c->files_total;
    out6[3] = c->files_rejected; out6[4] = c->traces_filtered; out6[5] = (int64_t)(c->strings.size() + c->sids.size() + c->trace_ids.size());
    return TW_OK;
}

const char* tw_corpus_string(const tw_corpus* c, int32_t id) {
    if (c == nullptr || id < 0) return nullptr;
    if (id >= kSidBase) return (size_t)(id - kSidBase) < c->sids.size() ? c->sids[(size_t)(id - kSidBase)].c_str() : nullptr;
    if (id >= kTraceBase) return (size_t)(id - kTraceBase) < c->trace_ids.size() ? c->trace_ids[(size_t)(id - kTraceBase)].c_str() : nullptr;
    return (size_t)id < c->strings.size() ? c->strings[(size_t)id].c_str() : nullptr;
}

const char* tw_corpus_loop_origin(const tw_corpus* c, const char* service) {
    if (c == nullptr || service == nullptr) return nullptr;
    auto it = c->loop_origin.find(service);
    return it == c->loop_origin.end() ? nullptr : it->second.c_str();
}

int tw_corpus_trace_names(const tw_corpus* c, int32_t* out) {
    if (c == nullptr || out == nullptr) return TW_ERR_ARG;
    for (size_t i = 0; i < c->trace_name.size(); i++) out[i] = c->trace_name[i];
    return TW_OK;
}

int tw_corpus_span_table(const tw_corpus* c, tw_span_table* t) {
    if (c == nullptr || t == nullptr) return TW_ERR_ARG;
    const int64_t n = (int64_t)c->rows.size();
    for (int64_t i = 0; i < n; i++) {
        const SpanRow& r = c->rows[(size_t)i];
        if (t->trace) t->trace[i] = r.trace;
        if (t->span_id) t->span_id[i] = r.sid;
        if (t->service) t->service[i] = r.service;
        if (t->op_name) t->op_name[i] = r.op;
        if (t->parent) t->parent[i] = r.parent;
        if (t->start) t->start[i] = r.start;
        if (t->duration) t->duration[i] = r.dur;
        if (t->kind) t->kind[i] = r.kind;
    }
    return TW_OK;
}

int tw_corpus_build_units(tw_corpus* c, tw_unit_set* out) {
    if (c == nullptr || out == nullptr) return TW_ERR_ARG;
    c->u_in_off.assign(1, 0); c->u_ep_off.assign(1, 0);
    c->u_in_start.clear(); c->u_in_end.clear(); c->u_out_start.clear(); c->u_out_end.clear();
    c->u_E.clear(); c->u_key_rank.clear(); c->u_truth.clear(); c->u_in_trace.clear(); c->u_in_row.clear(); c->u_out_row.clear();
    c->u_service.clear(); c->u_ep_name.clear(); c->u_in_ep.clear(); c->u_dag.clear(); c->u_order.clear();
    c->skipped_multi_in = c->skipped_skip_mode = c->skipped_small = c->skipped_cyclic = 0;
    // "client_<operation>" endpoint names of the entry services, interned ahead (in service / row order) so that the
    // services can then be built side by side without touching the string table
    std::unordered_map<int32_t, int32_t> root_key;   // operation id -> id of "client_<operation>"
    for (int32_t svc : c->service_order) {
        auto in_it = c->in_rows.find(svc);
        if (in_it == c->in_rows.end()) continue;
        for (int32_t row : in_it->second) {
            const SpanRow& r = c->rows[(size_t)row];
            if (r.parent < 0 && root_key.find(r.op) == root_key.end()) root_key.emplace(r.op, c->intern("client_" + c->strings[(size_t)r.op]));
        }
    }
    const int n_svc = (int)c->service_order.size();
    std::vector<UnitOut> built((size_t)n_svc);
    const int64_t n_traces = (int64_t)c->trace_name.size();
    {
        const int nt = std::max(1, std::min(std::min((int)std::thread::hardware_concurrency(), 16), c->rows.size() < 20000 ? 1 : n_svc));
        Crew crew(nt);
        std::atomic<int> next(0);
        crew.run([&](int) {
            std::vector<int32_t> first((size_t)n_traces, -1);   // per trace: the first outgoing span of the endpoint at hand
            for (int k = next.fetch_add(1); k < n_svc; k = next.fetch_add(1)) build_service(c, c->service_order[(size_t)k], root_key, first, built[(size_t)k]);
        });
    }
    for (int k = 0; k < n_svc; k++) {
        const UnitOut& U = built[(size_t)k];
        if (U.skip == 1) c->skipped_multi_in++;
        else if (U.skip == 2) c->skipped_skip_mode++;
        else if (U.skip == 3) c->skipped_small++;
        else if (U.skip == 4) c->skipped_cyclic++;
        if (!U.emitted) continue;
        const int E = (int)U.ep_name.size();
        const int64_t n = (int64_t)U.in_start.size();
        c->u_service.push_back(c->service_order[(size_t)k]);
        c->u_order.push_back(k);
        c->u_in_ep.push_back(U.in_ep);
        c->u_E.push_back(E);
        c->u_in_start.insert(c->u_in_start.end(), U.in_start.begin(), U.in_start.end());
        c->u_in_end.insert(c->u_in_end.end(), U.in_end.begin(), U.in_end.end());
        c->u_in_trace.insert(c->u_in_trace.end(), U.in_trace.begin(), U.in_trace.end());
        c->u_in_row.insert(c->u_in_row.end(), U.in_row.begin(), U.in_row.end());
        c->u_in_off.push_back(c->u_in_off.back() + n);
        c->u_out_start.insert(c->u_out_start.end(), U.out_start.begin(), U.out_start.end());
        c->u_out_end.insert(c->u_out_end.end(), U.out_end.begin(), U.out_end.end());
        c->u_out_row.insert(c->u_out_row.end(), U.out_row.begin(), U.out_row.end());
        for (int e = 0; e < E; e++) c->u_ep_off.push_back(c->u_ep_off.back() + U.ep_size[(size_t)e]);
        c->u_ep_name.insert(c->u_ep_name.end(), U.ep_name.begin(), U.ep_name.end());
        c->u_key_rank.insert(c->u_key_rank.end(), U.key_rank.begin(), U.key_rank.end());
        c->u_truth.insert(c->u_truth.end(), U.truth.begin(), U.truth.end());
        c->u_dag.insert(c->u_dag.end(), U.dag.begin(), U.dag.end());
    }
    out->n_units = (int32_t)c->u_E.size();
    out->unit_in_off = c->u_in_off.data(); out->unit_E = c->u_E.data(); out->ep_off = c->u_ep_off.data();
    out->dag = c->u_dag.data(); out->key_rank = c->u_key_rank.data();
    out->in_start = c->u_in_start.data(); out->in_end = c->u_in_end.data(); out->out_start = c->u_out_start.data(); out->out_end = c->u_out_end.data();
    out->true_child = c->u_truth.data(); out->in_trace = c->u_in_trace.data(); out->in_row = c->u_in_row.data(); out->out_row = c->u_out_row.data();
    out->unit_service = c->u_service.data(); out->ep_name = c->u_ep_name.data(); out->in_ep_name = c->u_in_ep.data(); out->unit_order = c->u_order.data();
    out->n_traces = (int64_t)c->trace_name.size();
    out->skipped[0] = c->skipped_multi_in; out->skipped[1] = c->skipped_skip_mode; out->skipped[2] = c->skipped_small; out->skipped[3] = c->skipped_cyclic;
    return TW_OK;
}

}  // extern "C"
