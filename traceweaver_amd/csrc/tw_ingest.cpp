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
#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "traceweaver_amd.h"

namespace {

// ------------------------------------------------------------------------------------------------
// JSON scanner (RFC 8259): no document tree -- the loader walks the text once, copies out the handful of fields it
// needs and skips everything else (tags it does not read, logs, warnings) without allocating.
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
    // a string value; `out` == nullptr just skips it
    bool str(std::string* out) {
        ws();
        if (p >= end || *p != '"') return fail("expected string");
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
        while (q < end && *q >= '0' && *q <= '9') q++;
        if (q < end && (*q == '.' || *q == 'e' || *q == 'E')) {
            integral = false;
            while (q < end && ((*q >= '0' && *q <= '9') || *q == '.' || *q == 'e' || *q == 'E' || *q == '+' || *q == '-')) q++;
        }
        const std::string tok(p, q);
        out = (integral && tok.size() <= 19) ? strtoll(tok.c_str(), nullptr, 10) : (int64_t)strtod(tok.c_str(), nullptr);
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
        std::string key;
        while (true) {
            if (!str(&key) || !expect(':', "expected ':'")) return false;
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
    std::string why, key_s, val_s;
    std::vector<std::string> span_tids;  // the traceID key of the trace may come after its spans in the text
    const bool ok = S.object([&](const std::string& k0) {
        if (k0 != "data") return S.skip();
        return S.array([&]() {
            if (++n_traces > 1) return S.skip();
            return S.object([&](const std::string& k1) {
                if (k1 == "traceID") { have_tid = S.is_string(); return have_tid ? S.str(&T.trace_id) : S.skip(); }
                if (k1 == "processes") {
                    S.ws();
                    if (S.p >= S.end || *S.p != '{') return S.skip();
                    have_procs = true;
                    return S.object([&](const std::string& pid) {
                        std::string name;
                        bool named = false;
                        S.ws();
                        if (S.p >= S.end || *S.p != '{') { if (why.empty()) why = "process without serviceName"; bad_span = true; return S.skip(); }
                        if (!S.object([&](const std::string& k3) {
                                if (k3 == "serviceName" && S.is_string() && !named) { named = true; return S.str(&name); }
                                return S.skip();
                            })) return false;
                        if (!named) { if (why.empty()) why = "process without serviceName"; bad_span = true; }
                        T.processes.emplace_back(pid, name);
                        return true;
                    });
                }
                if (k1 != "spans") return S.skip();
                S.ws();
                if (S.p >= S.end || *S.p != '[') return S.skip();
                have_spans = true;
                return S.array([&]() {
                    SpanTmp sp;
                    std::string stid, req;
                    bool has_sid = false, has_pid = false, has_tid = false, has_start = false, has_dur = false, has_op = false, has_req = false;
                    S.ws();
                    if (S.p >= S.end || *S.p != '{') { bad_span = true; if (why.empty()) why = "span without spanID / processID"; return S.skip(); }
                    if (!S.object([&](const std::string& k2) {
                            if (k2 == "spanID" && S.is_string() && !has_sid) { has_sid = true; return S.str(&sp.sid); }
                            if (k2 == "processID" && S.is_string() && !has_pid) { has_pid = true; return S.str(&sp.pid); }
                            if (k2 == "traceID" && S.is_string() && !has_tid) { has_tid = true; return S.str(&stid); }
                            if (k2 == "operationName" && !has_op) { has_op = true; return S.is_string() ? S.str(&sp.op) : S.skip(); }
                            if (k2 == "caller" && S.is_string() && !sp.has_caller) { sp.has_caller = true; return S.str(&sp.caller); }
                            if (k2 == "callee" && S.is_string() && !sp.has_callee) { sp.has_callee = true; return S.str(&sp.callee); }
                            if (k2 == "requestType" && !has_req) { has_req = true; if (S.is_string()) return S.str(&req); req.clear(); return S.skip(); }
                            if ((k2 == "startTime" && !has_start) || (k2 == "duration" && !has_dur)) {
                                S.ws();
                                const bool num = S.p < S.end && (*S.p == '-' || (*S.p >= '0' && *S.p <= '9'));
                                if (!num) return S.skip();
                                if (k2 == "startTime") { has_start = true; return S.i64(sp.start); }
                                has_dur = true;
                                return S.i64(sp.dur);
                            }
                            if (k2 == "tags") {
                                S.ws();
                                if (S.p >= S.end || *S.p != '[') return S.skip();
                                return S.array([&]() {
                                    S.ws();
                                    if (S.p >= S.end || *S.p != '{') return S.skip();
                                    bool is_kind = false, have_val = false;
                                    if (!S.object([&](const std::string& k3) {
                                            if (k3 == "key" && S.is_string()) { if (!S.str(&key_s)) return false; is_kind = key_s == "span.kind"; return true; }
                                            if (k3 == "value" && S.is_string()) { have_val = true; return S.str(&val_s); }
                                            return S.skip();
                                        })) return false;
                                    if (is_kind && have_val) sp.kind = val_s == "server" ? 1 : (val_s == "client" ? 2 : 3);
                                    return true;
                                });
                            }
                            if (k2 == "references") {
                                S.ws();
                                if (S.p >= S.end || *S.p != '[') return S.skip();
                                return S.array([&]() {
                                    std::string rt, rs;
                                    bool hrt = false, hrs = false;
                                    S.ws();
                                    if (S.p >= S.end || *S.p != '{') { bad_span = true; if (why.empty()) why = "malformed reference"; return S.skip(); }
                                    if (!S.object([&](const std::string& k3) {
                                            if (k3 == "traceID" && S.is_string()) { hrt = true; return S.str(&rt); }
                                            if (k3 == "spanID" && S.is_string()) { hrs = true; return S.str(&rs); }
                                            return S.skip();
                                        })) return false;
                                    if (!hrt || !hrs) { bad_span = true; if (why.empty()) why = "malformed reference"; }
                                    sp.refs.emplace_back(std::move(rt), std::move(rs));
                                    return true;
                                });
                            }
                            return S.skip();
                        })) return false;
                    if (!has_sid || !has_pid) { bad_span = true; if (why.empty()) why = "span without spanID / processID"; }
                    else if (!has_tid) { bad_span = true; if (why.empty()) why = "different trace ids for spans in the same trace"; }  // executor.py:367-372
                    else if (!has_start || !has_dur) { bad_span = true; if (why.empty()) why = "span without startTime / duration"; }
                    if (has_req) sp.op = req;  // executor.py:358-361: requestType wins over operationName
                    if (first_span) { T.request_type = has_req; first_span = false; }   // executor.py:774
                    if (sp.refs.empty() && !T.has_root) { T.has_root = true; T.root_start = (double)sp.start; }  // first span without references
                    T.spans.push_back(std::move(sp));
                    span_tids.push_back(std::move(stid));
                    return true;
                });
            });
        });
    });
    if (!ok) { T.error = "JSON: " + S.err; return; }
    if (n_traces != 1) { T.error = "expected exactly one trace under \"data\""; return; }
    if (!have_tid || !have_spans || T.spans.empty()) { T.error = "trace without traceID / spans"; return; }
    if (bad_span) { T.error = why; return; }
    for (const std::string& t : span_tids)
        if (t != T.trace_id) { T.error = "different trace ids for spans in the same trace"; return; }  // executor.py:367-372
    if (T.request_type) {  // ParseProcessesJson2: the process id is the service name
        T.processes.clear();
        for (const SpanTmp& sp : T.spans) T.processes.emplace_back(sp.pid, sp.pid);
    } else if (!have_procs) { T.error = "trace without a process map"; return; }
    T.ok = true;
}

bool read_file(const char* path, std::string& out) {
    FILE* f = fopen(path, "rb");
    if (f == nullptr) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const size_t got = n > 0 ? fread(&out[0], 1, (size_t)n, f) : 0;
    fclose(f);
    return got == out.size();
}

struct SpanRow {
    int32_t trace, sid, service, op, parent, n_children, first_child_service;
    int64_t start, dur;
    uint8_t kind;
};

}  // namespace

struct tw_corpus {
    std::string err;
    std::vector<std::string> strings;
    std::unordered_map<std::string, int32_t> string_ids;
    std::vector<SpanRow> rows;                 // spans reached from the root of an accepted trace, pre-order
    std::vector<int32_t> trace_name;           // string id per accepted trace
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

    int32_t intern(const std::string& s) {
        auto it = string_ids.find(s);
        if (it != string_ids.end()) return it->second;
        const int32_t id = (int32_t)strings.size();
        strings.push_back(s);
        string_ids.emplace(s, id);
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
struct Walked {
    std::vector<int> order, parent, n_children;      // pre-order; per span (local index): parent, number of children
    std::vector<std::string> service, child_service; // per span: its service, the service of its first child ("" = none)
};

bool walk_trace(const TraceTmp& T, const std::string& first_span, Walked& W) {
    const int n = (int)T.spans.size();
    std::unordered_map<std::string, int> by_sid;
    for (int i = 0; i < n; i++) by_sid[T.spans[(size_t)i].sid] = i;  // later duplicates win, like the dict of the reference
    std::unordered_map<std::string, std::string> proc;
    for (const auto& kv : T.processes) proc[kv.first] = kv.second;
    int root = -1;
    std::vector<std::vector<int>> children((size_t)n);
    W.parent.assign((size_t)n, -1);
    for (int i = 0; i < n; i++) {
        const SpanTmp& s = T.spans[(size_t)i];
        if (by_sid[s.sid] != i) continue;  // shadowed duplicate
        if (s.refs.empty()) root = i;      // the last span without references (executor.py:823-825)
        for (const auto& r : s.refs) {
            auto it = by_sid.find(r.second);
            if (r.first != T.trace_id || it == by_sid.end()) return false;  // spans[par_id] would raise
            children[(size_t)it->second].push_back(i);
        }
        if (!s.refs.empty()) W.parent[(size_t)i] = by_sid[s.refs[0].second];
    }
    if (root < 0) return false;
    if (!first_span.empty() && T.spans[(size_t)root].op != first_span) return false;  // executor.py:841
    for (auto& ch : children)
        std::stable_sort(ch.begin(), ch.end(), [&](int a, int b) { return T.spans[(size_t)a].start < T.spans[(size_t)b].start; });
    std::vector<int> stack{root};
    std::vector<char> seen((size_t)n, 0);
    W.order.clear();
    W.n_children.assign((size_t)n, 0);
    W.service.assign((size_t)n, std::string());
    W.child_service.assign((size_t)n, std::string());
    while (!stack.empty()) {
        const int v = stack.back();
        stack.pop_back();
        if (seen[(size_t)v]) return false;  // a span referenced twice / a cycle
        seen[(size_t)v] = 1;
        W.order.push_back(v);
        const SpanTmp& s = T.spans[(size_t)v];
        if (s.kind != 1 && s.kind != 2) return false;                 // AddSpanToProcess asserts on other kinds
        auto pit = proc.find(s.pid);
        if (pit == proc.end()) return false;
        if (s.kind == 2 && children[(size_t)v].size() != 1) return false;   // GetChildProcess asserts one child
        if (v != root && s.refs.size() != 1) return false;            // GetParentProcess asserts one reference
        W.service[(size_t)v] = pit->second;
        W.n_children[(size_t)v] = (int)children[(size_t)v].size();
        if (!children[(size_t)v].empty()) {
            auto cit = proc.find(T.spans[(size_t)children[(size_t)v][0]].pid);
            if (cit != proc.end()) W.child_service[(size_t)v] = cit->second;
        }
        for (size_t k = children[(size_t)v].size(); k-- > 0;) stack.push_back(children[(size_t)v][k]);
    }
    return true;
}

void append_trace(tw_corpus* c, const TraceTmp& T, const Walked& W) {
    const int32_t trace_no = (int32_t)c->trace_name.size();
    c->trace_name.push_back(c->intern(T.trace_id));
    std::vector<int32_t> row_of(T.spans.size(), -1);
    const int32_t base = (int32_t)c->rows.size();
    for (size_t k = 0; k < W.order.size(); k++) row_of[(size_t)W.order[k]] = base + (int32_t)k;
    for (int v : W.order) {
        const SpanTmp& s = T.spans[(size_t)v];
        SpanRow r;
        r.trace = trace_no;
        r.sid = (int32_t)c->strings.size();   // span ids are unique per trace: stored, not looked up
        c->strings.push_back(s.sid);
        r.service = c->intern(W.service[(size_t)v]);
        r.op = c->intern(s.op);
        r.parent = W.parent[(size_t)v] >= 0 ? row_of[(size_t)W.parent[(size_t)v]] : -1;
        r.n_children = W.n_children[(size_t)v];
        r.first_child_service = r.n_children > 0 ? c->intern(W.child_service[(size_t)v]) : -1;
        r.start = s.start;
        r.dur = s.dur;
        r.kind = (uint8_t)s.kind;
        const int32_t row = (int32_t)c->rows.size();
        c->rows.push_back(r);
        if (s.kind == 2) {
            if (c->out_rows.find(r.service) == c->out_rows.end()) c->service_order.push_back(r.service);
            c->out_rows[r.service].push_back(row);
        } else c->in_rows[r.service].push_back(row);
    }
}

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
    std::atomic<int> next(0);
    auto work = [&]() {  // everything that touches one trace only: read, parse, span surgery, walk
        std::string buf;
        for (int i = next.fetch_add(1); i < n_paths; i = next.fetch_add(1)) {
            TraceTmp& T = parsed[(size_t)i];
            if (!read_file(paths[i], buf)) { T.error = "cannot read file"; continue; }
            parse_trace(buf.data(), buf.size(), T);
            if (!T.ok) continue;
            bool ok = true;
            if (fix == TW_FIX_CLIENT_TWINS) ok = fix_client_twins(T, c->caller_of);
            else if (fix == TW_FIX_REROOT) ok = fix_reroot(T, fs);
            else if (fix == TW_FIX_RPC_TWINS) { self_calls(T, loops[(size_t)i]); continue; }   // the rewrite reads the corpus-wide self-call map: below
            usable[(size_t)i] = ok && walk_trace(T, fs, walked[(size_t)i]);
        }
    };
    // <= 0: up to 16 parser threads, one per ~64 files (measured on the 256-thread host: 1 thread 0.4 M spans/s, 8 threads 1.0 M, 256 threads 0.13 M)
    const int want = n_threads <= 0 ? std::min(std::min((int)std::thread::hardware_concurrency(), 16), n_paths / 64 + 1) : n_threads;
    const int nt = std::max(1, std::min(want, std::max(n_paths, 1)));
    const bool timing = getenv("TW_INGEST_TIMING") != nullptr;   // debug aid: phase times on stderr
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    if (timing) fprintf(stderr, "tw_corpus_add_files: %d files, %d threads: read+parse+walk %.3f s", n_paths, nt, since());
    // TimeOrder (executor.py:314-318): by the start of the root span, files without one last; ties keep the given order
    std::vector<int> idx((size_t)n_paths);
    for (int i = 0; i < n_paths; i++) idx[(size_t)i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
        const TraceTmp &A = parsed[(size_t)a], &B = parsed[(size_t)b];
        const bool ra = A.ok && A.has_root, rb = B.ok && B.has_root;
        if (ra != rb) return ra;
        return ra && A.root_start < B.root_start;
    });
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
        std::atomic<int> next2(0);
        auto rewrite = [&]() {
            for (int i = next2.fetch_add(1); i < n_paths; i = next2.fetch_add(1)) {
                TraceTmp& T = parsed[(size_t)i];
                if (T.ok) usable[(size_t)i] = fix_rpc_twins(T, c, seq_of[(size_t)i]) && walk_trace(T, fs, walked[(size_t)i]);
            }
        };
        std::vector<std::thread> pool2;
        for (int t = 1; t < nt; t++) pool2.emplace_back(rewrite);
        rewrite();
        for (auto& th : pool2) th.join();
        if (timing) fprintf(stderr, ", rewrite+walk done at %.3f s", since());
    }
    int64_t last_seq = -1;
    bool stopped = false;
    for (int i : idx) {
        TraceTmp& T = parsed[(size_t)i];
        if (!T.ok) continue;
        last_seq = seq_of[(size_t)i];
        if (usable[(size_t)i]) { append_trace(c, T, walked[(size_t)i]); accepted++; } else c->traces_filtered++;
        if (max_traces > 0 && accepted >= max_traces) { stopped = true; break; }  // executor.py:873 stops after 1001 accepted traces
    }
    if (fix == TW_FIX_RPC_TWINS && stopped) {   // the reference never reached the later traces: their self-calls are not in its map
        for (auto it = c->loop_seq.begin(); it != c->loop_seq.end();) {
            if (it->second <= last_seq) { ++it; continue; }
            auto nm = c->loop_by_rpc.find(it->first);
            if (nm != c->loop_by_rpc.end()) { c->loop_origin.erase(nm->second); c->loop_by_rpc.erase(nm); }
            it = c->loop_seq.erase(it);
        }
        c->traces_seen = last_seq + 1;
    }
    if (timing) fprintf(stderr, ", order+append done at %.3f s\n", since());
    return TW_OK;
}

int tw_corpus_counts(const tw_corpus* c, int64_t* out6) {
    if (c == nullptr || out6 == nullptr) return TW_ERR_ARG;
    out6[0] = (int64_t)c->rows.size(); out6[1] = (int64_t)c->trace_name.size(); out6[2] = c->files_total;
    out6[3] = c->files_rejected; out6[4] = c->traces_filtered; out6[5] = (int64_t)c->strings.size();
    return TW_OK;
}

const char* tw_corpus_string(const tw_corpus* c, int32_t id) {
    if (c == nullptr || id < 0 || (size_t)id >= c->strings.size()) return nullptr;
    return c->strings[(size_t)id].c_str();
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
    const auto by_time = [&](int32_t a, int32_t b) {
        const SpanRow &A = c->rows[(size_t)a], &B = c->rows[(size_t)b];
        return A.start != B.start ? A.start < B.start : A.start + A.dur < B.start + B.dur;
    };
    int32_t order = -1;
    for (int32_t svc : c->service_order) {
        order++;
        auto in_it = c->in_rows.find(svc);
        if (in_it == c->in_rows.end()) continue;
        // PartitionSpansByEndPoint: keys in order of first appearance, each partition sorted by (start, end), stable
        std::vector<int32_t> in_keys, out_keys;
        std::unordered_map<int32_t, std::vector<int32_t>> in_part, out_part;
        for (int32_t row : in_it->second) {
            const SpanRow& r = c->rows[(size_t)row];
            const int32_t key = r.parent < 0 ? c->intern("client_" + c->strings[(size_t)r.op]) : c->rows[(size_t)r.parent].service;
            if (in_part.find(key) == in_part.end()) in_keys.push_back(key);
            in_part[key].push_back(row);
        }
        for (int32_t row : c->out_rows[svc]) {
            const int32_t key = c->rows[(size_t)row].first_child_service;
            if (out_part.find(key) == out_part.end()) out_keys.push_back(key);
            out_part[key].push_back(row);
        }
        if (in_keys.size() != 1) { c->skipped_multi_in++; continue; }  // executor.py:1126-1128
        std::vector<int32_t>& ins = in_part[in_keys[0]];
        std::stable_sort(ins.begin(), ins.end(), by_time);
        const int64_t n = (int64_t)ins.size();
        const int E = (int)out_keys.size();
        bool equal = true;
        for (int32_t k : out_keys) {
            std::stable_sort(out_part[k].begin(), out_part[k].end(), by_time);
            equal = equal && (int64_t)out_part[k].size() == n;
        }
        if (!equal || E > TW_MAX_EP) { c->skipped_skip_mode++; continue; }   // skip mode / too many endpoints: not accelerated
        if (n < 2) { c->skipped_small++; continue; }
        // GetGroundTruth: first outgoing span of the same trace per endpoint (key order)
        std::vector<std::vector<int32_t>> truth((size_t)E, std::vector<int32_t>((size_t)n, -1));
        for (int a = 0; a < E; a++) {
            std::unordered_map<int32_t, int32_t> first;
            const std::vector<int32_t>& part = out_part[out_keys[(size_t)a]];
            for (int32_t j = 0; j < (int32_t)part.size(); j++) first.emplace(c->rows[(size_t)part[(size_t)j]].trace, j);
            for (int64_t i = 0; i < n; i++) {
                auto it = first.find(c->rows[(size_t)ins[(size_t)i]].trace);
                if (it != first.end()) truth[(size_t)a][(size_t)i] = it->second;
            }
        }
        // FindOrder: a -> b iff a ended no later than b started in every request
        std::vector<uint8_t> rel((size_t)(E * E), 1);
        for (int a = 0; a < E; a++) rel[(size_t)(a * E + a)] = 0;
        for (int64_t i = 0; i < n; i++)
            for (int a = 0; a < E; a++) {
                const int32_t xa = truth[(size_t)a][(size_t)i];
                if (xa < 0) continue;
                const SpanRow& A = c->rows[(size_t)out_part[out_keys[(size_t)a]][(size_t)xa]];
                for (int b = 0; b < E; b++) {
                    const int32_t xb = truth[(size_t)b][(size_t)i];
                    if (a == b || xb < 0) continue;
                    if (A.start + A.dur > c->rows[(size_t)out_part[out_keys[(size_t)b]][(size_t)xb]].start) rel[(size_t)(a * E + b)] = 0;
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
        if ((int)topo.size() != E) { c->skipped_cyclic++; continue; }
        // emit
        c->u_service.push_back(svc);
        c->u_order.push_back(order);
        c->u_in_ep.push_back(in_keys[0]);
        c->u_E.push_back(E);
        for (int64_t i = 0; i < n; i++) {
            const SpanRow& r = c->rows[(size_t)ins[(size_t)i]];
            c->u_in_start.push_back(r.start); c->u_in_end.push_back(r.start + r.dur);
            c->u_in_trace.push_back(r.trace); c->u_in_row.push_back(ins[(size_t)i]);
        }
        c->u_in_off.push_back(c->u_in_off.back() + n);
        for (int k = 0; k < E; k++) {
            const int a = topo[(size_t)k];
            const std::vector<int32_t>& part = out_part[out_keys[(size_t)a]];
            for (int32_t row : part) {
                const SpanRow& r = c->rows[(size_t)row];
                c->u_out_start.push_back(r.start); c->u_out_end.push_back(r.start + r.dur); c->u_out_row.push_back(row);
            }
            c->u_ep_off.push_back(c->u_ep_off.back() + (int64_t)part.size());
            c->u_ep_name.push_back(out_keys[(size_t)a]);
            c->u_key_rank.push_back(a);
            for (int64_t i = 0; i < n; i++) c->u_truth.push_back(truth[(size_t)a][(size_t)i]);
        }
        for (int k = 0; k < E; k++) for (int l = 0; l < E; l++) c->u_dag.push_back(rel[(size_t)(topo[(size_t)k] * E + topo[(size_t)l])]);
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
