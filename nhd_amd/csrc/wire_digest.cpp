// wire_digest.cpp - pod request digest straight from the wire format.
//
// The reference turns the pod's libconfig text into a CfgTopology object graph
// (nhd/TriadCfgParser.py:337-380 CfgToTopology, 134-309 ParseModGroups, 106-132 ParseMiscCores, 92-104
// ParseHugePages) and FindNode then re-derives a handful of integers from it (nhd/CfgTopology.py:199-232).
// This file goes from the text to the fixed 128-byte nhdfit_req in one pass: an own reader for the libconfig
// grammar (the reference uses the third-party `libconf` package) and a walk of the Triad topology section that
// follows the reference's parser statement by statement - including what it does NOT check (chained `!=` length
// tests, cores added before a later failure, map types other than NUMA / PCI left INVALID).
//
// Host-only C++ (no HIP): part of libnhdfit.so, callable without a GPU.
//   0                 request written
//   NHDFIT_WIRE_NONE  the reference's CfgToTopology returns None for this text (logged error, pod not scheduled)
//   NHDFIT_WIRE_RAISE the reference would raise (malformed text, wrong value types) - the scheduler thread would die
//   NHDFIT_WIRE_LIMIT more than NHDFIT_MAX_GROUPS groups / 255 cores per group (same limit as Packer.digest)
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/nhdfit.h"

namespace {

// ASCII classification without the locale-aware <cctype> calls (they were 13 % of the profile)
inline bool is_digit(char c) { return c >= '0' && c <= '9'; }
inline bool is_alpha(char c) { return (unsigned char)((c | 32) - 'a') < 26; }
inline bool is_alnum(char c) { return is_digit(c) || is_alpha(c); }
inline bool is_space(char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
inline bool is_xdigit(char c) { return is_digit(c) || (unsigned char)((c | 32) - 'a') < 6; }
inline char to_lower(char c) { return (c >= 'A' && c <= 'Z') ? (char)(c | 32) : c; }

struct Reject { std::string why; };      // reference returns None
struct Raise { std::string why; };       // reference raises
struct AttrMissing { std::string why; }; // AttributeError inside a path look-up (caught or not depends on the call site)
struct Limit { std::string why; };

struct Value;
using ValuePtr = Value*;                                              // owned by the Arena of the current digest
struct Value {
    enum Kind { Int, Float, Bool, Str, Array, List, Group } kind = Int;
    long long i = 0;
    double f = 0;
    bool b = false;
    std::string s;
    std::vector<ValuePtr> items;                                  // Array [..] / List (..)
    struct Name { const char* p; size_t n; std::string str() const { return std::string(p, n); } };   // points into the input text
    std::vector<std::pair<Name, ValuePtr>> fields;                // Group {..}, in file order

    const Value* find(const std::string& key) const { return find(key.data(), key.size()); }
    const Value* find(const char* key) const { return find(key, std::strlen(key)); }
    const Value* find(const char* key, size_t len) const {
        for (const auto& kv : fields)
            if (kv.first.n == len && std::memcmp(kv.first.p, key, len) == 0) return kv.second;
        return nullptr;
    }
    bool has(const std::string& key) const { return find(key) != nullptr; }
    bool is_seq() const { return kind == Array || kind == List; }
};

// Values live in chunks that are recycled from one digest to the next (one allocation per 256 values instead of one
// per value: the reader was allocation-bound)
class Arena {
public:
    Value* make() {
        if (used_ == chunks_.size() * kChunk) chunks_.emplace_back(new Value[kChunk]);
        Value* v = &chunks_[used_ / kChunk][used_ % kChunk];
        ++used_;
        v->kind = Value::Int; v->i = 0; v->f = 0; v->b = false;
        v->s.clear(); v->items.clear(); v->fields.clear();
        return v;
    }
    void reset() { used_ = 0; }
private:
    static constexpr size_t kChunk = 256;
    std::vector<std::unique_ptr<Value[]>> chunks_;
    size_t used_ = 0;
};
thread_local Arena g_arena;

// ---- libconfig reader (grammar of libconfig 1.7 as the `libconf` package accepts it) -------------------------
class Reader {
public:
    Reader(const char* p, size_t n) : p_(p), end_(p + n) {}

    ValuePtr parse_document() {
        Value* root = g_arena.make();
        root->kind = Value::Group;
        parse_settings(*root, /*top=*/true);
        skip_ws();
        if (p_ != end_) fail("unexpected character");
        return root;
    }

private:
    const char* p_;
    const char* end_;
    int depth_ = 0;                       // open '{' / '[' / '(' levels around the value being parsed
    // Nesting bound.  The reference's recursive-descent parser (libconf under CPython's default recursion limit of
    // 1000, several frames per level) raises RecursionError a few hundred levels deep - a catchable exception, the pod
    // is reported and skipped.  Recursing without a bound here would instead overflow the native stack and take the
    // scheduler process down.  Triad configs nest 4-5 levels; anything deeper than this is answered the
    // way the reference answers its own limit: WIRE_RAISE.
    static constexpr int kMaxDepth = 128;
    struct DepthGuard {
        int& d;
        explicit DepthGuard(int& depth) : d(depth) { ++d; }
        ~DepthGuard() { --d; }
    };

    [[noreturn]] void fail(const char* what) const { throw Raise{std::string("libconfig syntax: ") + what}; }

    void skip_ws() {
        for (;;) {
            while (p_ < end_ && is_space(*p_)) ++p_;
            if (p_ < end_ && *p_ == '#') { while (p_ < end_ && *p_ != '\n') ++p_; continue; }
            if (p_ + 1 < end_ && p_[0] == '/' && p_[1] == '/') { while (p_ < end_ && *p_ != '\n') ++p_; continue; }
            if (p_ + 1 < end_ && p_[0] == '/' && p_[1] == '*') {
                p_ += 2;
                while (p_ + 1 < end_ && !(p_[0] == '*' && p_[1] == '/')) ++p_;
                if (p_ + 1 >= end_) fail("unterminated comment");
                p_ += 2;
                continue;
            }
            return;
        }
    }
    bool peek(char c) { skip_ws(); return p_ < end_ && *p_ == c; }
    bool accept(char c) { if (peek(c)) { ++p_; return true; } return false; }

    static bool name_start(char c) { return is_alpha(c) || c == '*'; }
    static bool name_char(char c) { return is_alnum(c) || c == '_' || c == '-' || c == '*'; }

    void parse_settings(Value& group, bool top) {
        for (;;) {
            skip_ws();
            if (p_ == end_) { if (!top) fail("unterminated group"); return; }
            if (*p_ == '}') { if (top) fail("unbalanced '}'"); return; }
            if (*p_ == '@') fail("@include is not supported");
            if (!name_start(*p_)) fail("setting name expected");
            const char* b = p_;
            while (p_ < end_ && name_char(*p_)) ++p_;
            const Value::Name name{b, (size_t)(p_ - b)};
            skip_ws();
            if (p_ == end_ || (*p_ != '=' && *p_ != ':')) fail("'=' or ':' expected");
            ++p_;
            ValuePtr v = parse_value();
            bool replaced = false;                                               // libconf: a repeated name overwrites (dict assignment)
            for (auto& kv : group.fields)
                if (kv.first.n == name.n && std::memcmp(kv.first.p, name.p, name.n) == 0) { kv.second = v; replaced = true; break; }
            if (!replaced) group.fields.emplace_back(name, v);
            skip_ws();
            if (p_ < end_ && (*p_ == ';' || *p_ == ',')) ++p_;
        }
    }

    ValuePtr parse_value() {
        skip_ws();
        if (p_ == end_) fail("value expected");
        const DepthGuard guard(depth_);
        if (depth_ > kMaxDepth) fail("nesting deeper than 128 levels (the reference parser raises RecursionError)");
        Value* v = g_arena.make();
        if (*p_ == '{') {
            ++p_;
            v->kind = Value::Group;
            parse_settings(*v, false);
            if (!accept('}')) fail("'}' expected");
            return v;
        }
        if (*p_ == '[' || *p_ == '(') {
            const bool array = *p_ == '[';
            const char close = array ? ']' : ')';
            ++p_;
            v->kind = array ? Value::Array : Value::List;
            if (accept(close)) return v;
            for (;;) {
                ValuePtr e = parse_value();
                if (array && (e->kind == Value::Array || e->kind == Value::List || e->kind == Value::Group))
                    fail("arrays hold scalars only");
                v->items.push_back(std::move(e));
                if (accept(',')) { if (accept(close)) return v; continue; }      // trailing comma is accepted
                if (accept(close)) return v;
                fail("',' or closing bracket expected");
            }
        }
        return parse_scalar();
    }

    ValuePtr parse_scalar() {
        Value* v = g_arena.make();
        if (*p_ == '"') {
            v->kind = Value::Str;
            while (peek('"')) parse_string_piece(v->s);                           // adjacent strings concatenate
            return v;
        }
        // token classes in libconf's order, first match wins: float, hex, integer, boolean
        const char* q = p_;
        auto digits = [&](const char* c) { while (c < end_ && is_digit(*c)) ++c; return c; };
        {   // float: [-+]?(\d+)?\.\d*([eE][-+]?\d+)?  |  [-+]?\d+(\.\d*)?[eE][-+]?\d+
            const char* c = q;
            if (c < end_ && (*c == '+' || *c == '-')) ++c;
            const char* d = digits(c);
            const char* e = nullptr;
            if (d < end_ && *d == '.') {
                e = digits(d + 1);
                if (e < end_ && (*e == 'e' || *e == 'E')) {
                    const char* x = e + 1;
                    if (x < end_ && (*x == '+' || *x == '-')) ++x;
                    const char* y = digits(x);
                    if (y > x) e = y;
                }
            } else if (d > c && d < end_ && (*d == 'e' || *d == 'E')) {
                const char* x = d + 1;
                if (x < end_ && (*x == '+' || *x == '-')) ++x;
                const char* y = digits(x);
                if (y > x) e = y;
            }
            if (e) {
                bool any_digit = false;
                for (const char* z = q; z < e; ++z) any_digit = any_digit || is_digit(*z);
                if (!any_digit) fail("'.' is matched as a float and is not one");    // libconf: float('.') raises
                v->kind = Value::Float;
                v->f = std::strtod(std::string(q, e).c_str(), nullptr);
                p_ = e;
                return v;
            }
        }
        auto long_suffix = [&](const char* c) { if (c < end_ && *c == 'L') { ++c; if (c < end_ && *c == 'L') ++c; } return c; };
        if (q + 2 < end_ && q[0] == '0' && (q[1] == 'x' || q[1] == 'X') && is_xdigit(q[2])) {
            const char* c = q + 2;
            while (c < end_ && is_xdigit(*c)) ++c;
            v->kind = Value::Int;
            v->i = (long long)std::strtoull(std::string(q + 2, c).c_str(), nullptr, 16);
            p_ = long_suffix(c);
            return v;
        }
        {
            const char* c = q;
            if (c < end_ && (*c == '+' || *c == '-')) ++c;
            const char* d = digits(c);
            if (d > c) {
                v->kind = Value::Int;
                v->i = std::strtoll(std::string(q, d).c_str(), nullptr, 10);
                p_ = long_suffix(d);
                return v;
            }
        }
        for (const char* word : {"true", "false"}) {
            const size_t len = std::strlen(word);
            if ((size_t)(end_ - q) < len) continue;
            bool same = true;
            for (size_t k = 0; k < len; ++k) same = same && to_lower(q[k]) == word[k];
            if (same && (q + len == end_ || !(is_alnum(q[len]) || q[len] == '_'))) {      // \b
                v->kind = Value::Bool;
                v->b = word[0] == 't';
                p_ = q + len;
                return v;
            }
        }
        fail("value expected");
    }

    void parse_string_piece(std::string& out) {
        ++p_;                                                                     // opening quote
        for (;;) {
            if (p_ == end_) fail("unterminated string");
            char c = *p_++;
            if (c == '"') return;
            if (c != '\\') { out.push_back(c); continue; }
            if (p_ == end_) fail("unterminated string");
            c = *p_++;
            switch (c) {
                case 'n': out.push_back('\n'); break;
                case 'r': out.push_back('\r'); break;
                case 't': out.push_back('\t'); break;
                case 'f': out.push_back('\f'); break;
                case '\\': out.push_back('\\'); break;
                case '"': out.push_back('"'); break;
                case 'x': {
                    if (p_ + 1 >= end_ || !is_xdigit(p_[0]) || !is_xdigit(p_[1])) fail("bad \\x escape");
                    out.push_back((char)std::strtol(std::string(p_, p_ + 2).c_str(), nullptr, 16));
                    p_ += 2;
                    break;
                }
                default: fail("unknown escape");
            }
        }
    }
};

// ---- the Python operations the reference applies to the parsed values ------------------------------------------
// magicattr.get(cfg, "a.b[0].c"): the path is parsed as a Python expression (names, attributes, constant
// subscripts) and walked with getattr / subscription.  What is not such an expression (or is another kind of
// expression: "a-b", "f(x)", a keyword) raises; a missing attribute is an AttributeError (AttrMissing here - some call
// sites of the reference catch exactly that), a bad subscript an IndexError / KeyError / TypeError (Raise).
bool py_keyword(const char* w, size_t len) {
    static const char* kw[] = {"False", "None", "True", "and", "as", "assert", "async", "await", "break", "class", "continue",
                               "def", "del", "elif", "else", "except", "finally", "for", "from", "global", "if", "import", "in",
                               "is", "lambda", "nonlocal", "not", "or", "pass", "raise", "return", "try", "while", "with", "yield"};
    if (len < 2 || len > 8) return false;
    for (const char* k : kw)
        if (k[0] == w[0] && std::strncmp(k, w, len) == 0 && k[len] == 0) return true;
    return false;
}

// Paths are composed in one recycled buffer (no temporaries per look-up); valid until the next compose().
class PathBuf {
public:
    PathBuf& start(const std::string& prefix) { s_.assign(prefix); return *this; }
    PathBuf& add(const char* t) { s_.append(t); return *this; }
    PathBuf& add(const std::string& t) { s_.append(t); return *this; }
    PathBuf& index(size_t k) {
        char tmp[24];
        int n = 0;
        do { tmp[n++] = (char)('0' + k % 10); k /= 10; } while (k);
        s_.push_back('[');
        while (n) s_.push_back(tmp[--n]);
        s_.push_back(']');
        return *this;
    }
    const std::string& str() const { return s_; }
private:
    std::string s_;
};
thread_local PathBuf g_path;

struct PathStep { enum { Attr, Index, Key } kind; const char* name; size_t len; unsigned long idx; };   // name points into the path

// phase 1: the whole path must be a valid expression before anything is looked up (ast.parse comes first)
const std::vector<PathStep>& parse_path(const std::string& path) {
    thread_local std::vector<PathStep> steps;                               // recycled: no allocation per look-up
    steps.clear();
    size_t k = 0;
    const size_t n = path.size();
    auto bad = [&](const char* why) -> Raise { return Raise{std::string(why) + " in attribute path '" + path + "'"}; };
    auto skip = [&] { while (k < n && (path[k] == ' ' || path[k] == '\t')) ++k; };
    auto skip_in = [&] { while (k < n && (path[k] == ' ' || path[k] == '\t' || path[k] == '\n')) ++k; };   // inside [ ]: lines join
    auto ident = [&]() -> PathStep {
        const size_t b = k;
        if (k < n && (is_alpha(path[k]) || path[k] == '_')) {
            ++k;
            while (k < n && (is_alnum(path[k]) || path[k] == '_')) ++k;
        }
        if (k == b) throw bad("name expected");
        if (py_keyword(path.data() + b, k - b)) throw bad("keyword");
        return PathStep{PathStep::Attr, path.data() + b, k - b, 0};
    };
    // blank lines (also whitespace-only ones) before the expression are fine; an indented first line is not
    for (;;) {
        size_t e = k;
        while (e < n && (path[e] == ' ' || path[e] == '\t')) ++e;
        if (e < n && path[e] == '\n') { k = e + 1; continue; }
        if (e < n && path[e] == '#') {                                       // a comment line, indented or not
            while (e < n && path[e] != '\n') ++e;
            if (e == n) throw bad("empty expression");
            k = e + 1;
            continue;
        }
        if (e == n) throw bad("empty expression");
        if (e != k) throw bad("indented expression");
        break;
    }
    steps.push_back(ident());
    for (;;) {
        skip();
        if (k == n) break;
        if (path[k] == '#') break;                                          // a Python comment: the rest is not looked at
        if (path[k] == ';' || path[k] == '\n') {                            // end of the statement: only blanks / a comment may follow
            while (k < n && (path[k] == ';' || path[k] == '\n' || path[k] == ' ' || path[k] == '\t')) ++k;
            if (k < n && path[k] != '#') throw bad("more than one statement");
            break;
        }
        if (path[k] == '.') {
            ++k; skip();
            steps.push_back(ident());
            continue;
        }
        if (path[k] != '[') throw bad("unexpected character");
        ++k; skip_in();
        if (k < n && (path[k] == '\'' || path[k] == '"')) {                 // string subscript: dict key
            const char q = path[k++];
            const size_t b = k;
            while (k < n && path[k] != q) { if (path[k] == '\\' || path[k] == '\n') throw bad("escape in subscript"); ++k; }
            if (k == n) throw bad("unterminated string");
            steps.push_back({PathStep::Key, path.data() + b, k - b, 0});
            ++k; skip_in();
            if (k == n || path[k] != ']') throw bad("']' expected");
            ++k;
            continue;
        }
        std::string digits;
        bool prev_digit = false;
        while (k < n && (is_digit(path[k]) || path[k] == '_')) {
            if (path[k] == '_') { if (!prev_digit || k + 1 >= n || !is_digit(path[k + 1])) throw bad("bad integer"); prev_digit = false; }
            else { digits.push_back(path[k]); prev_digit = true; }
            ++k;
        }
        if (digits.empty()) throw bad("constant subscript expected");
        if (digits.size() > 1 && digits[0] == '0' && digits.find_first_not_of('0') != std::string::npos) throw bad("leading zeros");
        skip_in();
        if (k == n || path[k] != ']') throw bad("']' expected");
        ++k;
        steps.push_back({PathStep::Index, nullptr, 0, digits.size() > 9 ? 999999999ul : std::strtoul(digits.c_str(), nullptr, 10)});
    }
    return steps;
}

// phase 2: getattr / subscription, left to right
const Value& lookup(const Value& root, const std::string& path) {
    const std::vector<PathStep>& steps = parse_path(path);
    const Value* cur = &root;
    for (const PathStep& st : steps) {
        if (st.kind == PathStep::Attr) {
            const Value* next = cur->kind == Value::Group ? cur->find(st.name, st.len) : nullptr;
            if (!next) throw AttrMissing{"no attribute '" + std::string(st.name, st.len) + "' in '" + path + "'"};
            cur = next;
        } else if (st.kind == PathStep::Key) {
            const Value* next = cur->kind == Value::Group ? cur->find(st.name, st.len) : nullptr;
            if (!next) throw Raise{"bad string subscript in '" + path + "'"};      // KeyError / TypeError
            cur = next;
        } else if (cur->is_seq()) {
            if (st.idx >= cur->items.size()) throw Raise{"index out of range in '" + path + "'"};
            cur = cur->items[st.idx];
        } else if (cur->kind == Value::Str) {                                    // 'abc'[1] is a one-character string
            if (st.idx >= cur->s.size()) throw Raise{"index out of range in '" + path + "'"};
            Value* ch = g_arena.make();
            ch->kind = Value::Str;
            ch->s = cur->s.substr(st.idx, 1);
            cur = ch;
        } else {
            throw Raise{"subscript of a group or number in '" + path + "'"};     // KeyError / TypeError
        }
    }
    return *cur;
}

const Value& attr(const Value& g, const char* name) {                       // obj.name on an AttrDict
    const Value* v = g.kind == Value::Group ? g.find(name) : nullptr;
    if (!v) throw AttrMissing{std::string("no attribute '") + name + "'"};
    return *v;
}

long long py_int(const Value& v) {                                         // int(x)
    switch (v.kind) {
        case Value::Int: return v.i;
        case Value::Bool: return v.b ? 1 : 0;
        case Value::Float:
            if (!std::isfinite(v.f)) throw Raise{"int() of a non-finite float"};
            return (long long)std::trunc(v.f);
        case Value::Str: {
            size_t a = 0, b = v.s.size();
            while (a < b && is_space(v.s[a])) ++a;
            while (b > a && is_space(v.s[b - 1])) --b;
            std::string t = v.s.substr(a, b - a), digits;
            size_t k = 0;
            bool neg = false;
            if (k < t.size() && (t[k] == '+' || t[k] == '-')) { neg = t[k] == '-'; ++k; }
            bool prev_digit = false;
            for (; k < t.size(); ++k) {
                if (is_digit(t[k])) { digits.push_back(t[k]); prev_digit = true; }
                else if (t[k] == '_' && prev_digit && k + 1 < t.size() && is_digit(t[k + 1])) prev_digit = false;
                else throw Raise{"int() of a non-numeric string"};
            }
            if (digits.empty()) throw Raise{"int() of a non-numeric string"};
            const long long m = std::strtoll(digits.c_str(), nullptr, 10);
            return neg ? -m : m;
        }
        default: throw Raise{"int() of a group / list"};
    }
}

bool truthy(const Value& v) {                                              // bool(x)
    switch (v.kind) {
        case Value::Int: return v.i != 0;
        case Value::Float: return v.f != 0.0;
        case Value::Bool: return v.b;
        case Value::Str: return !v.s.empty();
        case Value::Group: return !v.fields.empty();
        default: return !v.items.empty();
    }
}

size_t py_len(const Value& v) {                                            // len(x)
    switch (v.kind) {
        case Value::Str: return v.s.size();
        case Value::Group: return v.fields.size();
        case Value::Array: case Value::List: return v.items.size();
        default: throw Raise{"len() of a scalar"};
    }
}

// NIC speeds are stored as they come (TriadCfgParser.py:205, 210) and only added up in FindNode
// (CfgTopology.py:219-232: 0 + s1 + s2 ...): a non-number parses fine there and raises later.  Here: remembered,
// reported as NHDFIT_WIRE_RAISE once the parse itself has succeeded.
double speed_of(const Value& v, bool& bad) {
    switch (v.kind) {
        case Value::Int: return (double)v.i;
        case Value::Float: return v.f;
        case Value::Bool: return v.b ? 1.0 : 0.0;
        default: bad = true; return 0.0;
    }
}

std::string py_format(const Value& v) {                                    // f"{x}" for the values that can name something
    switch (v.kind) {
        case Value::Str: return v.s;
        case Value::Int: return std::to_string(v.i);
        case Value::Bool: return v.b ? "True" : "False";
        default: return "<unnamed>";                                        // floats / containers: never a valid path element
    }
}

const std::string& str_of(const Value& v, const char* what) {
    if (v.kind != Value::Str) throw Raise{std::string(what) + " is not a string"};
    return v.s;
}

// key of the reference's gpumap dict (TriadCfgParser.py:231-240): Python equality / hashing of scalars
struct GpuKey {
    bool is_str; double num; std::string s;
    bool operator==(const GpuKey& o) const { return is_str == o.is_str && (is_str ? s == o.s : num == o.num); }
};
GpuKey gpu_key(const Value& v) {
    switch (v.kind) {
        case Value::Int: return {false, (double)v.i, {}};
        case Value::Float: return {false, v.f, {}};
        case Value::Bool: return {false, v.b ? 1.0 : 0.0, {}};
        case Value::Str: return {true, 0, v.s};
        default: throw Raise{"GPU device id is not a scalar"};
    }
}

struct GroupTotals {
    unsigned proc = 0, help = 0, gpus = 0;
    bool proc_smt = false, helper_smt = false, nic_use = false;
    double rx = 0, tx = 0;
    unsigned nic_pairs = 0;            // RX / TX core pairs of the group: more than one -> rx / tx are sums (NHDFIT_RF_NIC_SPLIT)
    bool dyadic = true;                // every speed a non-negative multiple of 2^-20 below 2^31 (NHDFIT_RF_NIC_SPLIT_DYADIC)
    void note(double v) { if (!(v >= 0.0) || !(v < 2147483648.0) || std::floor(v * 1048576.0) != v * 1048576.0) dyadic = false; }
};

// nhd/TriadCfgParser.py:134-309
void parse_mod_groups(const Value& cfg, const Value& topo, std::vector<GroupTotals>& groups, uint32_t& map_type, bool& bad_speed) {
    if (!topo.has("mod_defs")) throw Reject{"no mod_defs in TopologyCfg"};
    if (!topo.has("map_type")) throw Reject{"no map_type in TopologyCfg"};
    const Value& mt = attr(topo, "map_type");
    map_type = NHDFIT_MAP_INVALID;                                           // SetTopMapType: anything else stays INVALID
    if (mt.kind == Value::Str && mt.s == "NUMA") map_type = NHDFIT_MAP_NUMA;
    else if (mt.kind == Value::Str && mt.s == "PCI") map_type = NHDFIT_MAP_PCI;

    const Value& mod_defs = attr(topo, "mod_defs");
    if (!mod_defs.is_seq()) throw Raise{"mod_defs is not a list"};
    for (const auto& mdp : mod_defs.items) {
        const Value& md = *mdp;
        const Value& module = attr(md, "module");
        if (module.kind != Value::Str || !cfg.has(module.s)) throw Reject{"module not found at top level"};
        const Value& instances = *cfg.find(module.s);
        if (!instances.is_seq()) throw Raise{"module section is not a list of instances"};
        for (size_t idx = 0; idx < instances.items.size(); ++idx) {
            const Value& mi = *instances.items[idx];
            (void)attr(mi, "module");                                       // the reference formats mi.module into its log line
            GroupTotals pg;
            const std::string mattr = module.s + "[" + std::to_string(idx) + "]";

            if (md.has("helper_cores")) {
                if (!md.has("helper_cores_smt")) throw Reject{"helper_cores_smt not defined"};
                pg.helper_smt = truthy(attr(md, "helper_cores_smt"));
                const Value& hcs = attr(md, "helper_cores");
                if (!hcs.is_seq()) throw Raise{"helper_cores is not a list"};
                for (const auto& hc : hcs.items) {
                    const std::string name = mattr + "." + str_of(*hc, "helper core name");
                    const Value& a = lookup(cfg, name);
                    if (a.kind == Value::Array) {                           // libconf: [..] -> list, (..) -> tuple
                        for (size_t e = 0; e < a.items.size(); ++e) {       // the reference looks every element up by its own path
                            (void)py_int(lookup(cfg, g_path.start(name).index(e).str()));
                            pg.help++;
                        }
                    } else {
                        (void)py_int(a);
                        pg.help++;
                    }
                }
            }

            if (md.has("dp_group")) {
                const Value& dpg = attr(md, "dp_group");
                const Value* dp = nullptr;
                // the bare `except:` around this look-up logs md.dp_group.name again (TriadCfgParser.py:195): without a
                // name the handler itself raises; with one, any failure of the look-up is a logged error -> None
                const Value& dp_name = attr(dpg, "name");
                try {
                    dp = &lookup(cfg, mattr + "." + py_format(dp_name));
                } catch (const AttrMissing&) { throw Reject{"dp group attribute not found"}; }
                  catch (const Raise&) { throw Reject{"dp group attribute not found"}; }
                if (py_len(*dp) != 1) throw Reject{"DP groups of multiple NUMA nodes not supported"};
                if (!dp->is_seq()) throw Raise{"dp group is not a list"};
                const Value& d0 = *dp->items[0];
                // chained comparison of the reference (TriadCfgParser.py:193): `a != b != c != d` is
                // (a != b) and (b != c) and (c != d), evaluated left to right and short-circuited - the speed lists
                // are not even looked up when the two core lists have equal length
                const Value &rxc = attr(d0, "rx_cores"), &txc = attr(d0, "tx_cores");
                if (py_len(rxc) != py_len(txc)) {
                    const Value& rxs0 = attr(d0, "rx_speeds");
                    if (py_len(txc) != py_len(rxs0) && py_len(rxs0) != py_len(attr(d0, "tx_speeds")))
                        throw Reject{"core / speed list lengths differ"};
                }
                pg.proc_smt = truthy(attr(dpg, "proc_cores_smt"));
                // every core and speed is looked up afresh by its composed path, as the reference does
                // (f'{mattr}.{md.dp_group.name}[0].rx_cores[{gidx}]', TriadCfgParser.py:203-215)
                const std::string dpath = mattr + "." + py_format(dp_name) + "[0].";
                auto elem = [&](const char* field, size_t g) -> const Value& {
                    return lookup(cfg, g_path.start(dpath).add(field).index(g).str());
                };
                try {
                    const size_t n = py_len(rxc);
                    for (size_t g = 0; g < n; ++g) {
                        const double rs = speed_of(elem("rx_speeds", g), bad_speed);
                        (void)py_int(elem("rx_cores", g));
                        pg.proc++; pg.rx += rs; pg.nic_use = true; pg.nic_pairs++; pg.note(rs);
                        const double ts = speed_of(elem("tx_speeds", g), bad_speed);
                        (void)py_int(elem("tx_cores", g));
                        pg.proc++; pg.tx += ts; pg.note(ts);
                    }
                } catch (const Raise&) { throw Reject{"error when parsing NIC fields"}; }
                  catch (const AttrMissing&) { throw Reject{"error when parsing NIC fields"}; }
                if (d0.kind == Value::Group && d0.has("cpu_workers")) try {    // CPU workers: optional, errors are swallowed
                    const size_t n = py_len(attr(d0, "cpu_workers"));
                    for (size_t c = 0; c < n; ++c) {
                        (void)py_int(elem("cpu_workers", c));
                        pg.proc++;
                    }
                } catch (const Raise&) {} catch (const AttrMissing&) {}
                const Value& gm = attr(d0, "gpu_map");
                const size_t ng = py_len(gm);
                if (ng && !gm.is_seq()) throw Raise{"gpu_map is not a list"};
                std::vector<GpuKey> keys;
                std::vector<std::vector<size_t>> cores;                      // gpu_map entries (by index) of each device
                for (size_t g = 0; g < ng; ++g) {
                    const Value& e = *gm.items[g];
                    if (py_len(e) != 2) continue;                            // logged, not processed
                    if (!e.is_seq()) throw Raise{"gpu_map entry is not a pair"};
                    const GpuKey key = gpu_key(*e.items[1]);
                    size_t k = 0;
                    while (k < keys.size() && !(keys[k] == key)) ++k;
                    if (k == keys.size()) { keys.push_back(key); cores.emplace_back(); }
                    cores[k].push_back(g);
                }
                for (const auto& cl : cores) {
                    for (size_t g : cl) {                                     // the GPU's feeder cores count as proc cores
                        (void)py_int(lookup(cfg, g_path.start(dpath).add("gpu_map").index(g).add("[0]").str()));
                        pg.proc++;
                    }
                    pg.gpus++;
                }
            }

            if (md.has("nic_cores")) {
                const Value& nc = attr(md, "nic_cores");
                if (py_len(nc) != 5) throw Reject{"wrong number of parameters for nic_cores"};
                if (!nc.is_seq()) throw Raise{"nic_cores is not a list"};
                const Value *rxc, *rxs, *txc, *txs;
                try {
                    auto get = [&](size_t k) -> const Value* {
                        const Value& nm = *nc.items[k];
                        if (nm.kind != Value::Str) throw Raise{"nic_cores name"};
                        return &lookup(cfg, mattr + "." + nm.s);
                    };
                    rxc = get(0); rxs = get(1); txc = get(2); txs = get(3);
                } catch (const Raise&) { throw Reject{"could not find NIC attributes"}; }
                  catch (const AttrMissing&) { throw Reject{"could not find NIC attributes"}; }
                if (py_len(*rxc) != py_len(*rxs) && py_len(*rxs) != py_len(*txc) && py_len(*txc) != py_len(*txs))   // && short-circuits like the chain
                    throw Reject{"speed and core lengths differ"};
                pg.proc_smt = truthy(*nc.items[4]);
                const size_t n = py_len(*rxc);
                auto elem = [&](size_t which, size_t g) -> const Value& {     // f'{mattr}.{md.nic_cores[which]}[{g}]', looked up afresh
                    return lookup(cfg, g_path.start(mattr).add(".").add(nc.items[which]->s).index(g).str());
                };
                for (size_t g = 0; g < n; ++g) {
                    const double rs = speed_of(elem(1, g), bad_speed);
                    (void)py_int(elem(0, g));
                    pg.proc++; pg.rx += rs; pg.nic_use = true; pg.nic_pairs++; pg.note(rs);
                    const double ts = speed_of(elem(3, g), bad_speed);
                    (void)py_int(elem(2, g));
                    pg.proc++; pg.tx += ts; pg.note(ts);
                }
            }
            groups.push_back(pg);
        }
    }
}

unsigned half_up(unsigned n) { return (n + 1) / 2; }                        // math.ceil(n / 2.0)

// R = nhdfit_req (up to four processing groups) or nhdfit_big_req (up to eight: the general path's request record)
template <class R>
void digest(const char* text, size_t len, R& r) {
    constexpr unsigned kGroups = sizeof(r.gpus) / sizeof(r.gpus[0]);
    g_arena.reset();
    Reader reader(text, len);
    const ValuePtr rootp = reader.parse_document();
    const Value& cfg = *rootp;
    // CfgToTopology, nhd/TriadCfgParser.py:337-380
    if (!cfg.has("TopologyCfg")) throw Reject{"no TopologyCfg section"};
    const Value& topo = *cfg.find("TopologyCfg");
    if (topo.kind != Value::Group) throw Raise{"TopologyCfg is not a group"};
    for (const char* f : {"cpu_arch", "ext_cores", "kni_vlan"})
        if (!topo.has(f)) throw Reject{std::string("mandatory field ") + f + " missing"};
    {
        const Value& arch = attr(topo, "cpu_arch");
        static const char* known[] = {"ANY", "HASWELL", "BROADWELL", "SKYLAKE", "COOPER_LAKE", "ICE_LAKE"};
        bool ok = false;
        for (const char* k : known) ok = ok || (arch.kind == Value::Str && arch.s == k);
        if (!ok) throw Reject{"unknown cpu_arch"};
    }
    // ParseMiscCores, :106-132
    if (!topo.has("ext_cores_smt")) throw Reject{"no ext_cores_smt"};
    const bool misc_smt = truthy(attr(topo, "ext_cores_smt"));
    unsigned n_misc = 0;
    {
        const Value& ext = attr(topo, "ext_cores");
        // `for i in ext_cores`: a list / array yields its elements, a string its characters, a group its setting names
        auto one = [&](const std::string& name) {
            try {
                (void)py_int(lookup(cfg, name));
            } catch (const AttrMissing&) { throw Reject{"ext core not found"}; }     // the one exception type the reference catches here
            n_misc++;
        };
        if (ext.is_seq()) {
            for (const auto& e : ext.items) one(str_of(*e, "ext core name"));        // in order: an earlier failure wins
        } else if (ext.kind == Value::Str) {
            for (char ch : ext.s) one(std::string(1, ch));
        } else if (ext.kind == Value::Group) {
            for (const auto& kv : ext.fields) one(kv.first.str());
        } else {
            throw Raise{"ext_cores is not iterable"};
        }
    }
    // ParseKniDataVlan (:80-90) only stores a name.  ParseModGroups:
    std::vector<GroupTotals> groups;
    uint32_t map_type = NHDFIT_MAP_INVALID;
    bool bad_speed = false;
    try {
        parse_mod_groups(cfg, topo, groups, map_type, bad_speed);
    } catch (const AttrMissing& e) { throw Raise{e.why}; }                     // uncaught AttributeError in the reference
    // ParseHugePages, :92-104
    if (!cfg.has("Hugepages_GB")) throw Reject{"no Hugepages_GB"};
    const long long hp = py_int(*cfg.find("Hugepages_GB"));
    if (bad_speed) throw Raise{"a NIC speed is not a number (the reference fails when FindNode adds the speeds up)"};

    // the integers FindNode derives from the object graph (nhd/CfgTopology.py:199-232, nhd/Matcher.py:178-204)
    if (groups.size() > kGroups) throw Limit{kGroups == NHDFIT_MAX_GROUPS ? "more proc groups than NHDFIT_MAX_GROUPS" : "more proc groups than NHDFIT_BIG_MAX_GROUPS"};
    std::memset(&r, 0, sizeof r);
    r.n_groups = (uint32_t)groups.size();
    r.map_type = map_type;
    r.hugepages_gb = (int32_t)(hp < INT32_MIN ? INT32_MIN : hp > INT32_MAX ? INT32_MAX : hp);
    bool split_dyadic = true;
    for (size_t i = 0; i < groups.size(); ++i) {
        const GroupTotals& g = groups[i];
        if (g.proc > 255 || g.help > 255) throw Limit{"a proc group asks for more than 255 cores"};
        r.gpus[i] = (uint16_t)g.gpus;
        r.n_proc[i] = (uint8_t)g.proc;
        r.n_help[i] = (uint8_t)g.help;
        if (g.proc_smt) r.smt_bits |= (decltype(r.smt_bits))(1u << i);
        if (g.helper_smt) r.smt_bits |= (decltype(r.smt_bits))(1u << (kGroups + i));
        r.cpu_nosmt[i] = (uint16_t)(g.proc + g.help);
        r.cpu_smt[i] = (uint16_t)((g.proc_smt ? half_up(g.proc) : g.proc) + (g.helper_smt ? half_up(g.help) : g.help));
        r.rx[i] = g.rx;
        r.tx[i] = g.tx;
        if (g.nic_use) r.nic_use |= (uint8_t)(1u << i);
        if (g.nic_pairs > 1) { r.flags |= NHDFIT_RF_NIC_SPLIT; split_dyadic = split_dyadic && g.dyadic; }
    }
    if ((r.flags & NHDFIT_RF_NIC_SPLIT) && split_dyadic) r.flags |= NHDFIT_RF_NIC_SPLIT_DYADIC;
    if (n_misc > 65535) throw Limit{"too many misc cores"};
    r.misc_nosmt = (uint16_t)n_misc;
    r.n_misc = (uint8_t)(n_misc > 255 ? 255 : n_misc);
    r.misc_smt_enabled = misc_smt ? 1 : 0;
    // Matcher.py:198 tests the truthiness of the SMTSetting enum member, which is always true (quirk Q1)
    r.misc_smt = (uint16_t)half_up(n_misc);
}

void set_err(char* err, size_t errlen, const std::string& msg) {
    if (err && errlen) std::snprintf(err, errlen, "%s", msg.c_str());
}

}  // namespace

extern "C" int nhdfit_digest_triad_config(const char* text, size_t len, nhdfit_req* out, char* err, size_t errlen) {
    if (!text || !out) { set_err(err, errlen, "null argument"); return NHDFIT_E_INVAL; }
    try {
        digest(text, len, *out);
        set_err(err, errlen, "");
        return NHDFIT_OK;
    } catch (const Reject& e) { set_err(err, errlen, e.why); return NHDFIT_WIRE_NONE; }
      catch (const Raise& e) { set_err(err, errlen, e.why); return NHDFIT_WIRE_RAISE; }
      catch (const AttrMissing& e) { set_err(err, errlen, e.why); return NHDFIT_WIRE_RAISE; }
      catch (const Limit& e) { set_err(err, errlen, e.why); return NHDFIT_WIRE_LIMIT; }
      catch (const std::exception& e) { set_err(err, errlen, e.what()); return NHDFIT_E_INVAL; }
      catch (...) { set_err(err, errlen, "unknown failure"); return NHDFIT_E_INVAL; }
}

// the same text into the general path's request record (5..8 processing groups; include/nhdfit.h nhdfit_big_req)
extern "C" int nhdfit_digest_triad_config_big(const char* text, size_t len, nhdfit_big_req* out, char* err, size_t errlen) {
    if (!text || !out) { set_err(err, errlen, "null argument"); return NHDFIT_E_INVAL; }
    try {
        digest(text, len, *out);
        set_err(err, errlen, "");
        return NHDFIT_OK;
    } catch (const Reject& e) { set_err(err, errlen, e.why); return NHDFIT_WIRE_NONE; }
      catch (const Raise& e) { set_err(err, errlen, e.why); return NHDFIT_WIRE_RAISE; }
      catch (const AttrMissing& e) { set_err(err, errlen, e.why); return NHDFIT_WIRE_RAISE; }
      catch (const Limit& e) { set_err(err, errlen, e.why); return NHDFIT_WIRE_LIMIT; }
      catch (const std::exception& e) { set_err(err, errlen, e.what()); return NHDFIT_E_INVAL; }
      catch (...) { set_err(err, errlen, "unknown failure"); return NHDFIT_E_INVAL; }
}

// n texts at once (one FFI crossing per batch of pending pods): codes[i] = what nhdfit_digest_triad_config returns
// for text i; out[i] is written (zeroed unless the code is 0).  Returns the number of texts with a non-zero code.
extern "C" int nhdfit_digest_triad_configs(const char* const* texts, const size_t* lens, uint32_t n, nhdfit_req* out, int32_t* codes) {
    if (!texts || !lens || !out || !codes) return NHDFIT_E_INVAL;
    int bad = 0;
    for (uint32_t i = 0; i < n; ++i) {
        std::memset(&out[i], 0, sizeof out[i]);
        nhdfit_req r;
        codes[i] = nhdfit_digest_triad_config(texts[i], lens[i], &r, nullptr, 0);
        if (codes[i] == NHDFIT_OK) out[i] = r; else ++bad;
    }
    return bad;
}
