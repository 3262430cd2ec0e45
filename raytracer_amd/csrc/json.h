// json.h -- minimal JSON DOM (parse + a few writers) for the string-typed
// entry points the reference exposes (main.go:39-344 pass every argument and
// result as a JSON string).  Object member lookup is case-insensitive, as Go's
// encoding/json matches struct fields.
#pragma once

#include <cctype>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace rtjson {

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;

    bool is_null() const { return kind == Null; }
    const Value *get(const char *name) const {
        if (kind != Object) return nullptr;
        for (const auto &kv : obj)
            if (kv.first == name) return &kv.second;
        for (const auto &kv : obj) {   // Go falls back to a case-insensitive match
            if (kv.first.size() != std::strlen(name)) continue;
            bool eq = true;
            for (size_t i = 0; i < kv.first.size() && eq; ++i)
                eq = std::tolower((unsigned char)kv.first[i]) == std::tolower((unsigned char)name[i]);
            if (eq) return &kv.second;
        }
        return nullptr;
    }
    double number_or(const char *name, double dflt) const {
        const Value *v = get(name);
        if (!v) return dflt;
        if (v->kind == Number) return v->num;
        return dflt;
    }
    bool bool_or(const char *name, bool dflt) const {
        const Value *v = get(name);
        return (v && v->kind == Bool) ? v->b : dflt;
    }
    std::string string_or(const char *name, const std::string &dflt) const {
        const Value *v = get(name);
        return (v && v->kind == String) ? v->str : dflt;
    }
};

// Token-level reader: the primitives the DOM parser below is made of, also used
// directly where a document is too large for a DOM (the BVH hand-over text).
class Lexer {
  public:
    Lexer(const char *s, size_t n) : p_(s), end_(s + n) {}
    const std::string &error() const { return err_; }
    bool fail(const char *m) {
        if (err_.empty()) err_ = m;
        return false;
    }
    void skip_ws() {
        while (p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) ++p_;
    }
    bool at_end() {
        skip_ws();
        return p_ == end_;
    }
    // next significant character, 0 at the end of the text
    char peek() {
        skip_ws();
        return p_ < end_ ? *p_ : '\0';
    }
    bool consume(char c) {
        skip_ws();
        if (p_ < end_ && *p_ == c) {
            ++p_;
            return true;
        }
        return false;
    }
    bool literal(const char *lit) {
        skip_ws();
        size_t n = std::strlen(lit);
        if ((size_t)(end_ - p_) < n || std::memcmp(p_, lit, n) != 0) return fail("invalid literal");
        p_ += n;
        return true;
    }
    bool string(std::string &out) {
        skip_ws();
        if (p_ >= end_ || *p_ != '"') return fail("expected string");
        ++p_;
        out.clear();
        while (p_ < end_) {
            // bulk-copy the run up to the next quote or backslash
            const char *q = p_;
            while (q < end_ && *q != '"' && *q != '\\') ++q;
            out.append(p_, q);
            p_ = q;
            if (p_ >= end_) break;
            if (*p_ == '"') {
                ++p_;
                return true;
            }
            ++p_;   // backslash
            if (p_ >= end_) break;
            char c = *p_++;
            switch (c) {
            case '"': out.push_back('"'); break;
            case '\\': out.push_back('\\'); break;
            case '/': out.push_back('/'); break;
            case 'b': out.push_back('\b'); break;
            case 'f': out.push_back('\f'); break;
            case 'n': out.push_back('\n'); break;
            case 'r': out.push_back('\r'); break;
            case 't': out.push_back('\t'); break;
            case 'u': {
                unsigned cp;
                if (!hex4(cp)) return false;
                if (cp >= 0xD800 && cp < 0xDC00 && end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
                    p_ += 2;
                    unsigned lo;
                    if (!hex4(lo)) return false;
                    cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                }
                put_utf8(out, cp);
                break;
            }
            default: return fail("bad escape");
            }
        }
        return fail("unterminated string");
    }
    bool number(double &out) {
        skip_ws();
        const char *q = p_;
        if (q < end_ && *q == '-') ++q;
        if (q >= end_ || *q < '0' || *q > '9') return fail("expected number");
        while (q < end_ && ((*q >= '0' && *q <= '9') || *q == '.' || *q == 'e' || *q == 'E' || *q == '+' || *q == '-')) ++q;
        // from_chars takes no leading '+', like JSON; it must consume the whole token
        const auto r = std::from_chars(p_, q, out);
        if (r.ec != std::errc() || r.ptr != q) return fail("bad number");
        p_ = q;
        return true;
    }
    // Steps over one value of any kind, checking its syntax.
    bool skip_value(int depth = 0) {
        if (depth > 4096) return fail("JSON nested too deeply");
        const char c = peek();
        if (c == '{') {
            ++p_;
            if (consume('}')) return true;
            std::string key;
            for (;;) {
                if (!string(key)) return false;
                if (!consume(':')) return fail("expected ':'");
                if (!skip_value(depth + 1)) return false;
                if (consume(',')) continue;
                if (consume('}')) return true;
                return fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            ++p_;
            if (consume(']')) return true;
            for (;;) {
                if (!skip_value(depth + 1)) return false;
                if (consume(',')) continue;
                if (consume(']')) return true;
                return fail("expected ',' or ']'");
            }
        }
        if (c == '"') {
            std::string s;
            return string(s);
        }
        if (c == 't') return literal("true");
        if (c == 'f') return literal("false");
        if (c == 'n') return literal("null");
        if (c == '-' || (c >= '0' && c <= '9')) {
            double d;
            return number(d);
        }
        return fail(c == '\0' ? "unexpected end of JSON" : "unexpected character in JSON");
    }

  private:
    const char *p_, *end_;
    std::string err_;

    static void put_utf8(std::string &s, unsigned cp) {
        if (cp < 0x80) {
            s.push_back((char)cp);
        } else if (cp < 0x800) {
            s.push_back((char)(0xC0 | (cp >> 6)));
            s.push_back((char)(0x80 | (cp & 0x3F)));
        } else if (cp < 0x10000) {
            s.push_back((char)(0xE0 | (cp >> 12)));
            s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            s.push_back((char)(0x80 | (cp & 0x3F)));
        } else {
            s.push_back((char)(0xF0 | (cp >> 18)));
            s.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
            s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            s.push_back((char)(0x80 | (cp & 0x3F)));
        }
    }
    bool hex4(unsigned &out) {
        if (end_ - p_ < 4) return fail("short \\u escape");
        out = 0;
        for (int i = 0; i < 4; ++i) {
            char c = *p_++;
            out <<= 4;
            if (c >= '0' && c <= '9') out |= (unsigned)(c - '0');
            else if (c >= 'a' && c <= 'f') out |= (unsigned)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') out |= (unsigned)(c - 'A' + 10);
            else return fail("bad \\u escape");
        }
        return true;
    }
};

// ASCII case-insensitive comparison: encoding/json matches struct fields this
// way when no exact match exists.
inline bool key_is(const std::string &key, const char *name) {
    const size_t n = std::strlen(name);
    if (key.size() != n) return false;
    for (size_t i = 0; i < n; ++i)
        if (std::tolower((unsigned char)key[i]) != std::tolower((unsigned char)name[i])) return false;
    return true;
}

class Parser {
  public:
    Parser(const char *s, size_t n) : lx_(s, n) {}
    bool parse(Value &out, std::string &err) {
        if (!value(out, 0)) {
            err = lx_.error().empty() ? "invalid JSON" : lx_.error();
            return false;
        }
        if (!lx_.at_end()) {
            err = "trailing characters after JSON value";
            return false;
        }
        return true;
    }

  private:
    Lexer lx_;

    bool value(Value &out, int depth) {
        if (depth > 4096) return lx_.fail("JSON nested too deeply");
        const char c = lx_.peek();
        if (c == '\0') return lx_.fail("unexpected end of JSON");
        if (c == '{') {
            lx_.consume('{');
            out.kind = Value::Object;
            if (lx_.consume('}')) return true;
            out.obj.reserve(8);
            for (;;) {
                std::string key;
                if (!lx_.string(key)) return false;
                if (!lx_.consume(':')) return lx_.fail("expected ':'");
                out.obj.emplace_back(std::move(key), Value());
                if (!value(out.obj.back().second, depth + 1)) return false;
                if (lx_.consume(',')) continue;
                if (lx_.consume('}')) return true;
                return lx_.fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            lx_.consume('[');
            out.kind = Value::Array;
            if (lx_.consume(']')) return true;
            out.arr.reserve(4);
            for (;;) {
                out.arr.emplace_back();
                if (!value(out.arr.back(), depth + 1)) return false;
                if (lx_.consume(',')) continue;
                if (lx_.consume(']')) return true;
                return lx_.fail("expected ',' or ']'");
            }
        }
        if (c == '"') {
            out.kind = Value::String;
            return lx_.string(out.str);
        }
        if (c == 't') {
            out.kind = Value::Bool;
            out.b = true;
            return lx_.literal("true");
        }
        if (c == 'f') {
            out.kind = Value::Bool;
            out.b = false;
            return lx_.literal("false");
        }
        if (c == 'n') {
            out.kind = Value::Null;
            return lx_.literal("null");
        }
        if (c == '-' || (c >= '0' && c <= '9')) {
            out.kind = Value::Number;
            return lx_.number(out.num);
        }
        return lx_.fail("unexpected character in JSON");
    }
};

inline bool parse(const char *s, size_t n, Value &out, std::string &err) {
    Parser p(s, n);
    return p.parse(out, err);
}

// Shortest decimal text that round-trips a float32, in the form Go's
// encoding/json writes it (strconv 'f' format, 'e' below 1e-6 and from 1e21
// on, with "e-09" shortened to "e-9").
inline void append_f32(std::string &out, float f) {
    if (!std::isfinite(f)) {   // encoding/json refuses these; keep the output parseable
        out += "0";
        return;
    }
    char buf[64];
    const float a = std::fabs(f);
    const bool sci = a != 0.0f && (a < 1e-6f || a >= 1e21f);
    auto r = std::to_chars(buf, buf + sizeof buf, f,
                           sci ? std::chars_format::scientific : std::chars_format::fixed);
    size_t n = (size_t)(r.ptr - buf);
    if (sci && n >= 4 && buf[n - 4] == 'e' && buf[n - 3] == '-' && buf[n - 2] == '0') {
        buf[n - 2] = buf[n - 1];
        --n;
    }
    out.append(buf, n);
}

inline void append_escaped(std::string &out, const std::string &s) {
    out.push_back('"');
    for (unsigned char c : s) {
        switch (c) {
        case '"': out += "\\\""; break;
        case '\\': out += "\\\\"; break;
        case '\n': out += "\\n"; break;
        case '\r': out += "\\r"; break;
        case '\t': out += "\\t"; break;
        default:
            if (c < 0x20) {
                char buf[8];
                std::snprintf(buf, sizeof buf, "\\u%04x", c);
                out += buf;
            } else {
                out.push_back((char)c);
            }
        }
    }
    out.push_back('"');
}

}  // namespace rtjson
