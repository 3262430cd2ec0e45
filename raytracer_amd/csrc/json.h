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

class Parser {
  public:
    Parser(const char *s, size_t n) : p_(s), end_(s + n) {}
    bool parse(Value &out, std::string &err) {
        skip_ws();
        if (!value(out, 0)) {
            err = err_.empty() ? "invalid JSON" : err_;
            return false;
        }
        skip_ws();
        if (p_ != end_) {
            err = "trailing characters after JSON value";
            return false;
        }
        return true;
    }

  private:
    const char *p_, *end_;
    std::string err_;

    void skip_ws() {
        while (p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) ++p_;
    }
    bool fail(const char *m) {
        if (err_.empty()) err_ = m;
        return false;
    }
    bool literal(const char *lit) {
        size_t n = std::strlen(lit);
        if ((size_t)(end_ - p_) < n || std::memcmp(p_, lit, n) != 0) return fail("invalid literal");
        p_ += n;
        return true;
    }
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
    bool string(std::string &out) {
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
    bool value(Value &out, int depth) {
        if (depth > 4096) return fail("JSON nested too deeply");
        skip_ws();
        if (p_ >= end_) return fail("unexpected end of JSON");
        char c = *p_;
        if (c == '{') {
            ++p_;
            out.kind = Value::Object;
            skip_ws();
            if (p_ < end_ && *p_ == '}') {
                ++p_;
                return true;
            }
            for (;;) {
                skip_ws();
                std::string key;
                if (!string(key)) return false;
                skip_ws();
                if (p_ >= end_ || *p_ != ':') return fail("expected ':'");
                ++p_;
                out.obj.emplace_back(std::move(key), Value());
                if (!value(out.obj.back().second, depth + 1)) return false;
                skip_ws();
                if (p_ < end_ && *p_ == ',') {
                    ++p_;
                    continue;
                }
                if (p_ < end_ && *p_ == '}') {
                    ++p_;
                    return true;
                }
                return fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            ++p_;
            out.kind = Value::Array;
            skip_ws();
            if (p_ < end_ && *p_ == ']') {
                ++p_;
                return true;
            }
            for (;;) {
                out.arr.emplace_back();
                if (!value(out.arr.back(), depth + 1)) return false;
                skip_ws();
                if (p_ < end_ && *p_ == ',') {
                    ++p_;
                    continue;
                }
                if (p_ < end_ && *p_ == ']') {
                    ++p_;
                    return true;
                }
                return fail("expected ',' or ']'");
            }
        }
        if (c == '"') {
            out.kind = Value::String;
            return string(out.str);
        }
        if (c == 't') {
            out.kind = Value::Bool;
            out.b = true;
            return literal("true");
        }
        if (c == 'f') {
            out.kind = Value::Bool;
            out.b = false;
            return literal("false");
        }
        if (c == 'n') {
            out.kind = Value::Null;
            return literal("null");
        }
        if (c == '-' || (c >= '0' && c <= '9')) {
            const char *q = p_;
            if (*q == '-') ++q;
            while (q < end_ && ((*q >= '0' && *q <= '9') || *q == '.' || *q == 'e' || *q == 'E' || *q == '+' || *q == '-')) ++q;
            // from_chars takes no leading '+', like JSON; it must consume the whole token
            const auto r = std::from_chars(p_, q, out.num);
            if (r.ec != std::errc() || r.ptr != q) return fail("bad number");
            out.kind = Value::Number;
            p_ = q;
            return true;
        }
        return fail("unexpected character in JSON");
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
