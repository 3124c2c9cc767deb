// json.h — minimal JSON value + recursive-descent parser for the scene loader (no dependencies).
#pragma once

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace smr_host {

struct Json {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;  // insertion order kept (deny_unknown_fields reporting)

    const Json *get(const std::string &key) const {
        if (kind != Object) return nullptr;
        for (auto &kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool is_null() const { return kind == Null; }

    static Json number(double v) { Json j; j.kind = Number; j.num = v; return j; }
    static Json string(const std::string &v) { Json j; j.kind = String; j.str = v; return j; }
    static Json boolean(bool v) { Json j; j.kind = Bool; j.b = v; return j; }
    static Json array(std::vector<Json> v = {}) { Json j; j.kind = Array; j.arr = std::move(v); return j; }
    static Json object() { Json j; j.kind = Object; return j; }
    Json &set(const std::string &key, Json v) { obj.emplace_back(key, std::move(v)); return *this; }

    void dump(std::string &out) const {
        switch (kind) {
        case Null: out += "null"; break;
        case Bool: out += b ? "true" : "false"; break;
        case Number: {
            if (!std::isfinite(num)) { out += "null"; break; }
            char buf[40];
            snprintf(buf, sizeof(buf), "%.17g", num);
            out += buf;
            break;
        }
        case String: dump_string(str, out); break;
        case Array:
            out += '[';
            for (size_t i = 0; i < arr.size(); i++) { if (i) out += ','; arr[i].dump(out); }
            out += ']';
            break;
        case Object:
            out += '{';
            for (size_t i = 0; i < obj.size(); i++) {
                if (i) out += ',';
                dump_string(obj[i].first, out);
                out += ':';
                obj[i].second.dump(out);
            }
            out += '}';
            break;
        }
    }
    static void dump_string(const std::string &s, std::string &out) {
        out += '"';
        for (unsigned char c : s) {
            if (c == '"') out += "\\\"";
            else if (c == '\\') out += "\\\\";
            else if (c == '\n') out += "\\n";
            else if (c == '\r') out += "\\r";
            else if (c == '\t') out += "\\t";
            else if (c < 0x20) { char buf[8]; snprintf(buf, sizeof(buf), "\\u%04x", c); out += buf; }
            else out += (char)c;
        }
        out += '"';
    }
};

class JsonParser {
  public:
    explicit JsonParser(const std::string &s) : s_(s) {}
    bool parse(Json &out, std::string &err) {
        skip();
        if (!value(out, err, 0)) return false;
        skip();
        if (p_ != s_.size()) { err = "trailing characters after JSON value"; return false; }
        return true;
    }

  private:
    const std::string &s_;
    size_t p_ = 0;

    void skip() {
        while (p_ < s_.size() && (s_[p_] == ' ' || s_[p_] == '\n' || s_[p_] == '\t' || s_[p_] == '\r')) p_++;
    }
    bool fail(std::string &err, const char *msg) {
        err = std::string("JSON parse error at byte ") + std::to_string(p_) + ": " + msg;
        return false;
    }
    bool value(Json &out, std::string &err, int depth) {
        if (depth > 256) return fail(err, "nesting too deep");
        skip();
        if (p_ >= s_.size()) return fail(err, "unexpected end");
        char c = s_[p_];
        if (c == '{') return object(out, err, depth);
        if (c == '[') return array(out, err, depth);
        if (c == '"') { out.kind = Json::String; return string(out.str, err); }
        if (s_.compare(p_, 4, "true") == 0) { out.kind = Json::Bool; out.b = true; p_ += 4; return true; }
        if (s_.compare(p_, 5, "false") == 0) { out.kind = Json::Bool; out.b = false; p_ += 5; return true; }
        if (s_.compare(p_, 4, "null") == 0) { out.kind = Json::Null; p_ += 4; return true; }
        return number(out, err);
    }
    bool number(Json &out, std::string &err) {
        const char *start = s_.c_str() + p_;
        char *end = nullptr;
        double v = std::strtod(start, &end);
        if (end == start) return fail(err, "invalid value");
        p_ += (size_t)(end - start);
        out.kind = Json::Number;
        out.num = v;
        return true;
    }
    bool string(std::string &out, std::string &err) {
        p_++;  // opening quote
        out.clear();
        while (p_ < s_.size()) {
            char c = s_[p_++];
            if (c == '"') return true;
            if (c == '\\') {
                if (p_ >= s_.size()) break;
                char e = s_[p_++];
                switch (e) {
                case '"': out += '"'; break;
                case '\\': out += '\\'; break;
                case '/': out += '/'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'n': out += '\n'; break;
                case 'r': out += '\r'; break;
                case 't': out += '\t'; break;
                case 'u': {
                    if (p_ + 4 > s_.size()) return fail(err, "bad \\u escape");
                    unsigned cp = (unsigned)std::strtoul(s_.substr(p_, 4).c_str(), nullptr, 16);
                    p_ += 4;
                    if (cp < 0x80) out += (char)cp;
                    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                    else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                    break;
                }
                default: return fail(err, "bad escape");
                }
            } else {
                out += c;
            }
        }
        return fail(err, "unterminated string");
    }
    bool array(Json &out, std::string &err, int depth) {
        out.kind = Json::Array;
        p_++;
        skip();
        if (p_ < s_.size() && s_[p_] == ']') { p_++; return true; }
        while (true) {
            Json v;
            if (!value(v, err, depth + 1)) return false;
            out.arr.push_back(std::move(v));
            skip();
            if (p_ >= s_.size()) return fail(err, "unterminated array");
            if (s_[p_] == ',') { p_++; continue; }
            if (s_[p_] == ']') { p_++; return true; }
            return fail(err, "expected ',' or ']'");
        }
    }
    bool object(Json &out, std::string &err, int depth) {
        out.kind = Json::Object;
        p_++;
        skip();
        if (p_ < s_.size() && s_[p_] == '}') { p_++; return true; }
        while (true) {
            skip();
            if (p_ >= s_.size() || s_[p_] != '"') return fail(err, "expected object key");
            std::string key;
            if (!string(key, err)) return false;
            skip();
            if (p_ >= s_.size() || s_[p_] != ':') return fail(err, "expected ':'");
            p_++;
            Json v;
            if (!value(v, err, depth + 1)) return false;
            out.obj.emplace_back(std::move(key), std::move(v));
            skip();
            if (p_ >= s_.size()) return fail(err, "unterminated object");
            if (s_[p_] == ',') { p_++; continue; }
            if (s_[p_] == '}') { p_++; return true; }
            return fail(err, "expected ',' or '}'");
        }
    }
};

}  // namespace smr_host
