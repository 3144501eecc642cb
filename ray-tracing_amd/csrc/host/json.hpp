// Minimal JSON DOM parser (RFC 8259 subset sufficient for the scene schema): objects keep key
// order, numbers are doubles, strings support the standard escapes.  Errors throw
// std::runtime_error with a byte offset.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace v4ray {
namespace json {

struct Value;
using ValuePtr = std::shared_ptr<Value>;

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;

    bool is_null() const { return kind == Null; }
    bool is_string() const { return kind == String; }
    bool is_object() const { return kind == Object; }
    bool is_array() const { return kind == Array; }
    bool is_number() const { return kind == Number; }
    bool is_bool() const { return kind == Bool; }

    const Value *find(const std::string &key) const {
        if (kind != Object) return nullptr;
        for (auto &kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const Value &at(const std::string &key) const {
        const Value *v = find(key);
        if (!v) throw std::runtime_error("missing field `" + key + "`");
        return *v;
    }
    double as_number(const char *what) const {
        if (kind != Number) throw std::runtime_error(std::string("expected a number for ") + what);
        return num;
    }
    const std::string &as_string(const char *what) const {
        if (kind != String) throw std::runtime_error(std::string("expected a string for ") + what);
        return str;
    }
    bool as_bool(const char *what) const {
        if (kind != Bool) throw std::runtime_error(std::string("expected a boolean for ") + what);
        return b;
    }
};

class Parser {
  public:
    explicit Parser(const std::string &text) : s_(text) {}
    Value parse() {
        Value v = value();
        ws();
        if (i_ != s_.size()) fail("trailing characters");
        return v;
    }

  private:
    const std::string &s_;
    size_t i_ = 0;
    [[noreturn]] void fail(const char *msg) const {
        throw std::runtime_error(std::string("JSON: ") + msg + " at byte " + std::to_string(i_));
    }
    void ws() {
        while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\t' || s_[i_] == '\n' || s_[i_] == '\r')) ++i_;
    }
    bool eat(char c) {
        ws();
        if (i_ < s_.size() && s_[i_] == c) {
            ++i_;
            return true;
        }
        return false;
    }
    void expect(char c) {
        if (!eat(c)) fail("unexpected character");
    }
    Value value() {
        ws();
        if (i_ >= s_.size()) fail("unexpected end");
        char c = s_[i_];
        if (c == '{') return object();
        if (c == '[') return array();
        if (c == '"') {
            Value v;
            v.kind = Value::String;
            v.str = string();
            return v;
        }
        if (s_.compare(i_, 4, "true") == 0) {
            i_ += 4;
            Value v;
            v.kind = Value::Bool;
            v.b = true;
            return v;
        }
        if (s_.compare(i_, 5, "false") == 0) {
            i_ += 5;
            Value v;
            v.kind = Value::Bool;
            v.b = false;
            return v;
        }
        if (s_.compare(i_, 4, "null") == 0) {
            i_ += 4;
            return Value();
        }
        return number();
    }
    Value number() {
        const char *begin = s_.c_str() + i_;
        char *end = nullptr;
        double d = std::strtod(begin, &end);
        if (end == begin) fail("invalid number");
        i_ += (size_t)(end - begin);
        Value v;
        v.kind = Value::Number;
        v.num = d;
        return v;
    }
    static void append_utf8(std::string &out, uint32_t cp) {
        if (cp < 0x80) out.push_back((char)cp);
        else if (cp < 0x800) {
            out.push_back((char)(0xC0 | (cp >> 6)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
        } else if (cp < 0x10000) {
            out.push_back((char)(0xE0 | (cp >> 12)));
            out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
        } else {
            out.push_back((char)(0xF0 | (cp >> 18)));
            out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
            out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
        }
    }
    uint32_t hex4() {
        if (i_ + 4 > s_.size()) fail("bad \\u escape");
        uint32_t v = 0;
        for (int k = 0; k < 4; ++k) {
            char c = s_[i_++];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else fail("bad \\u escape");
        }
        return v;
    }
    std::string string() {
        expect('"');
        std::string out;
        while (true) {
            if (i_ >= s_.size()) fail("unterminated string");
            char c = s_[i_++];
            if (c == '"') break;
            if (c != '\\') {
                out.push_back(c);
                continue;
            }
            if (i_ >= s_.size()) fail("unterminated escape");
            char e = s_[i_++];
            switch (e) {
                case '"': out.push_back('"'); break;
                case '\\': out.push_back('\\'); break;
                case '/': out.push_back('/'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'n': out.push_back('\n'); break;
                case 'r': out.push_back('\r'); break;
                case 't': out.push_back('\t'); break;
                case 'u': {
                    uint32_t cp = hex4();
                    if (cp >= 0xD800 && cp < 0xDC00 && i_ + 1 < s_.size() && s_[i_] == '\\' && s_[i_ + 1] == 'u') {
                        i_ += 2;
                        uint32_t lo = hex4();
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    append_utf8(out, cp);
                    break;
                }
                default: fail("bad escape");
            }
        }
        return out;
    }
    Value array() {
        expect('[');
        Value v;
        v.kind = Value::Array;
        if (eat(']')) return v;
        while (true) {
            v.arr.push_back(value());
            if (eat(']')) break;
            expect(',');
        }
        return v;
    }
    Value object() {
        expect('{');
        Value v;
        v.kind = Value::Object;
        if (eat('}')) return v;
        while (true) {
            ws();
            std::string key = string();
            expect(':');
            v.obj.emplace_back(std::move(key), value());
            if (eat('}')) break;
            expect(',');
        }
        return v;
    }
};

inline Value parse(const std::string &text) { return Parser(text).parse(); }

}  // namespace json
}  // namespace v4ray
