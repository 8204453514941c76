// mini_json.hpp — a small JSON reader for the host side (scene files, benchmark task files).  Objects keep their key
// order; `//` and `/* */` comments are skipped, as the reference tells its parser to (json_loader.cpp:1103,
// Application/headless.cpp:73).  Errors are thrown as std::runtime_error("json: ... at byte N").
#pragma once

#include <cctype>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace RayZath::Hip::IO {

struct Json {
    enum Kind { Null, Bool, Int, Float, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Json> items;
    std::vector<std::pair<std::string, Json>> members;
    bool is_number() const { return kind == Int || kind == Float; }
    bool is_string() const { return kind == String; }
    bool is_object() const { return kind == Object; }
    bool is_array() const { return kind == Array; }
    bool is_bool() const { return kind == Bool; }
    const Json* find(const std::string& key) const {
        for (const auto& m : members)
            if (m.first == key) return &m.second;
        return nullptr;
    }
    bool contains(const std::string& key) const { return find(key) != nullptr; }
};
struct JsonParser {
    const std::string& s;
    size_t i = 0;
    int depth = 0;                         // nesting of the value being parsed
    static constexpr int kMaxDepth = 128;  // scene files nest 6 deep; an untrusted file must not recurse the host stack away
    explicit JsonParser(const std::string& text) : s(text) {}
    [[noreturn]] void error(const std::string& why) { throw std::runtime_error("json: " + why + " at byte " + std::to_string(i)); }
    void skip() {
        while (i < s.size()) {
            if (std::isspace(static_cast<unsigned char>(s[i]))) {
                ++i;
            } else if (s.compare(i, 2, "//") == 0) {
                while (i < s.size() && s[i] != '\n') ++i;
            } else if (s.compare(i, 2, "/*") == 0) {
                const size_t e = s.find("*/", i + 2);
                if (e == std::string::npos) error("unterminated comment");
                i = e + 2;
            } else {
                break;
            }
        }
    }
    Json parse() {
        Json v = value();
        skip();
        if (i != s.size()) error("trailing characters");
        return v;
    }
    Json value() {
        skip();
        if (i >= s.size()) error("unexpected end");
        Json v;
        const char c = s[i];
        struct Nest {
            int& d;
            explicit Nest(int& depth) : d(depth) { ++d; }
            ~Nest() { --d; }
        } nest(depth);
        if (depth > kMaxDepth) error("nested deeper than " + std::to_string(kMaxDepth) + " levels");
        if (c == '{') {
            v.kind = Json::Object;
            ++i;
            skip();
            if (i < s.size() && s[i] == '}') return ++i, v;
            while (true) {
                skip();
                if (i >= s.size() || s[i] != '"') error("expected a key");
                std::string key = string();
                skip();
                if (i >= s.size() || s[i] != ':') error("expected ':'");
                ++i;
                v.members.emplace_back(std::move(key), value());
                skip();
                if (i < s.size() && s[i] == ',') {
                    ++i;
                    continue;
                }
                if (i < s.size() && s[i] == '}') return ++i, v;
                error("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.kind = Json::Array;
            ++i;
            skip();
            if (i < s.size() && s[i] == ']') return ++i, v;
            while (true) {
                v.items.push_back(value());
                skip();
                if (i < s.size() && s[i] == ',') {
                    ++i;
                    continue;
                }
                if (i < s.size() && s[i] == ']') return ++i, v;
                error("expected ',' or ']'");
            }
        }
        if (c == '"') {
            v.kind = Json::String;
            v.str = string();
            return v;
        }
        if (s.compare(i, 4, "true") == 0) return i += 4, v.kind = Json::Bool, v.b = true, v;
        if (s.compare(i, 5, "false") == 0) return i += 5, v.kind = Json::Bool, v.b = false, v;
        if (s.compare(i, 4, "null") == 0) return i += 4, v;
        if (c == '-' || (c >= '0' && c <= '9')) {
            const size_t b = i;
            bool is_float = false;
            if (s[i] == '-') ++i;
            while (i < s.size() && (std::isdigit(static_cast<unsigned char>(s[i])) || s[i] == '.' || s[i] == 'e' || s[i] == 'E' || s[i] == '+' || s[i] == '-')) {
                if (s[i] == '.' || s[i] == 'e' || s[i] == 'E') is_float = true;
                ++i;
            }
            v.kind = is_float ? Json::Float : Json::Int;
            v.str = s.substr(b, i - b);
            v.num = std::strtod(v.str.c_str(), nullptr);
            return v;
        }
        error("unexpected character");
    }
    std::string string() {
        std::string out;
        ++i;  // opening quote
        while (i < s.size() && s[i] != '"') {
            if (s[i] == '\\') {
                if (++i >= s.size()) error("bad escape");
                switch (s[i]) {
                    case 'n': out.push_back('\n'); break;
                    case 't': out.push_back('\t'); break;
                    case 'r': out.push_back('\r'); break;
                    case 'b': out.push_back('\b'); break;
                    case 'f': out.push_back('\f'); break;
                    case 'u': {
                        if (i + 4 >= s.size()) error("bad \\u escape");
                        const unsigned cp = unsigned(std::strtoul(s.substr(i + 1, 4).c_str(), nullptr, 16));
                        i += 4;
                        if (cp < 0x80) out.push_back(char(cp));
                        else if (cp < 0x800) out.push_back(char(0xC0 | (cp >> 6))), out.push_back(char(0x80 | (cp & 0x3F)));
                        else out.push_back(char(0xE0 | (cp >> 12))), out.push_back(char(0x80 | ((cp >> 6) & 0x3F))), out.push_back(char(0x80 | (cp & 0x3F)));
                        break;
                    }
                    default: out.push_back(s[i]);
                }
                ++i;
            } else {
                out.push_back(s[i++]);
            }
        }
        if (i >= s.size()) error("unterminated string");
        ++i;
        return out;
    }
};
inline Json parseJson(const std::string& text) { return JsonParser(text).parse(); }

}  // namespace RayZath::Hip::IO
