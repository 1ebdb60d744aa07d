// Minimal TOML reader for the simulator's configuration files: top-level and
// [table] key/value pairs with strings, booleans, integers, floats and flat
// arrays of numbers; '#' comments.  Unknown keys are kept and ignored by the
// schema layer, as serde does without deny_unknown_fields
// (common/src/parameters.rs:10-55).
#pragma once
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace msmhost {

struct TomlValue {
    enum Kind { STRING, BOOL, INT, FLOAT, ARRAY } kind = STRING;
    std::string s;
    bool b = false;
    long long i = 0;
    double f = 0.0;
    std::vector<double> arr;
    bool is_number() const { return kind == INT || kind == FLOAT; }
    double as_double() const { return kind == INT ? (double)i : f; }
};

using TomlTable = std::map<std::string, TomlValue>;
struct TomlDoc {
    TomlTable root;
    std::map<std::string, TomlTable> tables;
    const TomlTable* table(const std::string& name) const {
        auto it = tables.find(name);
        return it == tables.end() ? nullptr : &it->second;
    }
};

inline std::string toml_trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) ++a;
    while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

inline std::string toml_strip_comment(const std::string& line) {
    bool in_str = false;
    for (size_t i = 0; i < line.size(); ++i) {
        const char c = line[i];
        if (c == '"' && (i == 0 || line[i - 1] != '\\')) in_str = !in_str;
        if (c == '#' && !in_str) return line.substr(0, i);
    }
    return line;
}

inline bool toml_parse_number(std::string t, TomlValue& v) {
    std::string u;
    for (char c : t) if (c != '_') u.push_back(c);
    if (u.empty()) return false;
    if (u == "inf" || u == "+inf") { v.kind = TomlValue::FLOAT; v.f = HUGE_VAL; return true; }
    if (u == "-inf") { v.kind = TomlValue::FLOAT; v.f = -HUGE_VAL; return true; }
    if (u == "nan" || u == "+nan" || u == "-nan") { v.kind = TomlValue::FLOAT; v.f = std::strtod("nan", nullptr); return true; }
    const bool is_float = u.find_first_of(".eE") != std::string::npos;
    char* end = nullptr;
    if (is_float) {
        v.f = std::strtod(u.c_str(), &end);
        v.kind = TomlValue::FLOAT;
    } else {
        v.i = std::strtoll(u.c_str(), &end, 10);
        v.kind = TomlValue::INT;
    }
    return end && *end == '\0';
}

inline TomlValue toml_parse_value(const std::string& raw, int lineno) {
    const std::string t = toml_trim(raw);
    TomlValue v;
    auto fail = [&](const std::string& why) -> TomlValue {
        throw std::runtime_error("TomlParseError: line " + std::to_string(lineno) + ": " + why + ": `" + t + "`");
    };
    if (t.empty()) return fail("missing value");
    if (t[0] == '"') {
        if (t.size() < 2 || t.back() != '"') return fail("unterminated string");
        v.kind = TomlValue::STRING;
        for (size_t i = 1; i + 1 < t.size(); ++i) {
            if (t[i] == '\\' && i + 2 < t.size()) {
                const char e = t[++i];
                v.s.push_back(e == 'n' ? '\n' : e == 't' ? '\t' : e);
            } else {
                v.s.push_back(t[i]);
            }
        }
        return v;
    }
    if (t == "true" || t == "false") { v.kind = TomlValue::BOOL; v.b = (t == "true"); return v; }
    if (t[0] == '[') {
        if (t.back() != ']') return fail("unterminated array");
        v.kind = TomlValue::ARRAY;
        std::stringstream ss(t.substr(1, t.size() - 2));
        std::string item;
        while (std::getline(ss, item, ',')) {
            item = toml_trim(item);
            if (item.empty()) continue;
            TomlValue e;
            if (!toml_parse_number(item, e)) return fail("only arrays of numbers are supported");
            v.arr.push_back(e.as_double());
        }
        return v;
    }
    if (!toml_parse_number(t, v)) return fail("invalid value");
    return v;
}

inline TomlDoc toml_parse(std::istream& in) {
    TomlDoc doc;
    TomlTable* cur = &doc.root;
    std::string line;
    int lineno = 0;
    while (std::getline(in, line)) {
        ++lineno;
        line = toml_trim(toml_strip_comment(line));
        if (line.empty()) continue;
        if (line[0] == '[') {
            if (line.back() != ']') throw std::runtime_error("TomlParseError: line " + std::to_string(lineno) + ": bad table header");
            cur = &doc.tables[toml_trim(line.substr(1, line.size() - 2))];
            continue;
        }
        const size_t eq = line.find('=');
        if (eq == std::string::npos) throw std::runtime_error("TomlParseError: line " + std::to_string(lineno) + ": expected key = value");
        const std::string key = toml_trim(line.substr(0, eq));
        if (key.empty()) throw std::runtime_error("TomlParseError: line " + std::to_string(lineno) + ": empty key");
        (*cur)[key] = toml_parse_value(line.substr(eq + 1), lineno);
    }
    return doc;
}

}  // namespace msmhost
