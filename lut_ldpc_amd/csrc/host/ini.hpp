// ini.hpp -- small INI reader in place of boost::property_tree::ini_parser
// (used by the reference at src/LDPC_BER_Sim.cpp:50,380, src/LDPC_DE.cpp:1149, prog/ber_sim.cpp:128).
// Sections, `key = value`, whole-line comments starting with ';' or '#'; values are trimmed.
#pragma once
#include <fstream>
#include <map>
#include <optional>
#include <sstream>
#include <stdexcept>
#include <string>

namespace lut_ldpc {

class Ini {
public:
    explicit Ini(const std::string &path) {
        std::ifstream f(path);
        if (!f) throw std::runtime_error("cannot open INI file " + path);
        std::string line, section;
        while (std::getline(f, line)) {
            const std::string t = trim(line);
            if (t.empty() || t[0] == ';' || t[0] == '#') continue;
            if (t[0] == '[') {
                const size_t e = t.find(']');
                if (e == std::string::npos) throw std::runtime_error("unterminated section header in " + path);
                section = trim(t.substr(1, e - 1));
                sections_[section];
                continue;
            }
            const size_t eq = t.find('=');
            if (eq == std::string::npos) throw std::runtime_error("line without '=' in " + path + ": " + t);
            sections_[section][trim(t.substr(0, eq))] = trim(t.substr(eq + 1));
        }
    }
    bool has_section(const std::string &s) const { return sections_.count(s) != 0; }
    std::optional<std::string> get_optional(const std::string &section, const std::string &key) const {
        auto s = sections_.find(section);
        if (s == sections_.end()) return std::nullopt;
        auto k = s->second.find(key);
        if (k == s->second.end()) return std::nullopt;
        return k->second;
    }
    // "Section.key" with a default, like ptree::get(path, default)
    std::string get(const std::string &path, const std::string &dflt) const {
        auto v = lookup(path);
        return v ? *v : dflt;
    }
    std::string get(const std::string &path, const char *dflt) const { return get(path, std::string(dflt)); }
    double get(const std::string &path, double dflt) const { auto v = lookup(path); return v ? std::stod(*v) : dflt; }
    int get(const std::string &path, int dflt) const { auto v = lookup(path); return v ? (int)std::stod(*v) : dflt; }
    bool get(const std::string &path, bool dflt) const {
        auto v = lookup(path);
        if (!v) return dflt;
        if (*v == "true" || *v == "1") return true;
        if (*v == "false" || *v == "0") return false;
        throw std::runtime_error("not a boolean: " + path + " = " + *v);
    }
    std::string require(const std::string &path) const {
        auto v = lookup(path);
        if (!v) throw std::runtime_error("missing key " + path);
        return *v;
    }
    std::optional<std::string> lookup(const std::string &path) const {
        const size_t dot = path.find('.');
        if (dot == std::string::npos) return std::nullopt;
        return get_optional(path.substr(0, dot), path.substr(dot + 1));
    }
    static std::string trim(const std::string &s) {
        size_t b = s.find_first_not_of(" \t\r\n"), e = s.find_last_not_of(" \t\r\n");
        return b == std::string::npos ? std::string() : s.substr(b, e - b + 1);
    }
private:
    std::map<std::string, std::map<std::string, std::string>> sections_;
};

}  // namespace lut_ldpc
