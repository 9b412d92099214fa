// itfile.hpp -- reader/writer for the IT++ ".it" container (file version 3), restricted to the
// value types the reference stores: results files (src/LDPC_BER_Sim.cpp:342-362) and
// lut_codec.it (src/LDPC_Code_LUT.cpp:568-697).  Wire format per scripts/itload.m:48-63 and
// scripts/itsave.m:55,85-99: magic "IT++" + version byte 3, then blocks of
//   u64 header_bytes, u64 data_bytes, u64 block_bytes, name\0, type\0, description\0, payload
// little-endian; vectors are u64 length + elements; an Array<ivec> ("ivecArray") is u64 count
// followed by that many ivec payloads.  MATLAB's aggregate_results.m / analyze_results.m read
// files written here unchanged.
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace lut_ldpc {

class it_file_writer {
public:
    explicit it_file_writer(const std::string &path, bool truncate = true)
        : f_(path, std::ios::binary | (truncate ? std::ios::trunc : std::ios::app)) {
        if (!f_) throw std::runtime_error("it_file: cannot open " + path + " for writing");
        if (truncate) { f_.write("IT++", 4); f_.put((char)3); }
    }
    void write(const std::string &name, bool v) { std::string d(1, (char)(v ? 1 : 0)); block(name, "bin", d); }
    void write(const std::string &name, int v) { block(name, "int32", raw(&v, 4)); }
    void write(const std::string &name, double v) { block(name, "float64", raw(&v, 8)); }
    void write(const std::string &name, const std::string &s) { block(name, "string", len(s.size()) + s); }
    void write(const std::string &name, const std::vector<int> &v) { block(name, "ivec", len(v.size()) + raw(v.data(), 4 * v.size())); }
    void write(const std::string &name, const std::vector<double> &v) { block(name, "dvec", len(v.size()) + raw(v.data(), 8 * v.size())); }
    void write(const std::string &name, const std::vector<unsigned char> &v) { block(name, "bvec", len(v.size()) + raw(v.data(), v.size())); }
    void write(const std::string &name, const std::vector<std::vector<int>> &a) {
        std::string d = len(a.size());
        for (auto &v : a) d += len(v.size()) + raw(v.data(), 4 * v.size());
        block(name, "ivecArray", d);
    }
    void close() { f_.close(); }
private:
    static std::string raw(const void *p, size_t n) { return std::string((const char *)p, n); }
    static std::string len(uint64_t n) { return raw(&n, 8); }
    void block(const std::string &name, const std::string &type, const std::string &data) {
        const uint64_t hdr = 24 + name.size() + 1 + type.size() + 1 + 1, dat = data.size(), tot = hdr + dat;
        f_.write((const char *)&hdr, 8); f_.write((const char *)&dat, 8); f_.write((const char *)&tot, 8);
        f_.write(name.c_str(), (std::streamsize)name.size() + 1);
        f_.write(type.c_str(), (std::streamsize)type.size() + 1);
        f_.put('\0');
        f_.write(data.data(), (std::streamsize)data.size());
        if (!f_) throw std::runtime_error("it_file: write failed");
    }
    std::ofstream f_;
};

class it_file_reader {
public:
    explicit it_file_reader(const std::string &path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("it_file: cannot open " + path);
        std::string all((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        if (all.size() < 5 || all.compare(0, 4, "IT++") != 0 || all[4] != 3) throw std::runtime_error("it_file: " + path + " is not an IT++ v3 file");
        size_t pos = 5;
        while (pos + 24 <= all.size()) {
            uint64_t hdr, dat, tot;
            std::memcpy(&hdr, &all[pos], 8); std::memcpy(&dat, &all[pos + 8], 8); std::memcpy(&tot, &all[pos + 16], 8);
            if (tot < hdr || pos + tot > all.size() || hdr < 27) throw std::runtime_error("it_file: corrupt block in " + path);
            const char *p = &all[pos + 24];
            std::string name(p), type(p + name.size() + 1);
            if (!type.empty()) vars_[name] = {type, all.substr(pos + hdr, dat)};
            pos += tot;
        }
    }
    bool has(const std::string &name) const { return vars_.count(name) != 0; }
    bool get_bool(const std::string &n) const { return payload(n, "bin").at(0) != 0; }
    int get_int(const std::string &n) const { int v; std::memcpy(&v, payload(n, "int32").data(), 4); return v; }
    double get_double(const std::string &n) const { double v; std::memcpy(&v, payload(n, "float64").data(), 8); return v; }
    std::string get_string(const std::string &n) const { const std::string &d = payload(n, "string"); return d.substr(8, count(d, 0, 1)); }
    std::vector<int> get_ivec(const std::string &n) const { const std::string &d = payload(n, "ivec"); std::vector<int> v(count(d, 0, 4)); if (!v.empty()) std::memcpy(v.data(), &d[8], 4 * v.size()); return v; }
    std::vector<double> get_dvec(const std::string &n) const { const std::string &d = payload(n, "dvec"); std::vector<double> v(count(d, 0, 8)); if (!v.empty()) std::memcpy(v.data(), &d[8], 8 * v.size()); return v; }
    std::vector<unsigned char> get_bvec(const std::string &n) const { const std::string &d = payload(n, "bvec"); std::vector<unsigned char> v(count(d, 0, 1)); if (!v.empty()) std::memcpy(v.data(), &d[8], v.size()); return v; }
    std::vector<std::vector<int>> get_ivec_array(const std::string &n) const {
        const std::string &d = payload(n, "ivecArray");
        if (d.size() < 8) throw std::runtime_error("it_file: truncated " + n);
        uint64_t na; std::memcpy(&na, d.data(), 8);
        std::vector<std::vector<int>> a;
        size_t pos = 8;
        for (uint64_t i = 0; i < na; i++) {
            std::vector<int> v(count(d, pos, 4));
            if (!v.empty()) std::memcpy(v.data(), &d[pos + 8], 4 * v.size());
            pos += 8 + 4 * v.size();
            a.push_back(std::move(v));
        }
        return a;
    }
private:
    struct Var { std::string type, data; };
    const std::string &payload(const std::string &n, const char *type) const {
        auto it = vars_.find(n);
        if (it == vars_.end()) throw std::runtime_error("it_file: no variable named " + n);
        if (it->second.type != type) throw std::runtime_error("it_file: " + n + " has type " + it->second.type + ", wanted " + type);
        return it->second.data;
    }
    static size_t count(const std::string &d, size_t pos, size_t elem) {
        if (pos + 8 > d.size()) throw std::runtime_error("it_file: truncated vector");
        uint64_t n; std::memcpy(&n, &d[pos], 8);
        if (pos + 8 + n * elem > d.size()) throw std::runtime_error("it_file: truncated vector");
        return (size_t)n;
    }
    std::map<std::string, Var> vars_;
};

}  // namespace lut_ldpc
