// NPY / NPZ codec for the simulator's file surface:
//   dumps   psi_{k:05}_real / _imag : NPY v1.0, C order, shape (N,N,N,1)   simulator/src/utils/io.rs:34-108
//   ICs     .npz with members real.npy / imag.npy                           simulator/src/ics.rs:680-726
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace msmhost {

template <typename T> struct NpyDescr;
template <> struct NpyDescr<double> { static constexpr const char* value = "<f8"; };
template <> struct NpyDescr<float> { static constexpr const char* value = "<f4"; };

template <typename T>
void write_npy(const std::string& path, const T* data, const std::vector<size_t>& shape) {
    std::string dict = std::string("{'descr': '") + NpyDescr<T>::value + "', 'fortran_order': False, 'shape': (";
    size_t count = 1;
    for (size_t i = 0; i < shape.size(); ++i) {
        dict += std::to_string(shape[i]) + (shape.size() == 1 || i + 1 < shape.size() ? ", " : "");
        count *= shape[i];
    }
    dict += "), }";
    // magic(6) + version(2) + header_len(2) + dict + padding + '\n' must be a multiple of 64
    const size_t unpadded = 10 + dict.size() + 1;
    dict.append((64 - unpadded % 64) % 64, ' ');
    dict.push_back('\n');
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("Failed to write to disk: " + path);
    const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
    const uint16_t hl = (uint16_t)dict.size();
    bool ok = std::fwrite(magic, 1, 8, f) == 8 && std::fwrite(&hl, 2, 1, f) == 1 &&
              std::fwrite(dict.data(), 1, dict.size(), f) == dict.size() &&
              std::fwrite(data, sizeof(T), count, f) == count;
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) throw std::runtime_error("Failed to write to disk: " + path);
}

struct NpyArray {
    std::string descr;
    bool fortran_order = false;
    std::vector<size_t> shape;
    std::vector<unsigned char> bytes;   // raw payload
    size_t count() const { size_t c = 1; for (size_t s : shape) c *= s; return c; }
};

inline NpyArray parse_npy(const unsigned char* p, size_t n, const std::string& what) {
    if (n < 10 || p[0] != 0x93 || std::memcmp(p + 1, "NUMPY", 5) != 0) throw std::runtime_error(what + ": not an NPY stream");
    const int major = p[6];
    size_t hl, off;
    if (major == 1) { hl = p[8] | (p[9] << 8); off = 10; }
    else { if (n < 12) throw std::runtime_error(what + ": truncated"); hl = p[8] | (p[9] << 8) | (p[10] << 16) | ((size_t)p[11] << 24); off = 12; }
    if (off + hl > n) throw std::runtime_error(what + ": truncated header");
    const std::string h((const char*)p + off, hl);
    NpyArray a;
    auto field = [&](const std::string& key) -> size_t {
        const size_t k = h.find("'" + key + "'");
        if (k == std::string::npos) throw std::runtime_error(what + ": header lacks " + key);
        return h.find(':', k) + 1;
    };
    {
        size_t s = h.find('\'', field("descr"));
        size_t e = h.find('\'', s + 1);
        a.descr = h.substr(s + 1, e - s - 1);
    }
    a.fortran_order = h.compare(h.find_first_not_of(' ', field("fortran_order")), 4, "True") == 0;
    {
        size_t s = h.find('(', field("shape")), e = h.find(')', s);
        std::string body = h.substr(s + 1, e - s - 1);
        size_t pos = 0;
        while (pos < body.size()) {
            while (pos < body.size() && !std::isdigit((unsigned char)body[pos])) ++pos;
            if (pos >= body.size()) break;
            size_t end = pos;
            while (end < body.size() && std::isdigit((unsigned char)body[end])) ++end;
            a.shape.push_back((size_t)std::stoull(body.substr(pos, end - pos)));
            pos = end;
        }
    }
    a.bytes.assign(p + off + hl, p + n);
    return a;
}

// --- zip container (stored or deflated members; zip64 sizes as numpy writes them) ---------------
namespace zipdetail {
inline uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint64_t rd64(const unsigned char* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }
}  // namespace zipdetail

inline std::vector<unsigned char> read_file(const std::string& path, const std::string& missing_msg) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error(missing_msg + ": " + path);
    return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// Returns the member's uncompressed bytes; throws if absent.
inline std::vector<unsigned char> zip_member(const std::vector<unsigned char>& z, const std::string& name) {
    using namespace zipdetail;
    if (z.size() < 22) throw std::runtime_error("failed to read file as npz");
    size_t eocd = std::string::npos;
    for (size_t i = z.size() - 22 + 1; i-- > 0;) {
        if (rd32(&z[i]) == 0x06054b50) { eocd = i; break; }
        if (z.size() - i > 22 + 65535) break;
    }
    if (eocd == std::string::npos) throw std::runtime_error("failed to read file as npz");
    uint64_t n_entries = rd16(&z[eocd + 10]), cd_off = rd32(&z[eocd + 16]);
    if (cd_off == 0xFFFFFFFFu || n_entries == 0xFFFF) {   // zip64 end-of-central-directory
        if (eocd < 20 || rd32(&z[eocd - 20]) != 0x07064b50) throw std::runtime_error("failed to read file as npz");
        const uint64_t e64 = rd64(&z[eocd - 20 + 8]);
        if (e64 + 56 > z.size() || rd32(&z[e64]) != 0x06064b50) throw std::runtime_error("failed to read file as npz");
        n_entries = rd64(&z[e64 + 32]);
        cd_off = rd64(&z[e64 + 48]);
    }
    size_t p = (size_t)cd_off;
    for (uint64_t e = 0; e < n_entries; ++e) {
        if (p + 46 > z.size() || rd32(&z[p]) != 0x02014b50) throw std::runtime_error("failed to read file as npz");
        const uint16_t method = rd16(&z[p + 10]);
        uint64_t csize = rd32(&z[p + 20]), usize = rd32(&z[p + 24]), lho = rd32(&z[p + 42]);
        const uint16_t nl = rd16(&z[p + 28]), xl = rd16(&z[p + 30]), cl = rd16(&z[p + 32]);
        const std::string nm((const char*)&z[p + 46], nl);
        // zip64 extra field: values appear in the order usize, csize, offset for fields that are 0xFFFFFFFF
        size_t x = p + 46 + nl;
        const size_t xend = x + xl;
        while (x + 4 <= xend) {
            const uint16_t id = rd16(&z[x]), sz = rd16(&z[x + 2]);
            if (id == 0x0001) {
                size_t q = x + 4;
                if (usize == 0xFFFFFFFFu) { usize = rd64(&z[q]); q += 8; }
                if (csize == 0xFFFFFFFFu) { csize = rd64(&z[q]); q += 8; }
                if (lho == 0xFFFFFFFFu) { lho = rd64(&z[q]); q += 8; }
            }
            x += 4 + sz;
        }
        if (nm == name) {
            if (lho + 30 > z.size() || rd32(&z[lho]) != 0x04034b50) throw std::runtime_error("failed to read file as npz");
            const size_t data = (size_t)lho + 30 + rd16(&z[lho + 26]) + rd16(&z[lho + 28]);
            if (data + csize > z.size()) throw std::runtime_error("failed to read file as npz");
            if (method == 0) return std::vector<unsigned char>(z.begin() + data, z.begin() + data + csize);
            if (method != 8) throw std::runtime_error("npz member uses an unsupported compression method");
            std::vector<unsigned char> out((size_t)usize);
            z_stream zs;
            std::memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) throw std::runtime_error("zlib init failed");
            zs.next_in = const_cast<unsigned char*>(&z[data]);
            zs.avail_in = (uInt)csize;
            zs.next_out = out.data();
            zs.avail_out = (uInt)out.size();
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END) throw std::runtime_error("failed to inflate npz member " + name);
            return out;
        }
        p += 46 + nl + xl + cl;
    }
    throw std::runtime_error("npz has no member " + name);
}

}  // namespace msmhost
