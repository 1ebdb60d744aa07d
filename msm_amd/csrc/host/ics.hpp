// Initial-condition builders of simulator/src/ics.rs, evaluated on the host in
// the simulation's real type T, producing two C-order planes (re, im) that go
// to msm_set_psi.  Flat index a*N^2 + b*N + c is the reference's device order.
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "npy_io.hpp"
#include "params.hpp"

namespace msmhost {

template <typename T>
struct Planes {
    std::vector<T> re, im;
    explicit Planes(size_t n = 0) : re(n, T(0)), im(n, T(0)) {}
};

inline size_t ipow(size_t b, size_t e) { size_t r = 1; while (e--) r *= b; return r; }

// utils/grid.rs:11-33: psi *= sqrt(dx^-dims / sum |psi|^2)
template <typename T>
void normalize(Planes<T>& g, T dx, size_t dims) {
    T norm = 0;
    for (size_t i = 0; i < g.re.size(); ++i) norm += g.re[i] * g.re[i] + g.im[i] * g.im[i];
    const T f = std::sqrt(std::pow(dx, -(T)dims) / norm);
    for (size_t i = 0; i < g.re.size(); ++i) { g.re[i] *= f; g.im[i] *= f; }
}

// ics.rs:165-280: sqrt(1 + delta / (1 + exp(slope (r/R - 1)))) at the cell centres of the proper
// box (dx = axis_length / size), normalised with the PARAMETER dx (comoving when expanding).
template <typename T>
Planes<T> spherical_tophat(T axis_length, T dx_param, size_t dims, size_t size, double radius, double delta, double slope) {
    const T dx = axis_length / (T)size;
    std::vector<T> x(size);
    for (size_t i = 0; i < size; ++i) x[i] = (T)(2 * i + 1) * dx / (T)2.0;
    const T half = axis_length / (T)2.0;
    const size_t ny = dims >= 2 ? size : 1, nz = dims == 3 ? size : 1;
    Planes<T> g(size * ny * nz);
    size_t p = 0;
    for (size_t a = 0; a < size; ++a)
        for (size_t b = 0; b < ny; ++b)
            for (size_t c = 0; c < nz; ++c) {
                const T xi = x[a] - half;
                const T yj = (dims >= 2 ? x[b] : half) - half;
                const T zk = (dims == 3 ? x[c] : half) - half;
                const T r = std::sqrt(xi * xi + yj * yj + zk * zk);
                const T ramp = (T)1 / ((T)1 + std::exp((T)slope * (r / (T)radius - (T)1)));
                g.re[p++] = std::sqrt((T)1 + (T)delta * ramp);
            }
    normalize(g, dx_param, dims);
    return g;
}

// ics.rs:24-162: product of per-axis real Gaussians, every factor and the product normalised with
// (dx, dims); mean[0]/std[0] belong to the FASTEST axis (the last index of the dumped array).
template <typename T>
Planes<T> cold_gauss(const std::vector<double>& mean, const std::vector<double>& std_, T dx, size_t dims, size_t size) {
    if (mean.size() != dims) throw std::runtime_error("Cold Gauss: Mean vector provided has incorrect dimensionality");
    if (std_.size() != dims) throw std::runtime_error("Cold Gauss: Std vector provided has incorrect dimensionality");
    std::vector<T> x(size);
    for (size_t i = 0; i < size; ++i) x[i] = (T)(2 * i + 1) * dx / (T)2.0;
    std::vector<std::vector<T>> fac(3, std::vector<T>(1, (T)1));
    for (size_t ax = 0; ax < dims; ++ax) {
        Planes<T> g(size);
        for (size_t i = 0; i < size; ++i)
            g.re[i] = std::exp((T)-0.5 * std::pow((x[i] - (T)mean[ax]) / (T)std_[ax], (T)2.0));
        normalize(g, dx, dims);
        fac[ax] = g.re;
    }
    const size_t n0 = fac[0].size(), n1 = fac[1].size(), n2 = fac[2].size();  // fastest .. slowest
    Planes<T> g(n0 * n1 * n2);
    for (size_t c = 0; c < n2; ++c)
        for (size_t b = 0; b < n1; ++b)
            for (size_t a = 0; a < n0; ++a) g.re[(c * n1 + b) * n0 + a] = (fac[0][a] * fac[1][b]) * fac[2][c];
    normalize(g, dx, dims);
    return g;
}

// Philox4x32-10 (Random123) on the host: the uniform stream of the k-space phases.
inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// element e's uniform in (0,1): first 53 bits of the block with counter (e_lo, e_hi, 0, 2), key = seed
inline double philox_uniform(uint64_t e, uint64_t seed) {
    uint32_t r[4];
    philox4x32_10((uint32_t)e, (uint32_t)(e >> 32), 0u, 2u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const uint64_t b = (((uint64_t)r[1] << 32) | r[0]) >> 11;
    return ((double)b + 0.5) * 0x1.0p-53;
}

// ics.rs:282-431 up to (not including) the transform: separable Gaussians in k (get_kgrid values,
// utils/fft.rs:100-120), every factor and the product normalised with (dk = dx, dims), times
// exp(2 pi i u), u uniform from the Philox engine seeded with phase_seed (default 0; ArrayFire's own
// uniform stream is not reproducible from the reference tree).  The caller finishes with the FORWARD
// unitary transform (forward_inplace, ics.rs:425) on the device.
template <typename T>
Planes<T> cold_gauss_kspace_k(const std::vector<double>& mean, const std::vector<double>& std_, T dx, size_t dims, size_t size,
                              uint64_t seed) {
    if (mean.size() != dims) throw std::runtime_error("Cold Gauss k-Space: Mean vector provided has incorrect dimensionality");
    if (std_.size() != dims) throw std::runtime_error("Cold Gauss k-Space: Std vector provided has incorrect dimensionality");
    std::vector<T> k(size);
    for (size_t i = 0; i < size; ++i) {
        const long ii = i < size / 2 ? (long)i : (long)i - (long)size;
        k[i] = (T)ii / ((T)size * dx);
    }
    std::vector<std::vector<T>> fac(3, std::vector<T>(1, (T)1));
    for (size_t ax = 0; ax < dims; ++ax) {
        Planes<T> g(size);
        for (size_t i = 0; i < size; ++i)
            g.re[i] = std::exp((T)-0.5 * std::pow((k[i] - (T)mean[ax]) / (T)std_[ax], (T)2.0));
        normalize(g, dx, dims);
        fac[ax] = g.re;
    }
    const size_t n0 = fac[0].size(), n1 = fac[1].size(), n2 = fac[2].size();  // fastest .. slowest
    Planes<T> g(n0 * n1 * n2);
    for (size_t c = 0; c < n2; ++c)
        for (size_t b = 0; b < n1; ++b)
            for (size_t a = 0; a < n0; ++a) g.re[(c * n1 + b) * n0 + a] = (fac[0][a] * fac[1][b]) * fac[2][c];
    normalize(g, dx, dims);
    const T twopi = (T)(2.0 * 3.14159265358979323846);
    for (size_t e = 0; e < g.re.size(); ++e) {
        const T ph = twopi * (T)philox_uniform((uint64_t)e, seed);
        const T amp = g.re[e];
        g.re[e] = amp * std::cos(ph);
        g.im[e] = amp * std::sin(ph);
    }
    return g;
}

// ics.rs:650-730: npz members real.npy / imag.npy of dtype T, cubic, ndim = dims, edge = size;
// taken as they are (no renormalisation).
template <typename T>
Planes<T> user_specified_ics(const std::string& path, size_t dims, size_t size) {
    const std::vector<unsigned char> z = read_file(path, "ics file does not exist");
    Planes<T> g;
    const char* names[2] = {"real.npy", "imag.npy"};
    const char* errs[2] = {"couldn't read real part of field", "couldn't read imag part of field"};
    for (int part = 0; part < 2; ++part) {
        std::vector<unsigned char> raw;
        try { raw = zip_member(z, names[part]); } catch (const std::exception& e) { throw std::runtime_error(std::string(errs[part]) + " (" + e.what() + ")"); }
        const NpyArray a = parse_npy(raw.data(), raw.size(), errs[part]);
        if (a.descr != NpyDescr<T>::value || a.fortran_order) throw std::runtime_error(std::string(errs[part]) + ": expected C-order " + NpyDescr<T>::value);
        if (a.shape.empty() || a.shape.size() > 3) throw std::runtime_error("User specified ICs have invalid number of dimensions.");
        for (size_t s : a.shape) if (s != a.shape[0]) throw std::runtime_error("Only uniform grids are supported at this time");
        if (a.shape.size() != dims)
            throw std::runtime_error("Dimensions of user-provided data (" + std::to_string(a.shape.size()) +
                                     ") do not match the dimensions specified in the toml (" + std::to_string(dims) + ")");
        if (a.shape[0] != size)
            throw std::runtime_error("Grid size of user-provided data (" + std::to_string(a.shape[0]) +
                                     ") does not match the size specified in the toml (" + std::to_string(size) + ")");
        if (a.bytes.size() < a.count() * sizeof(T)) throw std::runtime_error(std::string(errs[part]) + ": truncated");
        std::vector<T>& dst = part == 0 ? g.re : g.im;
        dst.resize(a.count());
        std::memcpy(dst.data(), a.bytes.data(), a.count() * sizeof(T));
    }
    return g;
}

}  // namespace msmhost
