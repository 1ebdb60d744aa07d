// Configuration schema of the simulator (common/src/parameters.rs,
// common/src/ics.rs, common/src/constants.rs) and the stream iterator
// (simulator/src/utils/io.rs:115-246).
#pragma once
#include <cmath>
#include <cstdint>
#include <optional>
#include <regex>
#include <string>
#include <vector>

#include "toml_lite.hpp"

namespace msmhost {

constexpr double POIS_CONST = 4.0 * 3.14159265358979323846 * 4.49e-12;  // common/src/constants.rs:2
constexpr double HBAR = 1.757e-90;                                        // :5
constexpr double LITTLE_H_TO_BIG_H = 1.022e-4;                            // :9

enum class IcsType { UserSpecified, ColdGauss, ColdGaussKSpace, SphericalTophat };
enum class SamplingScheme { Poisson = 0, Wigner = 1, Husimi = 2 };

struct InitialConditions {  // common/src/ics.rs:3-22
    IcsType type = IcsType::UserSpecified;
    std::string path;
    std::vector<double> mean, std;
    std::optional<uint64_t> phase_seed;
    double radius = 0, delta = 0, slope = 0;
};

struct CosmologyParameters {  // common/src/parameters.rs:71-86
    double omega_matter_now = 0, omega_radiation_now = 0, h = 0, z0 = 0;
    std::optional<double> max_dloga;
};

struct TomlSamplingParameters {  // :88-93
    SamplingScheme scheme = SamplingScheme::Wigner;
    std::vector<uint64_t> seeds;
};

struct TomlParameters {  // :10-55
    double axis_length = 0;
    std::optional<double> time;
    double final_sim_time = 0, cfl = 0;
    uint32_t num_data_dumps = 0;
    double total_mass = 0;
    std::optional<double> particle_mass, ntot, hbar_;
    std::string sim_name;
    double k2_cutoff = 0, alias_threshold = 0;
    size_t dims = 0, size = 0;
    InitialConditions ics;
    std::optional<TomlSamplingParameters> sampling;
    bool output_potential = false;
    std::optional<CosmologyParameters> cosmology;
};

// common/src/parameters.rs:148-202: `a..=b`, `a to b`, or a comma separated list (brackets allowed)
inline std::vector<uint64_t> parse_seeds(const std::string& s) {
    static const std::regex r1("\\d+..=\\d+"), r2("\\d+ to \\d+"), r3("(\\d+)");
    std::vector<uint64_t> out;
    auto range = [&](const std::string& sep) {
        const size_t p = s.find(sep);
        const uint64_t a = std::stoull(s.substr(0, p)), b = std::stoull(s.substr(p + sep.size()));
        for (uint64_t v = a; v <= b; ++v) out.push_back(v);
    };
    if (std::regex_search(s, r1)) { range("..="); return out; }
    if (std::regex_search(s, r2)) { range(" to "); return out; }
    for (auto it = std::sregex_iterator(s.begin(), s.end(), r3); it != std::sregex_iterator(); ++it)
        out.push_back(std::stoull((*it)[1].str()));
    if (out.empty())
        throw std::runtime_error("seeds did not match expected patterns: low..=high, low to high, [s1, s2, s3]");
    return out;
}

namespace detail {
inline const TomlValue& need(const TomlTable& t, const std::string& key) {
    auto it = t.find(key);
    if (it == t.end()) throw std::runtime_error("TomlParseError: missing field `" + key + "`");
    return it->second;
}
inline double need_f64(const TomlTable& t, const std::string& key) {
    const TomlValue& v = need(t, key);
    if (!v.is_number()) throw std::runtime_error("TomlParseError: invalid type for `" + key + "`: expected a number");
    return v.as_double();
}
inline long long need_int(const TomlTable& t, const std::string& key) {
    const TomlValue& v = need(t, key);
    if (v.kind != TomlValue::INT || v.i < 0)
        throw std::runtime_error("TomlParseError: invalid type for `" + key + "`: expected a non-negative integer");
    return v.i;
}
inline std::string need_str(const TomlTable& t, const std::string& key) {
    const TomlValue& v = need(t, key);
    if (v.kind != TomlValue::STRING) throw std::runtime_error("TomlParseError: invalid type for `" + key + "`: expected a string");
    return v.s;
}
inline std::optional<double> opt_f64(const TomlTable& t, const std::string& key) {
    auto it = t.find(key);
    if (it == t.end()) return std::nullopt;
    if (!it->second.is_number()) throw std::runtime_error("TomlParseError: invalid type for `" + key + "`: expected a number");
    return it->second.as_double();
}
inline std::vector<double> need_arr(const TomlTable& t, const std::string& key) {
    const TomlValue& v = need(t, key);
    if (v.kind != TomlValue::ARRAY) throw std::runtime_error("TomlParseError: invalid type for `" + key + "`: expected an array");
    return v.arr;
}
}  // namespace detail

// `with_cosmology` mirrors the cargo feature "expanding": the [cosmology] table is then required.
inline TomlParameters params_from_doc(const TomlDoc& doc, bool with_cosmology) {
    using namespace detail;
    TomlParameters p;
    const TomlTable& r = doc.root;
    p.axis_length = need_f64(r, "axis_length");
    p.time = opt_f64(r, "time");
    p.final_sim_time = need_f64(r, "final_sim_time");
    p.cfl = need_f64(r, "cfl");
    p.num_data_dumps = (uint32_t)need_int(r, "num_data_dumps");
    p.total_mass = need_f64(r, "total_mass");
    p.particle_mass = opt_f64(r, "particle_mass");
    p.ntot = opt_f64(r, "ntot");
    p.hbar_ = opt_f64(r, "hbar_");
    p.sim_name = need_str(r, "sim_name");
    p.k2_cutoff = need_f64(r, "k2_cutoff");
    p.alias_threshold = need_f64(r, "alias_threshold");
    p.dims = (size_t)need_int(r, "dims");
    p.size = (size_t)need_int(r, "size");
    if (auto it = r.find("output_potential"); it != r.end()) {
        if (it->second.kind != TomlValue::BOOL) throw std::runtime_error("TomlParseError: invalid type for `output_potential`");
        p.output_potential = it->second.b;
    }
    const TomlTable* ics = doc.table("ics");
    if (!ics) throw std::runtime_error("TomlParseError: missing field `ics`");
    const std::string type = need_str(*ics, "type");
    if (type == "UserSpecified") {
        p.ics.type = IcsType::UserSpecified;
        p.ics.path = need_str(*ics, "path");
    } else if (type == "ColdGauss") {
        p.ics.type = IcsType::ColdGauss;
        p.ics.mean = need_arr(*ics, "mean");
        p.ics.std = need_arr(*ics, "std");
    } else if (type == "ColdGaussKSpace") {
        p.ics.type = IcsType::ColdGaussKSpace;
        p.ics.mean = need_arr(*ics, "mean");
        p.ics.std = need_arr(*ics, "std");
        if (ics->count("phase_seed")) p.ics.phase_seed = (uint64_t)need_int(*ics, "phase_seed");
    } else if (type == "SphericalTophat") {
        p.ics.type = IcsType::SphericalTophat;
        p.ics.radius = need_f64(*ics, "radius");
        p.ics.delta = need_f64(*ics, "delta");
        p.ics.slope = need_f64(*ics, "slope");
    } else {
        throw std::runtime_error("TomlParseError: unknown variant `" + type +
                                 "`, expected one of `UserSpecified`, `ColdGauss`, `ColdGaussKSpace`, `SphericalTophat`");
    }
    if (const TomlTable* s = doc.table("sampling")) {
        TomlSamplingParameters sp;
        const std::string scheme = need_str(*s, "scheme");
        if (scheme == "Poisson") sp.scheme = SamplingScheme::Poisson;
        else if (scheme == "Wigner") sp.scheme = SamplingScheme::Wigner;
        else if (scheme == "Husimi") sp.scheme = SamplingScheme::Husimi;
        else throw std::runtime_error("TomlParseError: unknown variant `" + scheme + "`, expected one of `Poisson`, `Wigner`, `Husimi`");
        sp.seeds = parse_seeds(need_str(*s, "seeds"));
        p.sampling = sp;
    }
    if (const TomlTable* c = doc.table("cosmology")) {
        CosmologyParameters cp;
        cp.omega_matter_now = need_f64(*c, "omega_matter_now");
        cp.omega_radiation_now = need_f64(*c, "omega_radiation_now");
        cp.h = need_f64(*c, "h");
        cp.z0 = need_f64(*c, "z0");
        cp.max_dloga = opt_f64(*c, "max_dloga");
        p.cosmology = cp;
    } else if (with_cosmology) {
        throw std::runtime_error("TomlParseError: missing field `cosmology`");
    }
    return p;
}

// common/src/parameters.rs:96-107
inline TomlParameters read_toml(const std::string& path, bool with_cosmology) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("TomlReadError: failed to read toml at " + path);
    return params_from_doc(toml_parse(f), with_cosmology);
}

// common/src/parameters.rs:222-259: resolution order ntot -> particle_mass -> hbar_
inline void determine_pmass_hbar_(const TomlParameters& t, double& particle_mass, double& hbar_) {
    if (t.ntot) {
        particle_mass = t.total_mass / *t.ntot;
        hbar_ = t.hbar_ ? *t.hbar_ : HBAR / particle_mass;
    } else if (t.particle_mass) {
        particle_mass = *t.particle_mass;
        hbar_ = t.hbar_ ? *t.hbar_ : HBAR / particle_mass;
    } else if (t.hbar_) {
        hbar_ = *t.hbar_;
        particle_mass = HBAR / hbar_;
    } else {
        throw std::runtime_error(
            "You must specify the total mass and either exactly one of ntot (total number of particles) or "
            "particle_mass, or hbar_tilde ( hbar / particle_mass ).");
    }
}

// common/src/parameters.rs:204-220
inline double get_supercomoving_boxsize(double hbar_, const CosmologyParameters& c, double axis_length) {
    const double a0 = 1.0 / (1.0 + c.z0);
    const double comoving = axis_length / a0;
    const double hh = LITTLE_H_TO_BIG_H * c.h;
    return std::sqrt(std::sqrt(1.5 * c.omega_matter_now * hh * hh) / hbar_) * comoving;
}

// One entry per simulation the reference would run (simulator/src/utils/io.rs:183-241):
// N seeds -> N streams named `{sim}-stream{seed:05}`, then the mean-field run named `{sim}`.
struct StreamSpec {
    std::string sim_name;
    bool sampled = false;
    uint64_t seed = 0;
    SamplingScheme scheme = SamplingScheme::Wigner;
};

inline std::vector<StreamSpec> simulation_iter(const TomlParameters& t) {
    std::vector<StreamSpec> out;
    if (t.sampling) {
        for (uint64_t seed : t.sampling->seeds) {
            char buf[32];
            std::snprintf(buf, sizeof buf, "%05llu", (unsigned long long)seed);
            out.push_back({t.sim_name + "-stream" + buf, true, seed, t.sampling->scheme});
        }
    }
    out.push_back({t.sim_name, false, 0, SamplingScheme::Wigner});
    return out;
}

}  // namespace msmhost
