// C entry points into the host-side pieces (TOML schema, seeds grammar, stream iterator, IC
// builders, NPY writer, scale-factor solver) so that they can be exercised without the CLI.
// libmsm_host.so has no GPU dependency.
#include <cstdio>
#include <cstring>
#include <sstream>
#include <string>

#include "cosmo.hpp"
#include "expanding_loop.hpp"
#include "ics.hpp"
#include "npy_io.hpp"
#include "params.hpp"

using namespace msmhost;

namespace {
thread_local std::string g_err;

CosmologyParameters cosmo_from(const double c[5], int has_max_dloga) {
    CosmologyParameters p;
    p.omega_matter_now = c[0]; p.omega_radiation_now = c[1]; p.h = c[2]; p.z0 = c[3];
    if (has_max_dloga) p.max_dloga = c[4];
    return p;
}
std::string jnum(double v) { char b[64]; std::snprintf(b, sizeof b, "%.17g", v); return b; }
std::string jopt(const std::optional<double>& v) { return v ? jnum(*v) : "null"; }
std::string jarr(const std::vector<double>& v) {
    std::string s = "[";
    for (size_t i = 0; i < v.size(); ++i) s += (i ? ", " : "") + jnum(v[i]);
    return s + "]";
}
}  // namespace

extern "C" {

const char* msmh_last_error() { return g_err.c_str(); }

// returns the number of seeds (written up to cap), or -1
int msmh_parse_seeds(const char* s, unsigned long long* out, int cap) {
    try {
        const std::vector<uint64_t> v = parse_seeds(s);
        for (int i = 0; i < (int)v.size() && i < cap; ++i) out[i] = v[i];
        return (int)v.size();
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// JSON description of the parsed configuration and of the simulations it expands to
int msmh_describe_toml(const char* path, int expanding, char* buf, int cap) {
    try {
        const TomlParameters t = read_toml(path, expanding != 0);
        double pm, hb;
        determine_pmass_hbar_(t, pm, hb);
        std::ostringstream o;
        o << "{\"axis_length\": " << jnum(t.axis_length) << ", \"time\": " << jopt(t.time)
          << ", \"final_sim_time\": " << jnum(t.final_sim_time) << ", \"cfl\": " << jnum(t.cfl)
          << ", \"num_data_dumps\": " << t.num_data_dumps << ", \"total_mass\": " << jnum(t.total_mass)
          << ", \"particle_mass\": " << jopt(t.particle_mass) << ", \"ntot\": " << jopt(t.ntot) << ", \"hbar_\": " << jopt(t.hbar_)
          << ", \"sim_name\": \"" << t.sim_name << "\", \"k2_cutoff\": " << jnum(t.k2_cutoff)
          << ", \"alias_threshold\": " << jnum(t.alias_threshold) << ", \"dims\": " << t.dims << ", \"size\": " << t.size
          << ", \"output_potential\": " << (t.output_potential ? "true" : "false")
          << ", \"resolved_particle_mass\": " << jnum(pm) << ", \"resolved_hbar_\": " << jnum(hb);
        static const char* ics_names[] = {"UserSpecified", "ColdGauss", "ColdGaussKSpace", "SphericalTophat"};
        o << ", \"ics\": {\"type\": \"" << ics_names[(int)t.ics.type] << "\", \"path\": \"" << t.ics.path << "\", \"mean\": " << jarr(t.ics.mean)
          << ", \"std\": " << jarr(t.ics.std) << ", \"radius\": " << jnum(t.ics.radius) << ", \"delta\": " << jnum(t.ics.delta)
          << ", \"slope\": " << jnum(t.ics.slope) << "}";
        if (t.sampling) {
            static const char* sch[] = {"Poisson", "Wigner", "Husimi"};
            o << ", \"sampling\": {\"scheme\": \"" << sch[(int)t.sampling->scheme] << "\", \"seeds\": [";
            for (size_t i = 0; i < t.sampling->seeds.size(); ++i) o << (i ? ", " : "") << t.sampling->seeds[i];
            o << "]}";
        } else {
            o << ", \"sampling\": null";
        }
        if (t.cosmology)
            o << ", \"cosmology\": {\"omega_matter_now\": " << jnum(t.cosmology->omega_matter_now) << ", \"omega_radiation_now\": "
              << jnum(t.cosmology->omega_radiation_now) << ", \"h\": " << jnum(t.cosmology->h) << ", \"z0\": " << jnum(t.cosmology->z0)
              << ", \"max_dloga\": " << jopt(t.cosmology->max_dloga) << "}";
        else
            o << ", \"cosmology\": null";
        o << ", \"streams\": [";
        const auto specs = simulation_iter(t);
        for (size_t i = 0; i < specs.size(); ++i)
            o << (i ? ", " : "") << "{\"sim_name\": \"" << specs[i].sim_name << "\", \"sampled\": " << (specs[i].sampled ? "true" : "false")
              << ", \"seed\": " << specs[i].seed << "}";
        o << "]}";
        const std::string s = o.str();
        if ((int)s.size() + 1 > cap) { g_err = "buffer too small"; return -1; }
        std::memcpy(buf, s.c_str(), s.size() + 1);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// initial conditions named by the TOML, as two planes of T (precision 0 = f64, 1 = f32)
int msmh_build_ics(const char* toml_path, int expanding, int precision, double dx, void* re, void* im) {
    try {
        const TomlParameters t = read_toml(toml_path, expanding != 0);
        auto go = [&](auto tag) {
            using T = decltype(tag);
            Planes<T> g;
            switch (t.ics.type) {
                case IcsType::UserSpecified: g = user_specified_ics<T>(t.ics.path, t.dims, t.size); break;
                case IcsType::ColdGauss: g = cold_gauss<T>(t.ics.mean, t.ics.std, (T)dx, t.dims, t.size); break;
                case IcsType::SphericalTophat:
                    g = spherical_tophat<T>((T)t.axis_length, (T)dx, t.dims, t.size, t.ics.radius, t.ics.delta, t.ics.slope);
                    break;
                case IcsType::ColdGaussKSpace:   // the k-space field; the transform to real space happens on the device
                    g = cold_gauss_kspace_k<T>(t.ics.mean, t.ics.std, (T)dx, t.dims, t.size, t.ics.phase_seed.value_or(0));
                    break;
            }
            std::memcpy(re, g.re.data(), g.re.size() * sizeof(T));
            std::memcpy(im, g.im.data(), g.im.size() * sizeof(T));
        };
        if (precision == 0) go(double{}); else go(float{});
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

int msmh_write_npy(const char* path, const void* data, int precision, const unsigned long long* shape, int ndim) {
    try {
        std::vector<size_t> shp(shape, shape + ndim);
        if (precision == 0) write_npy<double>(path, (const double*)data, shp);
        else write_npy<float>(path, (const float*)data, shp);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// advance a fresh solver by dts[0..n) and report a after each
int msmh_scale_factor(const double cosmo[5], int has_max_dloga, const double* dts, int n, double* a_out) {
    try {
        ScaleFactorSolver s(cosmo_from(cosmo, has_max_dloga));
        for (int i = 0; i < n; ++i) a_out[i] = s.step(dts[i]);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

double msmh_get_tau(double t, const double cosmo[5], int has_max_dloga) { return get_tau(t, cosmo_from(cosmo, has_max_dloga)); }

double msmh_supercomoving_boxsize(double hbar_, const double cosmo[5], double axis_length) {
    return get_supercomoving_boxsize(hbar_, cosmo_from(cosmo, 0), axis_length);
}

// dt (Myr) for a step of dtau from a solver that has been advanced by t_elapsed from t = 0
double msmh_dt_from_dtau(const double cosmo[5], int has_max_dloga, double t_elapsed, double tau, double dtau) {
    const CosmologyParameters c = cosmo_from(cosmo, has_max_dloga);
    ScaleFactorSolver s(c);
    s.step(t_elapsed);
    return calculate_dt_from_dtau(s, c, tau, t_elapsed, dtau);
}

// ---- the expanding-box host loop of host/expanding_loop.hpp for callers outside the CLI (benchmarks, tests) ----
// precision 0: time and tau kept in double, 1: in float (the type the reference would be compiled with)
struct msmh_expanding {
    int precision;
    ExpandingHost<double>* d;
    ExpandingHost<float>* f;
};

void* msmh_expanding_new(const double cosmo[5], int has_max_dloga, int precision, int n_streams, double time0,
                         double final_sim_time, unsigned num_data_dumps) {
    try {
        if (n_streams < 1 || num_data_dumps < 1) { g_err = "n_streams and num_data_dumps must be >= 1"; return nullptr; }
        const CosmologyParameters c = cosmo_from(cosmo, has_max_dloga);
        auto* h = new msmh_expanding{precision, nullptr, nullptr};
        if (precision == 0) h->d = new ExpandingHost<double>(c, n_streams, time0, final_sim_time, num_data_dumps);
        else h->f = new ExpandingHost<float>(c, n_streams, time0, final_sim_time, num_data_dumps);
        return h;
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void msmh_expanding_free(void* p) {
    auto* h = (msmh_expanding*)p;
    if (!h) return;
    delete h->d;
    delete h->f;
    delete h;
}
// inputs of msm_step_begin and the snapped tau of msm_step_end, from every stream's count of dumps written
int msmh_expanding_begin(void* p, const int* current_dumps, double* a_now, double* tau_to_next_dump, double* snapped_tau) {
    try {
        auto* h = (msmh_expanding*)p;
        if (h->d) h->d->begin(current_dumps, a_now, tau_to_next_dump, snapped_tau);
        else h->f->begin(current_dumps, a_now, tau_to_next_dump, snapped_tau);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
// the two half-kicks: advances the solvers of active streams by the device's d tau, fills msm_step_end's inputs
int msmh_expanding_advance(void* p, const double* dtau, const int* active, double* kick_a, double* new_time, double* new_tau) {
    try {
        auto* h = (msmh_expanding*)p;
        if (h->d) h->d->advance(dtau, active, kick_a, new_time, new_tau);
        else h->f->advance(dtau, active, kick_a, new_time, new_tau);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
// the clocks the device reports after msm_step_end (snapped at dumps), for all streams
int msmh_expanding_commit(void* p, const double* time, const double* tau) {
    auto* h = (msmh_expanding*)p;
    const int n = h->d ? h->d->streams() : h->f->streams();
    for (int s = 0; s < n; ++s) {
        if (h->d) h->d->commit(s, time[s], tau[s]);
        else h->f->commit(s, time[s], tau[s]);
    }
    return 0;
}

}  // extern "C"
