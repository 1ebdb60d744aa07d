// Scale factor a(t) and the supercomoving time tau(t) for the expanding box
// (simulator/src/expanding.rs, simulator/src/utils/mod.rs:14-43,
// simulator/src/simulation_object.rs:1344-1388 and :1408-1453).
//
// The reference delegates a(t) to the `cosmology` 0.2.0 crate, which is not in
// its tree; this is the flat LCDM Friedmann equation da/dt = a H0 sqrt(Om a^-3 +
// Or a^-4 + Ode) integrated with classical RK4 in sub-steps bounded by max_dloga.
#pragma once
#include <cmath>
#include <functional>

#include "params.hpp"

namespace msmhost {

constexpr double DEFAULT_MAX_DLOGA = 1e-3;  // simulator/src/expanding.rs:27

class ScaleFactorSolver {
  public:
    explicit ScaleFactorSolver(const CosmologyParameters& c)
        : om_(c.omega_matter_now), orad_(c.omega_radiation_now), ode_(1.0 - c.omega_matter_now - c.omega_radiation_now),
          h0_(c.h * LITTLE_H_TO_BIG_H), max_dloga_(c.max_dloga ? *c.max_dloga : DEFAULT_MAX_DLOGA),
          a_(1.0 / (1.0 + c.z0)), t_(0.0) {
        if (om_ + orad_ > 1.0) throw std::runtime_error("Only flat cosmologies are supported; pick an omega_matter_now + omega_radiation_now <= 1.0");
        if (c.z0 < 0.0) throw std::runtime_error("Choose a positive initial redshift z0 >= 0.0");
        if (om_ < 0.0 || orad_ < 0.0) throw std::runtime_error("densities must be non-negative");
    }
    double get_a() const { return a_; }
    double get_dadt() const { return dadt(a_); }
    double get_time() const { return t_; }
    double step(double dt) {
        if (dt != 0.0) {
            int n = (int)std::ceil(std::fabs(dt) * dadt(a_) / a_ / max_dloga_);
            if (n < 1) n = 1;
            const double h = dt / n;
            double a = a_;
            for (int i = 0; i < n; ++i) {
                const double k1 = dadt(a), k2 = dadt(a + 0.5 * h * k1), k3 = dadt(a + 0.5 * h * k2), k4 = dadt(a + h * k3);
                a = a + h * (k1 + 2.0 * k2 + 2.0 * k3 + k4) / 6.0;
            }
            a_ = a;
            t_ += dt;
        }
        return a_;
    }

  private:
    double dadt(double a) const { return a * h0_ * std::sqrt(om_ / (a * a * a) + orad_ / (a * a * a * a) + ode_); }
    double om_, orad_, ode_, h0_, max_dloga_, a_, t_;
};

// simulator/src/utils/mod.rs:14-43
template <typename F>
double rk4(F&& f, double tn, double yn, double h) {
    const double k1 = f(tn, yn);
    const double k2 = f(tn + h / 2.0, yn + h * k1 / 2.0);
    const double k3 = f(tn + h / 2.0, yn + h * k2 / 2.0);
    const double k4 = f(tn + h, yn + h * k3);
    return yn + h * (k1 + 2.0 * k2 + 2.0 * k3 + k4) / 6.0;
}

inline double tau_prefactor(const CosmologyParameters& c) {
    const double hh = LITTLE_H_TO_BIG_H * c.h;
    return std::sqrt(1.5 * c.omega_matter_now * hh * hh);
}

// simulation_object.rs:1408-1453: tau(t) = int_0^t sqrt(3/2 Om H0^2) / a^2 dt, >= 1000 RK4 steps
inline double get_tau(double target_time, const CosmologyParameters& c) {
    ScaleFactorSolver solver(c);
    const double pref = tau_prefactor(c);
    auto dtau_dt = [&](double t, double) {
        const double a = solver.step(t - solver.get_time());
        return pref / (a * a);
    };
    double tau = 0.0, time = 0.0;
    while (time < target_time) {
        double dt = target_time / 1000.0;
        if (c.max_dloga) dt = std::min(dt, solver.get_a() / solver.get_dadt() * *c.max_dloga);
        dt = std::min(dt, target_time - time);
        tau = rk4(dtau_dt, time, tau, dt);
        time += dt;
    }
    return tau;
}

// simulation_object.rs:1344-1388: the time step (Myr) that advances tau by dtau, one RK4 step on a clone
inline double calculate_dt_from_dtau(const ScaleFactorSolver& state, const CosmologyParameters& c, double tau, double time, double dtau) {
    ScaleFactorSolver solver = state;
    const double pref = tau_prefactor(c);
    auto dt_dtau = [&](double, double t) {
        const double a = solver.step(t - solver.get_time());
        return 1.0 / (pref / (a * a));
    };
    return rk4(dt_dtau, tau, time, dtau) - time;
}

}  // namespace msmhost
