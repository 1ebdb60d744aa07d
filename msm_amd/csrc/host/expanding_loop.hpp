// Host side of one expanding-box update for many streams: the scalars that go into msm_step_begin and msm_step_end
// and the scale-factor solvers that are advanced between the two (simulator/src/simulation_object.rs:726-760 for the
// two half-kicks, :939-990 for the time to the next dump in supercomoving time, :628-630 for the snapped dump time).
// One object per device context; every array is indexed by the context's stream number.  T is the precision the
// reference would have been compiled in: its time and tau are of type T and are rounded where the reference rounds.
#pragma once
#include <map>
#include <vector>

#include "cosmo.hpp"

namespace msmhost {

template <typename T>
class ExpandingHost {
  public:
    ExpandingHost(const CosmologyParameters& c, int n_streams, double time0, double final_sim_time, unsigned num_data_dumps)
        : cosmo_(c), ns_(n_streams), time0_(time0), tf_((T)final_sim_time), nd_((T)num_data_dumps),
          solvers_((size_t)n_streams, ScaleFactorSolver(c)), time_((size_t)n_streams, (T)time0), tau_((size_t)n_streams, (T)0) {
        for (int s = 0; s < ns_; ++s) tau_[s] = (T)tau_of((double)time_[s]);
    }
    int streams() const { return ns_; }
    // tau(t) of simulation_object.rs:1408-1453 costs >= 1000 RK4 steps; dump times repeat, so results are kept
    double tau_of(double t) {
        auto it = tau_cache_.find(t);
        if (it == tau_cache_.end()) it = tau_cache_.emplace(t, get_tau(t, cosmo_)).first;
        return it->second;
    }
    // restart: the stream stands at its dump `dump_index` (a(t) re-integrated from the start)
    void place_at_dump(int s, int dump_index) {
        time_[s] = (T)dump_index * tf_ / nd_;
        solvers_[s] = ScaleFactorSolver(cosmo_);
        if (dump_index > 0) solvers_[s].step((double)time_[s] - time0_);
        tau_[s] = (T)tau_of((double)time_[s]);
    }
    // inputs of msm_step_begin (a_now, tau_to_next_dump) and the snapped tau msm_step_end wants, from every stream's
    // count of dumps written so far
    void begin(const int* current_dumps, double* a_now, double* tau_to_next_dump, double* snapped_tau) {
        for (int s = 0; s < ns_; ++s) {
            a_now[s] = solvers_[s].get_a();
            const T next = (T)(current_dumps[s] + 1) * tf_ / nd_;
            const T ttnd = next - time_[s];
            tau_to_next_dump[s] = (double)((T)tau_of((double)(time_[s] + ttnd)) - tau_[s]);
            snapped_tau[s] = (double)(T)tau_of((double)next);
        }
    }
    // between msm_step_begin and msm_step_end: dtau[s] is the step the device chose; two half-kicks, the scale
    // factor read before each and advanced after it (:726-760).  Streams that are not active keep their state.
    void advance(const double* dtau_in, const int* active, double* kick_a, double* new_time, double* new_tau) {
        for (int s = 0; s < ns_; ++s) {
            if (!active[s]) { kick_a[s] = 0; new_time[s] = (double)time_[s]; new_tau[s] = (double)tau_[s]; continue; }
            const T dtau = (T)dtau_in[s];
            T a_sum = 0, t_new = time_[s], tau_new = tau_[s];
            for (int half = 0; half < 2; ++half) {
                a_sum += (T)solvers_[s].get_a();
                const T dt_myr = (T)calculate_dt_from_dtau(solvers_[s], cosmo_, (double)tau_new, (double)t_new, (double)(dtau / (T)2.0));
                solvers_[s].step((double)dt_myr);
                t_new = t_new + dt_myr;
                tau_new = tau_new + dtau / (T)2.0;
            }
            kick_a[s] = (double)a_sum;
            new_time[s] = (double)t_new;
            new_tau[s] = (double)tau_new;
        }
    }
    // after msm_step_end: the stream's clock as the device left it (snapped to the dump time when the step was a dump)
    void commit(int s, double time, double tau) { time_[s] = (T)time; tau_[s] = (T)tau; }
    T time(int s) const { return time_[s]; }
    T tau(int s) const { return tau_[s]; }
    double a(int s) const { return solvers_[s].get_a(); }

  private:
    CosmologyParameters cosmo_;
    int ns_;
    double time0_;
    T tf_, nd_;
    std::vector<ScaleFactorSolver> solvers_;
    std::vector<T> time_, tau_;
    std::map<double, double> tau_cache_;
};

}  // namespace msmhost
