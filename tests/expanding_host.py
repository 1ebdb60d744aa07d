"""Host side of an expanding-box step for tests that drive the C ABI directly (msm_step_begin / msm_step_end).

The library does the grid work; the scale-factor bookkeeping between the two calls is the caller's
(simulation_object.rs:726-760, :939-990).  These helpers compute that bookkeeping from an oracle
`SimulationObject` WITHOUT advancing it, in the reference's arithmetic, the way msm-simulator's loop does."""
import copy

import sp_oracle as o


def step_inputs(sim):
    """(a_now, tau to the next dump, tau of the next dump time) before a step (:939-990, :628-631)."""
    p = sim.parameters
    T = p.T
    t_next = T(p.current_dumps + 1) * p.final_sim_time / T(p.num_data_dumps)
    ttnd = t_next - p.time
    tau_next = T(o.get_tau(float(p.time + ttnd), p.cosmo)) - p.tau
    snapped = T(o.get_tau(float(t_next), p.cosmo))
    return float(sim.scale_factor_solver.get_a()), float(tau_next), float(snapped)


def half_kicks(sim, dtau):
    """(a1 + a2, new time, new tau) of the two half-kicks for a step of `dtau` (:726-760)."""
    sh = copy.copy(sim)
    sh.parameters = copy.copy(sim.parameters)
    sh.scale_factor_solver = sim.scale_factor_solver.clone()
    p = sh.parameters
    T = p.T
    dtau = T(dtau)
    a_sum = T(0)
    for _ in range(2):
        a_sum = a_sum + T(sh.scale_factor_solver.get_a())
        dt_myr = T(sh._calculate_dt_from_dtau(dtau / T(2.0)))
        sh.scale_factor_solver.step(float(dt_myr))
        p.time = T(p.time + dt_myr)
        p.tau = T(p.tau + dtau / T(2.0))
    return float(a_sum), float(p.time), float(p.tau)


def update(ev, sims):
    """One expanding update of every stream of `ev`; sims[s] is the oracle object that shadows stream s and is
    advanced by its own update() as well.  Returns the oracle's info dicts."""
    n = len(sims)
    pre = [step_inputs(sim) for sim in sims]
    ev.step_begin([x[0] for x in pre], [x[1] for x in pre])
    st = ev.status()
    post = [half_kicks(sims[s], st[s].dt) if st[s].active else (0.0, float(sims[s].parameters.time), float(sims[s].parameters.tau))
            for s in range(n)]
    ev.step_end([x[0] for x in post], [x[1] for x in post], [x[2] for x in post], [x[2] for x in pre])
    return [sim.update() if sim.not_finished() else None for sim in sims]
