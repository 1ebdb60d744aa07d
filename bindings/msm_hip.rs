//! msm_hip.rs -- binding of libmsm_hip.so (include/msm.h) for the reference's simulator crate.
//!
//! Drop into `simulator/src/`, add `mod msm_hip;` to `lib.rs`, and let `build.rs` print
//! `cargo:rustc-link-search=native=$MSM_HIP_LIB_DIR`.  Source only: the image this library is built in has no Rust
//! toolchain, so the file is kept in step with the header by `tests/test_abi.py`, which compares the two symbol lists.
//!
//! `MsmStreams` mirrors the part of `SimulationObject` that touches the device
//! (simulator/src/simulation_object.rs:404-471 `new_from_params`, :475-661 `update`, :1113-1223 `dump`,
//! :1226-1228 `not_finished`), for all streams of a `SimulationIter` (utils/io.rs:115-246) at once.
#![allow(non_camel_case_types, dead_code)]

#[repr(C)]
pub struct MsmConfig {
    pub struct_size: i32, pub device: i32, pub dims: i32, pub size: i32,
    pub precision: i32, pub n_streams: i32, pub batch_streams: i32, pub expanding: i32,
    pub axis_length: f64, pub time: f64, pub final_sim_time: f64, pub cfl: f64,
    pub num_data_dumps: u32, pub reserved0: u32,
    pub total_mass: f64, pub particle_mass: f64, pub hbar_: f64,
    pub k2_cutoff: f64, pub alias_threshold: f64,
    pub omega_matter_now: f64, pub omega_radiation_now: f64, pub h: f64, pub z0: f64, pub max_dloga: f64,
    pub hip_stream: *mut core::ffi::c_void,
}
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct MsmStreamStatus {
    pub time: f64, pub dt: f64, pub max_phi: f64, pub alias_mass: f64, pub tau: f64,
    pub n_steps: i64, pub current_dumps: i32, pub dump: i32, pub active: i32, pub aliased: i32,
}
#[repr(C)] pub struct MsmCtx { _private: [u8; 0] }

#[link(name = "msm_hip")]
extern "C" {
    pub fn msm_create(cfg: *const MsmConfig, out: *mut *mut MsmCtx) -> i32;
    pub fn msm_destroy(ctx: *mut MsmCtx) -> i32;
    pub fn msm_last_error(ctx: *const MsmCtx) -> *const core::ffi::c_char;
    pub fn msm_set_psi(ctx: *mut MsmCtx, stream: i32, re: *const core::ffi::c_void, im: *const core::ffi::c_void) -> i32;
    pub fn msm_copy_psi(ctx: *mut MsmCtx, dst: i32, src: i32) -> i32;
    pub fn msm_sample(ctx: *mut MsmCtx, stream: i32, scheme: i32, seed: u64) -> i32;
    pub fn msm_fft(ctx: *mut MsmCtx, stream: i32, direction: i32) -> i32;
    pub fn msm_potential(ctx: *mut MsmCtx, first: i32, count: i32) -> i32;
    pub fn msm_step(ctx: *mut MsmCtx) -> i32;
    pub fn msm_step_begin(ctx: *mut MsmCtx, a_now: *const f64, tau_to_next_dump: *const f64) -> i32;
    pub fn msm_step_end(ctx: *mut MsmCtx, kick_a: *const f64, new_time: *const f64, new_tau: *const f64, snapped_tau: *const f64) -> i32;
    pub fn msm_sync(ctx: *mut MsmCtx) -> i32;
    pub fn msm_get_status(ctx: *mut MsmCtx, out: *mut MsmStreamStatus, n: i32) -> i32;
    pub fn msm_get_psi(ctx: *mut MsmCtx, stream: i32, re: *mut core::ffi::c_void, im: *mut core::ffi::c_void) -> i32;
    pub fn msm_get_psik(ctx: *mut MsmCtx, stream: i32, re: *mut core::ffi::c_void, im: *mut core::ffi::c_void) -> i32;
    pub fn msm_get_potential(ctx: *mut MsmCtx, stream: i32, phi: *mut core::ffi::c_void) -> i32;
    pub fn msm_density_sum(ctx: *mut MsmCtx, device_out_f64: *mut core::ffi::c_void) -> i32;
    pub fn msm_step_density(ctx: *mut MsmCtx, device_out_f64: *mut core::ffi::c_void) -> i32; // NULL switches it off
    pub fn msm_get_derived(ctx: *const MsmCtx, out: *mut f64 /* [8] */) -> i32;
    pub fn msm_get_batch(ctx: *const MsmCtx, out: *mut i32) -> i32;
    // dump pipeline / restart
    pub fn msm_get_psi_async(ctx: *mut MsmCtx, stream: i32, re: *mut core::ffi::c_void, im: *mut core::ffi::c_void, ticket: *mut i32) -> i32;
    pub fn msm_download_wait(ctx: *mut MsmCtx, ticket: i32) -> i32;
    pub fn msm_host_alloc(bytes: usize, out: *mut *mut core::ffi::c_void) -> i32;
    pub fn msm_host_free(p: *mut core::ffi::c_void) -> i32;
    pub fn msm_set_clock(ctx: *mut MsmCtx, stream: i32, time: f64, tau: f64, current_dumps: i32) -> i32;
    // in-situ synthesizer products (synthesizer/src/main.rs:63-187)
    pub fn msm_synth_accumulate(ctx: *mut MsmCtx, stream: i32, slot: i32) -> i32;
    pub fn msm_synth_device(ctx: *mut MsmCtx, slot: i32, device_sums: *mut *mut core::ffi::c_void, count: *mut i64) -> i32;
    pub fn msm_synth_set_count(ctx: *mut MsmCtx, slot: i32, count: i64) -> i32;
    pub fn msm_synth_fetch(ctx: *mut MsmCtx, slot: i32, field: i32, re: *mut f64, im: *mut f64) -> i32;
    pub fn msm_synth_qx(ctx: *mut MsmCtx, slot: i32, qx_re: *mut f64, qx_im: *mut f64) -> i32;
    pub fn msm_synth_qx_noalloc(ctx: *mut MsmCtx, slot: i32, device_out2: *mut f64) -> i32;
    pub fn msm_synth_release(ctx: *mut MsmCtx, slot: i32) -> i32;
    pub fn msm_synth_allreduce(ctx: *mut MsmCtx, slot: i32) -> i32;
    // several GPUs: RCCL inside the library
    pub fn msm_comm_unique_id(id_out: *mut u8 /* [128] */) -> i32;
    pub fn msm_comm_origin() -> *const core::ffi::c_char;
    pub fn msm_comm_init(ctx: *mut MsmCtx, id: *const u8, rank: i32, world: i32) -> i32;
    pub fn msm_comm_allreduce_f64(ctx: *mut MsmCtx, values: *mut f64, n: i32, op: i32) -> i32;
    pub fn msm_mean_density_begin(ctx: *mut MsmCtx, total_streams: i64) -> i32;
    pub fn msm_mean_density(ctx: *mut MsmCtx, host_or_null: *mut f64, device_or_null: *mut *mut core::ffi::c_void) -> i32;
    // measurement only
    pub fn msm_timer_start(ctx: *mut MsmCtx) -> i32;
    pub fn msm_timer_stop_ms(ctx: *mut MsmCtx, ms: *mut f32) -> i32;
    pub fn msm_profile(ctx: *mut MsmCtx, enable: i32) -> i32;
    pub fn msm_profile_report(ctx: *mut MsmCtx, max_kinds: i32, total_ms: *mut f64, launches: *mut i64, n_kinds: *mut i32) -> i32;
    pub fn msm_profile_kind_name(kind: i32) -> *const core::ffi::c_char;
}

pub const MSM_F64: i32 = 0;
pub const MSM_F32: i32 = 1;
pub const MSM_POISSON: i32 = 0;
pub const MSM_WIGNER: i32 = 1;
pub const MSM_HUSIMI: i32 = 2;

/// All streams of one TOML on one GPU.
pub struct MsmStreams {
    ctx: *mut MsmCtx,
    pub n_streams: usize,
    pub cells: usize,
    pub status: Vec<MsmStreamStatus>,
}

fn check(ctx: *const MsmCtx, rc: i32, what: &str) -> Result<(), String> {
    if rc == 0 {
        return Ok(());
    }
    let msg = unsafe { std::ffi::CStr::from_ptr(msm_last_error(ctx)) }.to_string_lossy().into_owned();
    Err(format!("{what} failed ({rc}): {msg}"))
}

impl MsmStreams {
    /// `SimulationObject::new_from_params` for every seed and the mean-field run: `psi0` is the
    /// mean-field wavefunction as two planes (what every builder of ics.rs ends with); `seeds[s]` is `Some((scheme,
    /// seed))` for sampled streams (ics.rs:434-648), `None` for the mean-field run.
    pub fn new_from_params(cfg: &MsmConfig, re: &[f64], im: &[f64], seeds: &[Option<(i32, u64)>]) -> Result<Self, String> {
        let mut ctx: *mut MsmCtx = std::ptr::null_mut();
        check(std::ptr::null(), unsafe { msm_create(cfg, &mut ctx) }, "msm_create")?;
        let me = MsmStreams { ctx, n_streams: seeds.len(), cells: re.len(), status: vec![MsmStreamStatus::default(); seeds.len()] };
        check(ctx, unsafe { msm_set_psi(ctx, 0, re.as_ptr() as *const _, im.as_ptr() as *const _) }, "msm_set_psi")?;
        for s in 1..seeds.len() {
            check(ctx, unsafe { msm_copy_psi(ctx, s as i32, 0) }, "msm_copy_psi")?;
        }
        for (s, sd) in seeds.iter().enumerate() {
            if let Some((scheme, seed)) = sd {
                check(ctx, unsafe { msm_sample(ctx, s as i32, *scheme, *seed) }, "msm_sample")?;
            }
        }
        Ok(me)
    }

    /// One `SimulationObject::update` of every stream that has not finished (static box).  Returns the streams whose
    /// step landed on a dump time; an aliased stream is reported as the reference's panic text (:607-617).
    pub fn update(&mut self) -> Result<Vec<usize>, String> {
        check(self.ctx, unsafe { msm_step(self.ctx) }, "msm_step")?;
        check(self.ctx, unsafe { msm_get_status(self.ctx, self.status.as_mut_ptr(), self.n_streams as i32) }, "msm_get_status")?;
        if let Some((s, st)) = self.status.iter().enumerate().find(|(_, st)| st.aliased != 0) {
            return Err(format!("simulation aliased Fourier aliasing occurred; p_mass is {} (stream {s})", st.alias_mass));
        }
        Ok(self.status.iter().enumerate().filter(|(_, st)| st.dump != 0).map(|(s, _)| s).collect())
    }

    /// `not_finished` over all streams (main.rs:65).
    pub fn not_finished(&self) -> bool {
        self.status.iter().any(|st| st.active != 0)
    }

    /// The device-to-host half of `dump`: the two planes `complex_array_to_disk` serialises (utils/io.rs:34-88).
    pub fn dump(&mut self, stream: usize, re: &mut [f64], im: &mut [f64]) -> Result<(), String> {
        assert!(re.len() == self.cells && im.len() == self.cells);
        check(self.ctx, unsafe { msm_get_psi(self.ctx, stream as i32, re.as_mut_ptr() as *mut _, im.as_mut_ptr() as *mut _) }, "msm_get_psi")
    }
}

impl Drop for MsmStreams {
    fn drop(&mut self) {
        unsafe { msm_destroy(self.ctx) };
    }
}
