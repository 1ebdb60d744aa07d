// msm-simulator: the reference's process surface (simulator/src/main.rs:9-89) on top of the
// C ABI of include/msm.h.  TOML in, .npz initial conditions in, sim-data/{name}/psi_{k:05}_real|_imag
// NPY dumps out.  Where the reference runs its N + 1 simulations one after the other
// (main.rs:43-79), this driver advances them together in one device context.
//
//   msm-simulator --toml <path> [-v|--verbose] [--test]
//                 [--expanding] [--precision f64|f32] [--device N] [--batch B] [--synthesize] [--resume]
//                 [--rank R --world W [--comm-file PATH]] [--mean-density] [--wave STREAMS]
//
// --expanding stands for the reference's cargo feature of the same name (compile-time there).
// --synthesize also writes the msm-synthesizer's products (synthesizer/src/main.rs:63-187) for the seed
//   streams, accumulated on the device as each stream reaches a dump: sim-data/{sim_name}-combined/.
// --resume restarts every stream from its last complete dump under sim-data/ (the reference cannot: its dumps
//   are write-only, simulation_object.rs:627,915); streams without dumps start from the initial conditions.
// --rank R --world W: this process evolves block R of the seed list split into W blocks (one process per GPU,
//   streams are independent: main.rs:43-79); the mean-field run goes to rank 0.
// --wave STREAMS: at most that many streams share the device at a time; the rest follow in further waves (the
//   reference runs them one by one anyway).  Without it the driver starts with all of them and halves the wave
//   whenever the device reports that it is out of memory.
// --comm-file PATH: the W processes also join one RCCL communicator (W = 1 is allowed: a communicator of one) -- rank 0 publishes the
//   128-byte id in PATH (a name unique to this launch, on a file system all ranks see), the others wait for it.
//   They then stop together when a stream aliases anywhere, and --synthesize works across the ranks: the
//   synthesizer's sums are all-reduced per dump index and rank 0 writes the -combined/ products.
// --mean-density (with --comm-file): the ensemble density <|psi|^2> over all ranks' seed streams is formed inside every
//   step and all-reduced beside the next one (msm_mean_density_begin: the synthesizer's psi2, synthesizer/src/lib.rs:217-229,
//   in situ and per step).  Whenever a step lands every seed stream of every rank on the same dump k, rank 0 writes it
//   as sim-data/{sim_name}-combined/mean_density_{k:05} (one real NPY array).  The mean-field run is evolved after
//   the seed streams, in a context of its own (the synthesizer leaves it out, lib.rs:190-193).
// A rank that fails leaves a note in PATH.abort; a watchdog thread in every other rank sees it and ends that rank with a
//   non-zero status, also when it is blocked inside a collective.  (PATH must be new for every launch: a PATH.abort left
//   by an earlier launch ends the next one at once, with the old note in the message.)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <functional>
#include <future>
#include <mutex>
#include <condition_variable>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/msm.h"
#include "cosmo.hpp"
#include "expanding_loop.hpp"
#include "ics.hpp"
#include "npy_io.hpp"
#include "params.hpp"

using namespace msmhost;

namespace {

constexpr size_t MAX_CONCURRENT_GRID_WRITES = 16;   // simulation_object.rs:39 (x2 for real + imag)
const char* SIM_DATA_FOLDER = "sim-data";           // simulation_object.rs:1116

struct Args {
    std::string toml;
    bool verbose = false, test = false, expanding = false, synthesize = false, resume = false, mean_density = false;
    int precision = MSM_F64, device = 0, batch = 0, rank = 0, world = 1, wave = 0;
    std::string comm_file;
};

[[noreturn]] void usage(const char* why) {
    std::fprintf(stderr, "error: %s\n\nUSAGE:\n    msm-simulator [OPTIONS] --toml <TOML>\n\nOPTIONS:\n"
                         "    -t, --toml <TOML>\n    -v, --verbose\n        --test\n        --expanding\n"
                         "        --precision <f64|f32>\n        --device <N>\n        --batch <STREAMS>\n        --synthesize\n        --resume\n        --rank <R> --world <W> [--comm-file <PATH>]\n        --wave <STREAMS>\n", why);
    std::exit(2);
}

Args parse_args(int argc, char** argv) {
    Args a;
    for (int i = 1; i < argc; ++i) {
        const std::string s = argv[i];
        auto val = [&]() -> std::string {
            if (i + 1 >= argc) usage(("a value is required for '" + s + "'").c_str());
            return argv[++i];
        };
        if (s == "--toml" || s == "-t") a.toml = val();
        else if (s.rfind("--toml=", 0) == 0) a.toml = s.substr(7);
        else if (s == "--verbose" || s == "-v") a.verbose = true;
        else if (s == "--test") a.test = true;
        else if (s == "--expanding") a.expanding = true;
        else if (s == "--synthesize") a.synthesize = true;
        else if (s == "--resume") a.resume = true;
        else if (s == "--mean-density") a.mean_density = true;
        else if (s == "--rank") a.rank = std::atoi(val().c_str());
        else if (s == "--world") a.world = std::atoi(val().c_str());
        else if (s == "--comm-file") a.comm_file = val();
        else if (s == "--wave") a.wave = std::atoi(val().c_str());
        else if (s == "--precision") { const std::string p = val(); if (p == "f64") a.precision = MSM_F64; else if (p == "f32") a.precision = MSM_F32; else usage("--precision takes f64 or f32"); }
        else if (s == "--device") a.device = std::atoi(val().c_str());
        else if (s == "--batch") a.batch = std::atoi(val().c_str());
        else usage(("Found argument '" + s + "' which wasn't expected").c_str());
    }
    if (a.toml.empty()) usage("The following required arguments were not provided: --toml <TOML>");
    if (a.world < 1 || a.rank < 0 || a.rank >= a.world) usage("--rank must lie in [0, --world)");
    if (a.world > 1 && a.synthesize && a.comm_file.empty()) usage("--synthesize over several processes needs --comm-file (the ranks' sums are all-reduced)");
    // a resumed stream has passed dumps whose ensemble sums were never formed in this process
    if (a.resume && a.synthesize) usage("--resume cannot be combined with --synthesize");
    if (a.mean_density && a.comm_file.empty()) usage("--mean-density needs --comm-file (the density is all-reduced over the communicator; --world 1 is allowed)");
    if (a.mean_density && a.resume) usage("--resume cannot be combined with --mean-density");
    return a;
}

void check(msm_ctx* ctx, int rc, const char* what) {
    if (rc != MSM_OK) throw std::runtime_error(std::string(what) + " failed: " + msm_last_error(ctx));
}

// ---- several ranks: a failing rank must not leave the others waiting in a collective for ever
std::string abort_marker(const std::string& comm_file) { return comm_file + ".abort"; }
void leave_abort_note(const Args& a, const std::string& why) {
    if (a.comm_file.empty()) return;
    const std::string tmp = abort_marker(a.comm_file) + ".tmp" + std::to_string(a.rank);
    if (FILE* f = std::fopen(tmp.c_str(), "w")) {
        std::fprintf(f, "rank %d: %s\n", a.rank, why.c_str());
        std::fclose(f);
        std::error_code ec;
        std::filesystem::rename(tmp, abort_marker(a.comm_file), ec);
    }
}
bool read_abort_note(const std::string& comm_file, std::string& note) {
    std::error_code ec;
    if (!std::filesystem::exists(abort_marker(comm_file), ec)) return false;
    note.clear();
    if (FILE* f = std::fopen(abort_marker(comm_file).c_str(), "r")) {
        char buf[512];
        if (std::fgets(buf, sizeof buf, f)) note = buf;
        std::fclose(f);
    }
    while (!note.empty() && (note.back() == '\n' || note.back() == '\r')) note.pop_back();
    return true;
}
// Polls for another rank's note while this rank works (or waits inside RCCL) and ends the process when one appears.
class AbortWatch {
  public:
    AbortWatch(const Args& a) : on_(!a.comm_file.empty()) {
        if (!on_) return;
        th_ = std::thread([this, file = a.comm_file, rank = a.rank] {
            while (!stop_.load()) {
                std::string note;
                if (read_abort_note(file, note)) {
                    std::fprintf(stderr, "rank %d: stopping, another rank failed (%s)\n", rank, note.c_str());
                    std::fflush(nullptr);
                    std::_Exit(1);   // the main thread may be blocked in a collective that will never complete
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(100));
            }
        });
    }
    ~AbortWatch() { stop(); }
    void stop() {
        if (!on_) return;
        stop_.store(true);
        if (th_.joinable()) th_.join();
        on_ = false;
    }

  private:
    bool on_;
    std::atomic<bool> stop_{false};
    std::thread th_;
};

// dump writers: two NPY files per array, written on worker threads, at most 32 in flight
// (utils/io.rs:34-88, simulation_object.rs:1123-1136)
class DumpWriters {
  public:
    template <typename T>
    void submit(const std::string& base, std::vector<T>&& re, std::vector<T>&& im, const std::vector<size_t>& shape) {
        while (active_.size() >= MAX_CONCURRENT_GRID_WRITES * 2) drain(false);
        auto task = [](std::string path, std::vector<T> data, std::vector<size_t> shp) { write_npy<T>(path, data.data(), shp); };
        active_.push_back(std::async(std::launch::async, task, base + "_real", std::move(re), shape));
        active_.push_back(std::async(std::launch::async, task, base + "_imag", std::move(im), shape));
    }
    // a task that waits for its own data (an asynchronous download) and then writes
    void submit_task(std::function<void()> task) {
        while (active_.size() >= MAX_CONCURRENT_GRID_WRITES * 2) drain(false);
        active_.push_back(std::async(std::launch::async, std::move(task)));
    }
    void drain(bool all) {
        for (size_t i = 0; i < active_.size();) {
            if (all || active_[i].wait_for(std::chrono::milliseconds(all ? 0 : 10)) == std::future_status::ready) {
                active_[i].get();   // rethrows a failed write
                active_[i] = std::move(active_.back());
                active_.pop_back();
            } else {
                ++i;
            }
        }
    }

  private:
    std::vector<std::future<void>> active_;
};

// One device context's worth of streams, start to finish.  `nomem` reports that the device could not hold them
// (nothing has been written then): the caller retries with fewer streams per wave.
template <typename T>
int run_wave(const Args& args, const TomlParameters& toml, const std::vector<StreamSpec>& streams, bool& nomem) {
    nomem = false;
    const bool lead = args.rank == 0;   // holds the mean-field run and writes the ensemble products
    const int ns = (int)streams.size();
    double particle_mass, hbar_;
    determine_pmass_hbar_(toml, particle_mass, hbar_);
    if (toml.dims < 1 || toml.dims > 3) throw std::runtime_error("dims must be 1, 2 or 3");

    msm_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = (int32_t)sizeof cfg;
    cfg.device = args.device;
    cfg.dims = (int32_t)toml.dims;
    cfg.size = (int32_t)toml.size;
    cfg.precision = args.precision;
    cfg.n_streams = ns;
    cfg.batch_streams = args.batch;
    cfg.expanding = args.expanding ? 1 : 0;
    cfg.axis_length = toml.axis_length;
    cfg.time = toml.time.value_or(0.0);
    cfg.final_sim_time = toml.final_sim_time;
    cfg.cfl = toml.cfl;
    cfg.num_data_dumps = toml.num_data_dumps;
    cfg.total_mass = toml.total_mass;
    cfg.particle_mass = particle_mass;
    cfg.hbar_ = hbar_;
    cfg.k2_cutoff = toml.k2_cutoff;
    cfg.alias_threshold = toml.alias_threshold;
    if (args.expanding) {
        cfg.omega_matter_now = toml.cosmology->omega_matter_now;
        cfg.omega_radiation_now = toml.cosmology->omega_radiation_now;
        cfg.h = toml.cosmology->h;
        cfg.z0 = toml.cosmology->z0;
        cfg.max_dloga = toml.cosmology->max_dloga.value_or(0.0);
    }
    msm_ctx* ctx = nullptr;
    if (const int rc = msm_create(&cfg, &ctx); rc != MSM_OK) {
        if (rc == MSM_ERR_NOMEM) { nomem = true; return 0; }
        throw std::runtime_error(std::string("msm_create failed: ") + msm_last_error(nullptr));
    }

    // ---- several processes: one RCCL communicator (include/msm.h: msm_comm_*)
    const bool comm = !args.comm_file.empty();
    if (comm) {
        unsigned char id[MSM_COMM_ID_BYTES];
        if (lead) {
            if (msm_comm_unique_id(id) != MSM_OK) throw std::runtime_error(std::string("msm_comm_unique_id failed: ") + msm_last_error(nullptr));
            const std::string tmp = args.comm_file + ".tmp";
            FILE* f = std::fopen(tmp.c_str(), "wb");
            if (!f || std::fwrite(id, 1, sizeof id, f) != sizeof id) throw std::runtime_error("cannot write " + tmp);
            std::fclose(f);
            std::filesystem::rename(tmp, args.comm_file);   // appears complete or not at all
        } else {
            const auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                std::error_code ec;
                if (std::filesystem::file_size(args.comm_file, ec) == sizeof id && !ec) {
                    FILE* f = std::fopen(args.comm_file.c_str(), "rb");
                    const bool ok = f && std::fread(id, 1, sizeof id, f) == sizeof id;
                    if (f) std::fclose(f);
                    if (ok) break;
                }
                std::string note;
                if (read_abort_note(args.comm_file, note)) throw std::runtime_error("another rank failed before the communicator was formed (" + note + ")");
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(300)) throw std::runtime_error("no communicator id appeared in " + args.comm_file);
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
            }
        }
        check(ctx, msm_comm_init(ctx, id, args.rank, args.world), "msm_comm_init");
        if (lead) std::filesystem::remove(args.comm_file);   // every rank has joined: the id is spent
    }
    // values every rank must agree on before it acts (op: RCCL's 0 sum, 2 max, 3 min)
    auto agree = [&](double v, int op) {
        if (comm) check(ctx, msm_comm_allreduce_f64(ctx, &v, 1, op), "msm_comm_allreduce_f64");
        return v;
    };

    // per-step ensemble density over every rank's seed streams (this context holds seed streams only then: run())
    const bool mean_on = comm && args.mean_density;
    if (mean_on) {
        const double total = agree((double)ns, 0);
        check(ctx, msm_mean_density_begin(ctx, (int64_t)(total + 0.5)), "msm_mean_density_begin");
    }

    double derived[8];
    check(ctx, msm_get_derived(ctx, derived), "msm_get_derived");
    const T dx = (T)derived[0];
    const size_t N = toml.size, n_cells = ipow(N, toml.dims);

    // ---- initial conditions (simulation_object.rs:406-435)
    Planes<T> ics;
    switch (toml.ics.type) {
        case IcsType::UserSpecified: ics = user_specified_ics<T>(toml.ics.path, toml.dims, N); break;
        case IcsType::ColdGauss: ics = cold_gauss<T>(toml.ics.mean, toml.ics.std, dx, toml.dims, N); break;
        case IcsType::SphericalTophat:
            ics = spherical_tophat<T>((T)toml.axis_length, dx, toml.dims, N, toml.ics.radius, toml.ics.delta, toml.ics.slope);
            break;
        case IcsType::ColdGaussKSpace:
            ics = cold_gauss_kspace_k<T>(toml.ics.mean, toml.ics.std, dx, toml.dims, N, toml.ics.phase_seed.value_or(0));
            break;
    }
    check(ctx, msm_set_psi(ctx, 0, ics.re.data(), ics.im.data()), "msm_set_psi");
    if (toml.ics.type == IcsType::ColdGaussKSpace)   // k space -> real space is the reference's forward_inplace (ics.rs:425)
        check(ctx, msm_fft(ctx, 0, MSM_FORWARD), "msm_fft");
    for (int s = 1; s < ns; ++s) check(ctx, msm_copy_psi(ctx, s, 0), "msm_copy_psi");
    for (int s = 0; s < ns; ++s)
        if (streams[s].sampled) check(ctx, msm_sample(ctx, s, (int)streams[s].scheme, streams[s].seed), "msm_sample");
    check(ctx, msm_sync(ctx), "msm_sync");

    if (args.verbose) {
        for (const StreamSpec& sp : streams) {
            std::printf("\n\nWorking on simulation %s\nSimulation Parameters\n%s\n", sp.sim_name.c_str(), std::string(40, '-').c_str());
            std::printf("axis_length         = %.6e\n", toml.axis_length);
            if (args.expanding) std::printf("comoving_boxsize    = %.6e\n", derived[6]);
            std::printf("dx                  = %.6e\ncurrent_time        = %.6e\nfinal_sim_time      = %.6e\n", derived[0], cfg.time, toml.final_sim_time);
            std::printf("cfl                 = %.6e\nnum_data_dumps      = %u\ntotal_mass          = %.6e\n", toml.cfl, toml.num_data_dumps, toml.total_mass);
            std::printf("particle_mass       = %.6e\nhbar_               = %.6e\nsim_name            = %s\n", particle_mass, hbar_, sp.sim_name.c_str());
            std::printf("k2_cutoff           = %.6e\nalias_threshold     = %.6e\nk2_max              = %.6e\n", toml.k2_cutoff, toml.alias_threshold, derived[2]);
            std::printf("n_tot               = %.6e\ndims                = %zu\nsize                = %zu\n%s\n", derived[4], toml.dims, toml.size, std::string(40, '-').c_str());
            if (sp.sampled) std::printf("\n[sampling_parameters]\nsampling_scheme     = %s\nseed                = %llu\n",
                                        sp.scheme == SamplingScheme::Wigner ? "Wigner" : sp.scheme == SamplingScheme::Husimi ? "Husimi" : "Poisson",
                                        (unsigned long long)sp.seed);
            std::printf("Physical Constants\nHBAR = %.5e\nPOIS_CONSTANT = %.5e\n", HBAR, POIS_CONST);
        }
    }
    if (args.test) { msm_destroy(ctx); return 0; }

    // ---- dumps (simulation_object.rs:1113-1223)
    std::vector<size_t> shape(4, 1);
    for (size_t d = 0; d < toml.dims; ++d) shape[d] = N;
    // pinned host buffers for the dump downloads: the copy runs on the library's second stream while the
    // next steps execute, a writer thread waits for it and serialises the two planes (utils/io.rs:58-85)
    struct PinnedPool {
        std::vector<std::pair<T*, T*>> free;
        std::mutex m;
        std::condition_variable cv;
    } pool;
    const int n_pinned = 8;
    for (int i = 0; i < n_pinned; ++i) {
        void *a = nullptr, *b = nullptr;
        if (msm_host_alloc(n_cells * sizeof(T), &a) != MSM_OK || msm_host_alloc(n_cells * sizeof(T), &b) != MSM_OK)
            throw std::runtime_error("pinned host allocation failed");
        pool.free.emplace_back((T*)a, (T*)b);
    }
    DumpWriters writers;   // declared after the pool: its pending tasks are joined before the pool goes away
    auto dump = [&](int s, int k) {
        const std::string dir = std::string(SIM_DATA_FOLDER) + "/" + streams[s].sim_name;
        std::filesystem::create_directories(dir);
        char name[64];
        std::snprintf(name, sizeof name, "/psi_%05d", k);
        std::pair<T*, T*> buf;
        {
            std::unique_lock<std::mutex> lk(pool.m);
            while (pool.free.empty()) { lk.unlock(); writers.drain(false); lk.lock(); if (pool.free.empty()) pool.cv.wait_for(lk, std::chrono::milliseconds(5)); }
            buf = pool.free.back();
            pool.free.pop_back();
        }
        int32_t ticket = -1;
        check(ctx, msm_get_psi_async(ctx, s, buf.first, buf.second, &ticket), "msm_get_psi_async");
        const std::string base = dir + name;
        writers.submit_task([ctx, ticket, buf, base, shape, &pool] {
            if (msm_download_wait(ctx, ticket) != MSM_OK) throw std::runtime_error("msm_download_wait failed");
            auto re_done = std::async(std::launch::async, [&] { write_npy<T>(base + "_real", buf.first, shape); });
            write_npy<T>(base + "_imag", buf.second, shape);
            re_done.get();
            { std::lock_guard<std::mutex> lk(pool.m); pool.free.push_back(buf); }
            pool.cv.notify_one();
        });
        if (toml.output_potential) {
            std::snprintf(name, sizeof name, "/potential_%05d", k);
            std::vector<T> phi(n_cells), zero(n_cells, (T)0);
            check(ctx, msm_potential(ctx, s, 1), "msm_potential");
            check(ctx, msm_get_potential(ctx, s, phi.data()), "msm_get_potential");
            writers.submit<T>(dir + name, std::move(phi), std::move(zero), shape);
        }
    };
    // ---- in-situ synthesizer (synthesizer/src/main.rs:63-187): seed streams only (the mean-field run is the last
    // entry on the lead rank).  A stream adds itself to the slot of a dump index when it gets there; a slot is
    // finished -- summed over the ranks, averaged, written by the lead rank -- once every rank's seed streams have.
    int n_seed_streams = 0;
    for (const StreamSpec& sp : streams) n_seed_streams += sp.sampled ? 1 : 0;   // they come first; the mean-field run is last
    std::vector<int> synth_count(toml.num_data_dumps + 1, 0);
    std::vector<std::pair<double, double>> qx(toml.num_data_dumps + 1, {0.0, 0.0});
    std::vector<char> qx_done(toml.num_data_dumps + 1, 0);
    int synth_next = 0;
    // seed streams over all ranks: without any there is nothing to synthesize (a TOML without [sampling])
    const bool synth_on = args.synthesize && agree((double)n_seed_streams, 0) > 0.5;
    const std::string combined = std::string(SIM_DATA_FOLDER) + "/" + toml.sim_name + "-combined";
    auto synthesize = [&](int s, int k) {
        if (!synth_on || s >= n_seed_streams || k < 0 || k > (int)toml.num_data_dumps) return;
        check(ctx, msm_synth_accumulate(ctx, s, k), "msm_synth_accumulate");
        ++synth_count[k];
    };
    // collective when there is a communicator: every rank calls it once per loop iteration
    auto flush_synth = [&]() {
        if (!synth_on) return;
        for (; synth_next <= (int)toml.num_data_dumps; ++synth_next) {
            const int k = synth_next;
            if (agree(synth_count[k] == n_seed_streams ? 1.0 : 0.0, 3) < 0.5) break;
            if (comm) check(ctx, msm_synth_allreduce(ctx, k), "msm_synth_allreduce");
            if (lead) {
                std::filesystem::create_directories(combined);
                static const char* fields[4] = {"psi", "psi2", "psik", "psik2"};
                for (int f = 0; f < 4; ++f) {
                    char name[64];
                    std::snprintf(name, sizeof name, "/%s_%05d", fields[f], k);
                    std::vector<double> re(n_cells), im(n_cells);
                    check(ctx, msm_synth_fetch(ctx, k, f, re.data(), im.data()), "msm_synth_fetch");
                    writers.submit<double>(combined + name, std::move(re), std::move(im), shape);
                }
                check(ctx, msm_synth_qx(ctx, k, &qx[k].first, &qx[k].second), "msm_synth_qx");
                qx_done[k] = 1;
            }
            check(ctx, msm_synth_release(ctx, k), "msm_synth_release");
        }
    };
    // ---- restart: last complete dump of every stream (both planes present and of the right shape)
    std::vector<int> resumed(ns, -1);
    if (args.resume) {
        for (int s = 0; s < ns; ++s) {
            const std::string dir = std::string(SIM_DATA_FOLDER) + "/" + streams[s].sim_name;
            for (int k = (int)toml.num_data_dumps; k >= 0 && resumed[s] < 0; --k) {
                char name[64];
                std::snprintf(name, sizeof name, "/psi_%05d", k);
                try {
                    Planes<T> g(0);
                    for (int part = 0; part < 2; ++part) {
                        const std::vector<unsigned char> raw = read_file(dir + name + (part ? "_imag" : "_real"), "no such dump");
                        const NpyArray a = parse_npy(raw.data(), raw.size(), "dump");
                        if (a.descr != NpyDescr<T>::value || a.fortran_order || a.count() != n_cells || a.bytes.size() < n_cells * sizeof(T))
                            throw std::runtime_error("dump has another shape or type");
                        std::vector<T>& dst = part ? g.im : g.re;
                        dst.resize(n_cells);
                        std::memcpy(dst.data(), a.bytes.data(), n_cells * sizeof(T));
                    }
                    check(ctx, msm_set_psi(ctx, s, g.re.data(), g.im.data()), "msm_set_psi");
                    resumed[s] = k;
                } catch (const std::exception&) { /* try the previous dump */ }
            }
            if (resumed[s] >= 0) std::printf("(%s) resuming from dump %d\n", streams[s].sim_name.c_str(), resumed[s]);
        }
    }
    for (int s = 0; s < ns; ++s) if (resumed[s] < 0) { dump(s, 0); synthesize(s, 0); }   // main.rs:61
    flush_synth();

    // ---- expanding-box host state (simulation_object.rs:726-760, :939-990): host/expanding_loop.hpp
    std::unique_ptr<ExpandingHost<T>> xhost;
    if (args.expanding) xhost = std::make_unique<ExpandingHost<T>>(*toml.cosmology, ns, cfg.time, toml.final_sim_time, toml.num_data_dumps);
    for (int s = 0; s < ns; ++s) {
        if (resumed[s] < 0) continue;
        const T t_dump = (T)resumed[s] * (T)toml.final_sim_time / (T)toml.num_data_dumps;   // the snapped dump time (:628-630)
        if (xhost) xhost->place_at_dump(s, resumed[s]);
        check(ctx, msm_set_clock(ctx, s, (double)t_dump, xhost ? (double)xhost->tau(s) : 0.0, resumed[s]), "msm_set_clock");
    }

    std::vector<msm_stream_status> st(ns), prev(ns);
    check(ctx, msm_get_status(ctx, prev.data(), ns), "msm_get_status");
    const auto t_loop = std::chrono::steady_clock::now();
    bool aliased_here = false, mean_skipped_note = false;
    for (;;) {
        bool any_active = false;
        for (int s = 0; s < ns; ++s) any_active = any_active || prev[s].active;
        if (comm) {
            // one exchange per iteration: "a stream is still running somewhere" and "a stream aliased somewhere".  All
            // ranks leave the loop in the same iteration; none is left waiting in a collective the others never enter.
            double flags[2] = {any_active ? 1.0 : 0.0, aliased_here ? 1.0 : 0.0};
            check(ctx, msm_comm_allreduce_f64(ctx, flags, 2, 2), "msm_comm_allreduce_f64");
            if (flags[1] > 0.5) {
                writers.drain(true);
                if (!aliased_here) std::fprintf(stderr, "rank %d: stopping, a stream on another rank aliased\n", args.rank);
                msm_destroy(ctx);
                return 101;
            }
            if (flags[0] < 0.5) break;
            // idle here, others still stepping: stay in the exchanges -- and, when every step carries the density
            // all-reduce, in the steps too (a step without live streams only contributes its zeros to the sum)
            if (!any_active && !mean_on) { flush_synth(); continue; }
        } else if (!any_active) {
            break;   // main.rs:65: while not_finished()
        }
        if (!args.expanding) {
            check(ctx, msm_step(ctx), "msm_step");
        } else {
            std::vector<double> a_now(ns), tau_next(ns), kick_a(ns), new_time(ns), new_tau(ns), snapped(ns), dtau(ns);
            std::vector<int> dumps(ns), live(ns);
            for (int s = 0; s < ns; ++s) { dumps[s] = prev[s].current_dumps; live[s] = prev[s].active; }
            xhost->begin(dumps.data(), a_now.data(), tau_next.data(), snapped.data());
            check(ctx, msm_step_begin(ctx, a_now.data(), tau_next.data()), "msm_step_begin");
            check(ctx, msm_get_status(ctx, st.data(), ns), "msm_get_status");
            for (int s = 0; s < ns; ++s) dtau[s] = st[s].dt;
            xhost->advance(dtau.data(), live.data(), kick_a.data(), new_time.data(), new_tau.data());
            check(ctx, msm_step_end(ctx, kick_a.data(), new_time.data(), new_tau.data(), snapped.data()), "msm_step_end");
        }
        check(ctx, msm_get_status(ctx, st.data(), ns), "msm_get_status");
        for (int s = 0; s < ns; ++s) {
            if (st[s].n_steps == prev[s].n_steps) continue;   // stream was already finished
            if (st[s].aliased) {
                // the reference panics here (simulation_object.rs:607-617)
                std::fprintf(stderr, "thread 'main' panicked at 'simulation aliased Fourier aliasing occurred; p_mass is %g and the "
                                     "threshold was set at %g with k2_cutoff = %g' (%s)\n",
                             st[s].alias_mass, toml.alias_threshold, toml.k2_cutoff, streams[s].sim_name.c_str());
                aliased_here = true;
                if (comm) break;          // the other ranks learn of it at the next exchange; all exit there
                writers.drain(true);
                msm_destroy(ctx);
                return 101;
            }
            if (xhost) xhost->commit(s, st[s].time, st[s].tau);
            if (st[s].dump) {
                dump(s, st[s].current_dumps);
                synthesize(s, st[s].current_dumps);
                if (args.expanding) std::printf("(%s) z = %g  [%d/%u]\n", streams[s].sim_name.c_str(), 1.0 / xhost->a(s) - 1.0, st[s].current_dumps, toml.num_data_dumps);
                else std::printf("(%s) t = %g  [%d/%u]\n", streams[s].sim_name.c_str(), st[s].time, st[s].current_dumps, toml.num_data_dumps);
            }
            if (!st[s].active && args.verbose)
                std::printf("Finished %s in %lld seconds\n", streams[s].sim_name.c_str(),
                            (long long)std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - t_loop).count());
        }
        if (mean_on) {
            // did this step land EVERY seed stream of EVERY rank on the same dump?  (k here, or -1) -> max and -min
            int k_here = -1;
            bool same = true;
            for (int s = 0; s < ns; ++s) {
                const bool dumped = st[s].n_steps != prev[s].n_steps && st[s].dump && !st[s].aliased;
                if (!dumped) { same = false; break; }
                if (k_here < 0) k_here = st[s].current_dumps;
                same = same && st[s].current_dumps == k_here;
            }
            if (!same) k_here = -1;
            double kk[2] = {(double)k_here, (double)-k_here};
            check(ctx, msm_comm_allreduce_f64(ctx, kk, 2, 2), "msm_comm_allreduce_f64");
            if (kk[0] >= 0 && kk[0] == -kk[1]) {
                if (lead) {
                    std::vector<double> mean(n_cells);
                    check(ctx, msm_mean_density(ctx, mean.data(), nullptr), "msm_mean_density");
                    std::filesystem::create_directories(combined);
                    char name[64];
                    std::snprintf(name, sizeof name, "/mean_density_%05d", (int)kk[0]);
                    writers.submit_task([path = combined + name, data = std::move(mean), shape] { write_npy<double>(path, data.data(), shape); });
                }
            } else if (!mean_skipped_note) {
                mean_skipped_note = true;
                if (lead) std::printf("(mean density) the streams do not reach their dumps in the same step here: the per-step product is "
                                      "written only for steps that land every stream on one dump\n");
            }
        }
        prev = st;
        flush_synth();
    }
    if (synth_on && lead) {
        // Qx time series over the dumps (post_combine, synthesizer/src/lib.rs:351-609), shape (n_dumps + 1, 1, 1, 1)
        std::vector<double> re, im;
        for (size_t k = 0; k < qx.size(); ++k)
            if (qx_done[k]) { re.push_back(qx[k].first); im.push_back(qx[k].second); }
        if (!re.empty()) {
            const std::vector<size_t> qshape = {re.size(), 1, 1, 1};
            writers.submit<double>(combined + "/Qx", std::move(re), std::move(im), qshape);
        }
    }
    writers.drain(true);
    for (auto& b : pool.free) { msm_host_free(b.first); msm_host_free(b.second); }
    pool.free.clear();
    msm_destroy(ctx);
    return 0;
}

template <typename T>
int run(const Args& args) {
    const auto t_start = std::chrono::steady_clock::now();
    const TomlParameters toml = read_toml(args.toml, args.expanding);
    std::vector<StreamSpec> streams = simulation_iter(toml);
    if (args.world > 1) {
        // block partition of the seed streams (ceil(n / world) each); the trailing mean-field run stays with rank 0
        const size_t n_seed = streams.size() - 1, per = (n_seed + args.world - 1) / args.world;
        const size_t first = std::min(per * (size_t)args.rank, n_seed), last = std::min(first + per, n_seed);
        std::vector<StreamSpec> mine(streams.begin() + first, streams.begin() + last);
        if (args.rank == 0) mine.push_back(streams.back());
        streams.swap(mine);
        if (!args.comm_file.empty() && last == first) throw std::runtime_error("--comm-file: more ranks than seed streams");
        if (streams.empty()) { std::printf("rank %d of %d: no streams\n", args.rank, args.world); return 0; }
    }
    // --mean-density: the context that all-reduces its density every step holds seed streams only; the mean-field run
    // (last entry, lead rank) follows in a context of its own, outside the communicator
    std::vector<StreamSpec> mean_field_after;
    if (args.mean_density) {
        if (!streams.empty() && !streams.back().sampled) { mean_field_after.push_back(streams.back()); streams.pop_back(); }
        if (streams.empty()) throw std::runtime_error("--mean-density needs seed streams ([sampling] in the TOML) on every rank");
    }
    const bool multi_stream = streams.size() + mean_field_after.size() > 2;   // SimulationIter::len() counts the seeds (io.rs:121-125)
    // The reference runs its simulations one after the other (main.rs:43-79), so any grouping of them is legal: all
    // at once when the device holds them, otherwise in waves of as many as fit (halved until msm_create succeeds).
    size_t wave = streams.size();
    if (args.wave > 0 && (size_t)args.wave < wave) {
        if (args.synthesize || !args.comm_file.empty()) throw std::runtime_error("--wave cannot be combined with --synthesize or --comm-file");
        wave = (size_t)args.wave;
    }
    for (size_t w0 = 0; w0 < streams.size();) {
        const size_t w1 = std::min(w0 + wave, streams.size());
        const std::vector<StreamSpec> part(streams.begin() + w0, streams.begin() + w1);
        bool nomem = false;
        const int rc = run_wave<T>(args, toml, part, nomem);
        if (nomem) {
            if (wave == 1) throw std::runtime_error(std::string("msm_create failed: ") + msm_last_error(nullptr));
            if (args.synthesize || !args.comm_file.empty())
                throw std::runtime_error("the device cannot hold this process's streams at once, which --synthesize and --comm-file need: use more ranks");
            wave = (wave + 1) / 2;
            std::printf("device memory holds fewer than %zu streams of this size: running in waves of %zu\n", w1 - w0, wave);
            continue;
        }
        if (rc != 0) return rc;
        w0 = w1;
    }
    if (!mean_field_after.empty()) {
        Args alone = args;
        alone.comm_file.clear();
        alone.mean_density = alone.synthesize = false;
        bool nomem = false;
        const int rc = run_wave<T>(alone, toml, mean_field_after, nomem);
        if (nomem) throw std::runtime_error(std::string("msm_create failed: ") + msm_last_error(nullptr));
        if (rc != 0) return rc;
    }
    if (multi_stream)
        std::printf("Finished all streams in %lld seconds\n",
                    (long long)std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - t_start).count());
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    const Args args = parse_args(argc, argv);
    AbortWatch watch(args);
    try {
        const int rc = args.precision == MSM_F64 ? run<double>(args) : run<float>(args);
        watch.stop();
        return rc;
    } catch (const std::exception& e) {
        watch.stop();
        std::fprintf(stderr, "Error: %s\n", e.what());
        leave_abort_note(args, e.what());   // the other ranks' watchdogs end them: none stays behind in a collective
        return 1;
    }
}
