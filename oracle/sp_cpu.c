/* sp_cpu.c -- plain C restatement (OpenMP) of the reference's static-box
 * split-step update, fp64, dims = 3.
 *
 * TEST INFRASTRUCTURE ONLY: a second, independent CPU statement of the
 * algorithm (its own FFT, no numpy) that (a) cross-checks the numpy oracle and
 * the HIP path in tests and (b) is the timed "CPU port" baseline of bench.py.
 * Nothing in the product path links or loads it.
 *
 * Follows, by reference file:line (paths relative to the reference checkout):
 *   get_kgrid / spec_grid          simulator/src/utils/fft.rs:100-161
 *   forward / inverse (unitary)    simulator/src/utils/fft.rs:6-98
 *   calculate_density/potential    simulator/src/simulation_object.rs:1031-1110
 *   get_timestep                   simulator/src/simulation_object.rs:878-934
 *   update                         simulator/src/simulation_object.rs:475-661
 *   check_alias                    simulator/src/simulation_object.rs:1249-1293
 * Pinned by tests/test_cpu_port.py against the golden vectors and the numpy
 * oracle (which is itself pinned to the reference's known answers).
 *
 * Build: make -C oracle   ->  oracle/libsp_cpu.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define POIS_CONST (4.0 * M_PI * 4.49e-12) /* common/src/constants.rs:2 */
#define BLK 8                               /* neighbouring lines gathered together on strided axes */

typedef struct { double re, im; } cplx;

typedef struct sp_sim {
    int N;
    size_t n3;
    double L, hbar, M, cfl, Tf, k2_cutoff, alias_thr, dx, k2max, fourpi2;
    int ndumps, dumps;
    long nsteps;
    double time;
    cplx *psi, *psik, *work;
    double *phi, *ksq;
    cplx *tw; /* exp(-2 pi i p / N) */
    int *rev;
} sp_sim;

/* ---- 1-D radix-2 FFT on BLK interleaved lines: a[i*BLK + b] ---------------- */
static void fft_lines(const sp_sim *s, cplx *a, int sign) {
    const int N = s->N;
    for (int i = 0; i < N; ++i) {
        const int j = s->rev[i];
        if (j > i)
            for (int b = 0; b < BLK; ++b) { cplx t = a[i * BLK + b]; a[i * BLK + b] = a[j * BLK + b]; a[j * BLK + b] = t; }
    }
    for (int len = 2; len <= N; len <<= 1) {
        const int half = len >> 1, step = N / len;
        for (int i0 = 0; i0 < N; i0 += len)
            for (int k = 0; k < half; ++k) {
                const double wr = s->tw[k * step].re, wi = sign < 0 ? s->tw[k * step].im : -s->tw[k * step].im;
                cplx *u = a + (size_t)(i0 + k) * BLK, *v = a + (size_t)(i0 + k + half) * BLK;
                for (int b = 0; b < BLK; ++b) {
                    const double tr = v[b].re * wr - v[b].im * wi, ti = v[b].re * wi + v[b].im * wr;
                    v[b].re = u[b].re - tr; v[b].im = u[b].im - ti;
                    u[b].re += tr; u[b].im += ti;
                }
            }
    }
}

/* unitary 3-D transform in place: unnormalised passes, then scale 1/N^{3/2} (utils/fft.rs:17,44) */
static void fft3(const sp_sim *s, cplx *g, int sign) {
    const int N = s->N;
    const size_t N2 = (size_t)N * N;
    const size_t strides[3] = {1, (size_t)N, N2};
    for (int ax = 0; ax < 3; ++ax) {
        const size_t st = strides[ax];
        /* lines are indexed by the two other coordinates (o1 fastest); gather BLK of them along o1 */
        const size_t s1 = ax == 0 ? N : 1, s2 = ax == 2 ? N : N2;
#pragma omp parallel
        {
            cplx *buf = (cplx *)malloc(sizeof(cplx) * (size_t)N * BLK);
#pragma omp for collapse(2) schedule(static)
            for (int o2 = 0; o2 < N; ++o2)
                for (int o1 = 0; o1 < N; o1 += BLK) {
                    const size_t base = (size_t)o2 * s2 + (size_t)o1 * s1;
                    for (int i = 0; i < N; ++i)
                        for (int b = 0; b < BLK; ++b) buf[i * BLK + b] = g[base + (size_t)b * s1 + (size_t)i * st];
                    fft_lines(s, buf, sign);
                    for (int i = 0; i < N; ++i)
                        for (int b = 0; b < BLK; ++b) g[base + (size_t)b * s1 + (size_t)i * st] = buf[i * BLK + b];
                }
            free(buf);
        }
    }
    const double sc = 1.0 / pow((double)N, 1.5);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < s->n3; ++i) { g[i].re *= sc; g[i].im *= sc; }
}

static inline double k2_at(const sp_sim *s, size_t idx) {
    const int N = s->N;
    const int x = (int)(idx % N), y = (int)((idx / N) % N), z = (int)(idx / ((size_t)N * N));
    return ((s->ksq[x] + s->ksq[y]) + s->ksq[z]) * s->fourpi2; /* spec_grid association, utils/fft.rs:140-160 */
}

sp_sim *sp_create(int N, double L, double hbar, double M, double cfl, double Tf, int ndumps, double k2_cutoff,
                  double alias_thr, double time0) {
    if (N < 8 || (N & (N - 1)) || N % BLK) return NULL;
    sp_sim *s = (sp_sim *)calloc(1, sizeof(sp_sim));
    s->N = N; s->n3 = (size_t)N * N * N;
    s->L = L; s->hbar = hbar; s->M = M; s->cfl = cfl; s->Tf = Tf; s->ndumps = ndumps;
    s->k2_cutoff = k2_cutoff; s->alias_thr = alias_thr; s->time = time0;
    s->dx = L / (double)N;
    s->psi = (cplx *)malloc(sizeof(cplx) * s->n3);
    s->psik = (cplx *)malloc(sizeof(cplx) * s->n3);
    s->work = (cplx *)malloc(sizeof(cplx) * s->n3);
    s->phi = (double *)malloc(sizeof(double) * s->n3);
    s->ksq = (double *)malloc(sizeof(double) * N);
    s->tw = (cplx *)malloc(sizeof(cplx) * N);
    s->rev = (int *)malloc(sizeof(int) * N);
    for (int i = 0; i < N; ++i) {
        const int ii = i < N / 2 ? i : i - N;
        const double k = (double)ii / ((double)N * s->dx);
        s->ksq[i] = k * k;
        s->tw[i].re = (double)cosl(-2.0L * M_PIl * i / N);
        s->tw[i].im = (double)sinl(-2.0L * M_PIl * i / N);
        int r = 0;
        for (int b = 1, v = i; b < N; b <<= 1, v >>= 1) r = (r << 1) | (v & 1);
        s->rev[i] = r;
    }
    s->fourpi2 = pow(2.0 * M_PI, 2.0);
    const double km = s->ksq[N / 2];
    s->k2max = ((km + km) + km) * s->fourpi2;
    return s;
}

void sp_destroy(sp_sim *s) {
    if (!s) return;
    free(s->psi); free(s->psik); free(s->work); free(s->phi); free(s->ksq); free(s->tw); free(s->rev); free(s);
}

void sp_set_psi(sp_sim *s, const double *re, const double *im) {
    for (size_t i = 0; i < s->n3; ++i) { s->psi[i].re = re[i]; s->psi[i].im = im[i]; }
    s->nsteps = 0;
}
void sp_get_psi(const sp_sim *s, double *re, double *im) {
    for (size_t i = 0; i < s->n3; ++i) { re[i] = s->psi[i].re; im[i] = s->psi[i].im; }
}
void sp_get_psik(const sp_sim *s, double *re, double *im) {
    for (size_t i = 0; i < s->n3; ++i) { re[i] = s->psik[i].re; im[i] = s->psik[i].im; }
}
void sp_get_phi(const sp_sim *s, double *phi) { memcpy(phi, s->phi, sizeof(double) * s->n3); }
void sp_fft(sp_sim *s, int sign) { fft3(s, s->psi, sign); }

/* calculate_potential; returns max |phi| */
double sp_potential(sp_sim *s) {
    cplx *w = s->work;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < s->n3; ++i) {
        w[i].re = s->M * (s->psi[i].re * s->psi[i].re + s->psi[i].im * s->psi[i].im);
        w[i].im = 0.0;
    }
    fft3(s, w, -1);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < s->n3; ++i) {
        const double k2 = k2_at(s, i);
        if (k2 == 0.0) { w[i].re = 0.0; w[i].im = 0.0; }           /* NaN -> 0, :1097-1102 */
        else { w[i].re = (-POIS_CONST * w[i].re) / k2; w[i].im = (-POIS_CONST * w[i].im) / k2; }
    }
    fft3(s, w, +1);
    double mx = 0.0;
#pragma omp parallel for schedule(static) reduction(max : mx)
    for (size_t i = 0; i < s->n3; ++i) {
        s->phi[i] = w[i].re;
        const double a = fabs(w[i].re);
        if (a > mx) mx = a;
    }
    return mx;
}

static void drift(const sp_sim *s, double c) { /* psi_k *= exp(i c k^2), :504-516 */
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < s->n3; ++i) {
        double sn, cs;
        sincos(c * k2_at(s, i), &sn, &cs);
        const double r = s->psik[i].re * cs - s->psik[i].im * sn, m = s->psik[i].re * sn + s->psik[i].im * cs;
        s->psik[i].re = r; s->psik[i].im = m;
    }
}

/* one update; out = {dt, dump, alias_mass, max_phi, time}; returns 1 if aliased */
int sp_update(sp_sim *s, double out[5]) {
    const size_t n3 = s->n3;
    if (s->nsteps == 0) { memcpy(s->psik, s->psi, sizeof(cplx) * n3); fft3(s, s->psik, -1); } /* :477-479 */
    const double pmax = sp_potential(s);                                                      /* :497 */
    /* get_timestep :878-934 */
    const double kinetic = s->cfl * 2.0 * s->L / sqrt(s->k2max) / s->hbar;
    const double potential = s->cfl * (2.0 * M_PI) * s->hbar / (2.0 * pmax);
    const double ttnd = (double)(s->dumps + 1) * s->Tf / (double)s->ndumps - s->time;
    const double dt = fmin(fmin(kinetic, potential), ttnd);
    const int dump = dt == ttnd;
    const double ck = -dt / 4.0 * s->hbar;
    drift(s, ck);
    memcpy(s->psi, s->psik, sizeof(cplx) * n3); fft3(s, s->psi, +1);                          /* :523 */
    sp_potential(s);                                                                           /* :530 */
    const double cr = -dt / s->hbar;                                                           /* :535-545 */
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n3; ++i) {
        double sn, cs;
        sincos(cr * s->phi[i], &sn, &cs);
        const double r = s->psi[i].re * cs - s->psi[i].im * sn, m = s->psi[i].re * sn + s->psi[i].im * cs;
        s->psi[i].re = r; s->psi[i].im = m;
    }
    memcpy(s->psik, s->psi, sizeof(cplx) * n3); fft3(s, s->psik, -1);                          /* :552 */
    drift(s, ck);                                                                              /* :562-574 */
    memcpy(s->psi, s->psik, sizeof(cplx) * n3); fft3(s, s->psi, +1);                           /* :581 */
    s->time = s->time + dt;                                                                    /* :590 */
    /* check_alias :1249-1293 */
    const double cut = s->k2max * s->k2_cutoff;
    double sum = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : sum)
    for (size_t i = 0; i < n3; ++i)
        if (k2_at(s, i) > cut) sum += s->psik[i].re * s->psik[i].re + s->psik[i].im * s->psik[i].im;
    const double p_mass = sum * pow(s->dx, 3.0);
    const int aliased = p_mass > s->alias_thr;
    if (!aliased && dump) {                                                                    /* :620-631 */
        s->dumps += 1;
        s->time = (double)s->dumps * s->Tf / (double)s->ndumps;
    }
    s->nsteps += 1;
    out[0] = dt; out[1] = (double)dump; out[2] = p_mass; out[3] = pmax; out[4] = s->time;
    return aliased;
}

int sp_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
