// The small-halo class: ONE WAVEFRONT per halo, the halo's slab resident in LDS from the selection to the last property.
//
// Nine halos in ten of a chunk hold a few hundred particles in their final search sphere (C3: 57 % below 128, 77 % below
// 256, 87 % below 512), and what they cost in the thread-group classes of halo_kernel.inc is not arithmetic but the
// ~100 dependent round trips to a slab in global memory that one wavefront with 256 registers (two per SIMD) waits for.
// Here the whole per-halo state is SNS slab entries in LDS (81 bytes each) plus registers:
//
//   select     candidate cells exactly as shared_mesh.py:146-179, rows streamed with SEL_UT tiles in flight; the test is the
//              minimum-image r^2 <= R^2 of shared_mesh.py:183-194; only the packed index of a selected particle is kept
//   gather     one payload record per selected particle (all loads of the wave in flight together): wrapped offsets
//              (halo_tasks.py:113-117), radius, minimum-image r^2, velocity, mass, membership bits
//   sort       (r, index) bitonic network in REGISTERS (entry i of the slab belongs to lane i mod 64), then the columns are
//              permuted into sorted order through LDS
//   cumsum     the sequential fp64 chain of numpy.cumsum (SO_properties.py:400) by one lane over LDS
//   radius loop / stage A    halo_tasks.py:74-187 with lanes = calculations, the same decisions as k_halo of halo_kernel.inc
//              (shared code: halo_so_eval.inc); the spheres of the rounds are counted with ballots over the r^2 registers
//   stage B    prefix moments from one inclusive wave scan per block of 64 entries, the radial scan (v_max, half-mass radii),
//              spin, dark matter flow rates; finalisation with lanes = calculations (shared code: halo_finalize.inc)
//
// No scratch, no slab traffic to HBM: what the kernel reads is 24 B per candidate and 80 B per selected particle, what it
// writes is the halo's row of the result table.  A halo whose sphere outgrows SNS entries -- or that needs something this
// kernel leaves to the general path (radius ties, neutrinos, inertia tensors, ...) -- comes back with
// SOAP_HALO_FATAL_SLAB_OVERFLOW and is done by the wavefront class of halo_kernel.inc (engine.hip escalates it).
//
// Compiled once per variant: -DSNS=<slab entries> -DSHYDRO=<0: dark matter only chunk, 1: gas / stars / black holes too>.
#include "engine_common.h"

#ifndef SNS
#define SNS 128
#endif
#ifndef SHYDRO
#define SHYDRO 0
#endif
#define SE (SNS / 64)          // slab entries per lane: entry i belongs to lane i & 63, register slot i >> 6
#ifndef SGLOBAL
#define SGLOBAL 0              // 1: the columns that only the property passes stream (mass, offsets, velocities: 56 of the 81 bytes of an
#endif                         // entry) live in a slice of global memory that belongs to the wave, the slab of a 512-entry halo takes 11 KB of LDS
#define SPREFB 12              // calculations whose prefix moments are finalised together
#define SMAXPJ 8               // projected apertures per run in the hydro variants (engine.hip keeps longer lists on the general path)
#if SHYDRO
#define SMOM A_NSUM
#define SNHM NHM
#else
#define SMOM (A_RMAX + 1)
#define SNHM 2
#endif
#ifndef SEL_UT
#define SEL_UT 4               // candidate tiles in flight per trip of the selection
#endif
#ifndef SWPE
#define SWPE 3
#endif

#define CAS __attribute__((address_space(4)))   // constant memory: uniform addresses become scalar loads

#define SOAP_CAT4_(a, b, c, d) a##b##c##d
#define SOAP_CAT4(a, b, c, d) SOAP_CAT4_(a, b, c, d)
#ifndef SINERTIA
#define SINERTIA 0             // 1: the inertia tensor pass compiled in (its LDS and registers stay out of the variant every other run uses)
#endif
#if SHYDRO
#define SVAR h
#elif SINERTIA
#define SVAR i
#else
#define SVAR d
#endif

namespace {


struct SmallWave {
    // the slab, sorted by radius once it is built
    double r[SNS], cum[SNS];
#if !SGLOBAL
    double m[SNS], dx[SNS], dy[SNS], dz[SNS], vx[SNS], vy[SNS], vz[SNS];
#else
    GAS double *cols;             // this wave's slice of RunParams.small_cols: seven columns of SNS doubles
#endif
    unsigned pidx[SNS];
    unsigned char flags[SNS];
    // stage B
    double prefc[SPREFB][SMOM];   // prefix moments of up to SPREFB calculations
    int bnd[MAXCALC];             // prefix lengths of a family's calculations in ascending order
#if SINERTIA
    double it_par[64][12];        // inertia tensors: eigenvectors (9) and reciprocal semi-axes (3) of the 64 states, read by every lane in the sweeps
#endif
#if SHYDRO
    double rot_v[MAXCALC][6];     // centre-of-mass velocities of gas (0..2) and stars (3..5) per calculation, slab frame (parked by finalize_moments for the rotation pass)
    double pj[SMAXPJ][24];        // prefix sums of the projected apertures of one axis, in order of radius
#endif
    // per-halo context, named as in halo_kernel.inc (the shared includes read them through g_S)
    double c[3], L, shift[3];
    double nu_density, a, mpc, mpc20, G_eff;
    int hydro, has_nu, mass_f32;
    unsigned ticket;   // queue position of the halo the wave is working on
#ifdef SMALL_CHECK
    unsigned long long *dbg;
#endif
};

__shared__ SmallWave g_W;
#ifdef SMALL_CHECK
// diagnostic build: a violated bound is reported through P.phase[16 + code] (largest offending value + 1) and the access skipped
#define SCHECK(cond, code, val) ((cond) ? true : (atomicMax((unsigned long long *)&g_W.dbg[16 + (code)], (unsigned long long)(val) + 1ull), false))
#else
#define SCHECK(cond, code, val) true
#endif
#define g_S g_W
#define LANE ((int)(threadIdx.x & 63))

// the streamed columns of the slab
#if SGLOBAL
enum { SC_m = 0, SC_dx, SC_dy, SC_dz, SC_vx, SC_vy, SC_vz, SC_NCOLS };
#define CX(name) (small_cols_base() + SC_##name * SNS)
#else
#define CX(name) g_W.name
#endif

// LDS traffic between the lanes of ONE wave: its LDS instructions execute in order, so only the compiler has to be kept
// from moving accesses across the hand-over (no s_waitcnt on outstanding global loads, unlike a workgroup-scope fence)
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// Cross-lane reads.  v_readlane / v_readfirstlane of a CONSTANT fold to that constant -- rightly so when every lane holds it,
// but the compiler also applies the fold along single edges of a lane-dependent branch ("in the lanes that skipped
// `if (LANE == 0) x = ...`, x is 0, hence readfirstlane(x) is 0") and then threads later uniform branches per lane: the
// queue loop of this kernel was compiled into an exit that only lane 0 took.  Every value read across lanes therefore
// passes an empty asm first, after which nothing is known about it; values that exist in one lane only are better not
// created at all (LDS hand-over, or the same arithmetic in every lane).  And: these reads belong in code that EVERY lane
// executes -- inside a lane-dependent branch the asm copies the value in that branch's lanes only.
SOAP_DEV int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
SOAP_DEV int rl(int v, int lane) { return __builtin_amdgcn_readlane(opaque(v), lane); }
SOAP_DEV double rl64(double v, int lane) {
    const int lo = rl(__double2loint(v), lane), hi = rl(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
SOAP_DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(opaque(v)); }   // a value that is the same in every lane, made known to the compiler as such
SOAP_DEV double uni(double v) {
    const int lo = uni(__double2loint(v)), hi = uni(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
SOAP_DEV long long uni(long long v) {
    const unsigned lo = (unsigned)uni((int)(unsigned)v), hi = (unsigned)uni((int)(unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

#if SGLOBAL
SOAP_DEV GAS double *small_cols_base() {   // (uniform: an SGPR pair)
    const unsigned long long v = (unsigned long long)g_W.cols;
    const unsigned lo = (unsigned)uni((int)(unsigned)v), hi = (unsigned)uni((int)(unsigned)(v >> 32));
    return (GAS double *)(((unsigned long long)hi << 32) | lo);
}
#endif

// enclosed density of slab entry i as materialize() of halo_kernel.inc tabulates it (SO_properties.py:404-406, :420)
SOAP_DEV double small_dens(int i) {
    const double nuK = ((g_S.nu_density * 4.0) / 3.0) * M_PI;
    const double K43 = (4.0 / 3.0) * M_PI;
    const double r3 = cube(g_W.r[i]);
    // (with float32 masses: the reference's casts of the cumulative mass, so_cum_nu of halo_so_eval.inc)
    const double c = g_W.mass_f32 ? (double)(float)g_W.cum[i] : g_W.cum[i];
    const double cn = (g_W.mass_f32 && nuK != 0.0) ? (double)(float)(c + nuK * r3) : c + nuK * r3;
    return cn / (K43 * r3);
}
#define SV_R(i) g_W.r[i]
#define SV_CUM(i) so_cum32(g_W.cum[i])
#define SV_DENS(i) small_dens(i)
#define RES_OK 0
#define RES_TOO_SMALL 1
#include "halo_so_eval.inc"
#if SHYDRO
#define FZ_PARK_ROT(c, d, vg, vs) { g_W.rot_v[c][d] = (vg); g_W.rot_v[c][3 + (d)] = (vs); }
#else
#define FZ_PARK_ROT(c, d, vg, vs) { }   // (no gas or stars in this variant)
#endif
// log10 / 10**x of the concentration as real calls: inlined, the two library routines hold ~100 registers at their peak and push
// the kernel's long-lived values into scratch
#ifdef SMALL_INLINE_MATH
__device__ __forceinline__ double small_log10(double x) { return log10(x); }
__device__ __forceinline__ double small_exp10(double x) { return exp10(x); }
#else
__device__ __attribute__((noinline)) double small_log10(double x) { return log10(x); }
__device__ __attribute__((noinline)) double small_exp10(double x) { return exp10(x); }
#endif
#define FZ_LOG10(x) small_log10(x)
#define FZ_EXP10(x) small_exp10(x)
#define FZ_MOM const volatile double *   // (the moments are read from LDS where they are used: hoisted, they would occupy 80 registers)
#include "halo_finalize.inc"

// ---- 256-bit cell masks: runs of set bits without a loop over the cells ------------------------------------------------
SOAP_DEV unsigned long long mask_word(const Mask256 &m, int k) { return k == 0 ? m.w0 : (k == 1 ? m.w1 : (k == 2 ? m.w2 : m.w3)); }
// first set (want = 1) or clear (want = 0) bit at position >= from, or 256
SOAP_DEV int mask_next(const Mask256 &m, int from, int want) {
    for (int k = from >> 6; k < 4; k++) {
        unsigned long long w = mask_word(m, k);
        if (!want) w = ~w;
        if (k == (from >> 6)) { const int sh = from & 63; w = (w >> sh) << sh; }
        if (w) return 64 * k + __ffsll((long long)w) - 1;
    }
    return 256;
}

// candidate cells of type `T` along the three axes for a sphere of radius R about the halo centre (shared_mesh.py:146-179):
// lane d < 3 works on axis d, every lane gets all three masks
SOAP_DEV void cell_masks(const CAS TypeDev &T, double R, Mask256 &mx, Mask256 &my, Mask256 &mz, bool *nowrap = nullptr) {
    const int d = LANE < 2 ? LANE : 2;
    const double pmin = d == 0 ? T.pos_min[0] : (d == 1 ? T.pos_min[1] : T.pos_min[2]);
    const double pmax = d == 0 ? T.pos_max[0] : (d == 1 ? T.pos_max[1] : T.pos_max[2]);
    const double csz = d == 0 ? T.cell_size[0] : (d == 1 ? T.cell_size[1] : T.cell_size[2]);
    const double c = d == 0 ? g_W.c[0] : (d == 1 ? g_W.c[1] : g_W.c[2]), L = g_W.L;
    const int res = T.res;
    Mask256 mk;
    int min_copy = 0, max_copy = 0, guard = 0;
    while (c + (double)(min_copy - 1) * L + R >= pmin && guard++ < 4096) min_copy--;
    guard = 0;
    while (c + (double)(max_copy + 1) * L - R <= pmax && guard++ < 4096) max_copy++;
    for (int cp = min_copy; cp <= max_copy; cp++) {
        const double lo = fmax(pmin, c + (double)cp * L - R);
        const double hi = fmin(pmax, c + (double)cp * L + R);
        const double qlo = floor((lo - pmin) / csz);
        const double qhi = floor((hi - pmin) / csz);
        const int ilo = qlo < 0.0 ? 0 : (qlo > 1e9 ? 1000000000 : (int)qlo);
        const int ihi = qhi < -1.0 ? -1 : (qhi > (double)(res - 1) ? res - 1 : (int)qhi);
        if (ilo <= ihi) mk.set_range(ilo, ihi);
    }
    // no periodic copy of the sphere reaches the mesh along any axis and every candidate lies within L/2 of the centre (R + one cell):
    // the minimum-image separation of the mesh query is then x - c itself (shared_mesh.py:138-142 changes nothing)
    if (nowrap) *nowrap = __ballot(LANE < 3 && (min_copy != 0 || max_copy != 0 || !(R + 1.5 * csz <= 0.5 * L))) == 0ull;
#define BC64(w, l) ((unsigned long long)(unsigned)rl((int)(unsigned)(w), l) | ((unsigned long long)(unsigned)rl((int)(unsigned)((w) >> 32), l) << 32))
    mx.w0 = BC64(mk.w0, 0); mx.w1 = BC64(mk.w1, 0); mx.w2 = BC64(mk.w2, 0); mx.w3 = BC64(mk.w3, 0);
    my.w0 = BC64(mk.w0, 1); my.w1 = BC64(mk.w1, 1); my.w2 = BC64(mk.w2, 1); my.w3 = BC64(mk.w3, 1);
    mz.w0 = BC64(mk.w0, 2); mz.w1 = BC64(mk.w1, 2); mz.w2 = BC64(mk.w2, 2); mz.w3 = BC64(mk.w3, 2);
#undef BC64
}

#define SMAXRUN 6
// runs of set bits of the x mask (one per periodic copy of the sphere), 8 bits per run in two words; beyond SMAXRUN the
// last run swallows the rest (a superset of the cells is still exact: the r^2 test decides)
SOAP_DEV int small_x_runs(const Mask256 &mx, int res, unsigned long long &lo_pack, unsigned long long &hi_pack) {
    int nrun = 0;
    lo_pack = 0ull; hi_pack = 0ull;
    int i0 = mask_next(mx, 0, 1);
    while (i0 < res && nrun < SMAXRUN) {
        int i1 = mask_next(mx, i0, 0) - 1;
        if (i1 > res - 1) i1 = res - 1;
        if (nrun == SMAXRUN - 1) {
            for (int q = res - 1; q > i1; q--) if (mx.get(q)) { i1 = q; break; }
        }
        lo_pack |= (unsigned long long)i0 << (8 * nrun);
        hi_pack |= (unsigned long long)i1 << (8 * nrun);
        nrun++;
        i0 = mask_next(mx, i1 + 1, 1);
    }
    return nrun;
}
SOAP_DEV int pick_run(unsigned long long pack, int r) { return (int)((pack >> (8 * r)) & 0xffull); }

// Particles with minimum-image r^2 <= R^2 (shared_mesh.py:183-194): their packed indices to g_W.pidx[0, n).  Returns n,
// or -1 as soon as the sphere is known not to fit the slab.
__device__ __forceinline__ int small_select(const CAS RunParams &P, double R) {
    const double R2 = R * R;
    const double cx = g_W.c[0], cy = g_W.c[1], cz = g_W.c[2], L = g_W.L;
    int nsel = 0;
#pragma unroll 1
    for (int ts = 0; ts < SOAP_NTYPES; ts++) {
        const CAS TypeDev &T = P.t[ts];
        if (T.n == 0) continue;
        const int res = T.res;
        Mask256 mx, my, mz;
        bool nowrap;
        cell_masks(T, R, mx, my, mz, &nowrap);
        unsigned long long run_lo, run_hi;
        const int nrun = small_x_runs(mx, res, run_lo, run_hi);
        const int nk = mz.count(), nj = my.count();
        const int ntrip = nk * nj * nrun;
        const GAS double *tx = (const GAS double *)T.x, *ty = (const GAS double *)T.y, *tz = (const GAS double *)T.z;
        const GAS unsigned *toff = (const GAS unsigned *)T.cell_offset;
#pragma unroll 1
        for (int t0 = 0; t0 < ntrip; t0 += 64) {
            unsigned mybeg = 0, myend = 0;
            {
                const int tl = t0 + LANE;
                if (tl < ntrip) {
                    const int rr = tl % nrun, pj = (tl / nrun) % nj, pk = tl / (nrun * nj);
                    const int k = mz.nth(pk), j = my.nth(pj);
                    const long long rowbase = (long long)res * j + (long long)res * res * k;
                    if (SCHECK(k >= 0 && k < res && j >= 0 && j < res && pick_run(run_hi, rr) < res && pick_run(run_lo, rr) <= pick_run(run_hi, rr), 1, rowbase + pick_run(run_hi, rr))) {
                    mybeg = toff[rowbase + pick_run(run_lo, rr)];
                    myend = toff[rowbase + pick_run(run_hi, rr) + 1];
                    }
                    (void)SCHECK((long long)myend <= T.n && mybeg <= myend, 2, myend);
                    if ((long long)myend > T.n || mybeg > myend) { mybeg = 0; myend = 0; }
                }
            }
            const int cnt = ntrip - t0 < 64 ? ntrip - t0 : 64;
#pragma unroll 1
            for (int q = 0; q < cnt; q++) {
                const unsigned beg = (unsigned)rl((int)mybeg, q);
                const unsigned end = (unsigned)rl((int)myend, q);
#pragma unroll 1
                for (unsigned pb = beg; pb < end; pb += SEL_UT * 64) {
                    unsigned pp[SEL_UT];
                    double xx[SEL_UT], yy[SEL_UT], zz[SEL_UT];
#pragma unroll
                    for (int u = 0; u < SEL_UT; u++) {
                        pp[u] = pb + u * 64 + LANE;
                        xx[u] = yy[u] = zz[u] = 0.0;
                        if (pp[u] < end) { xx[u] = tx[pp[u]]; yy[u] = ty[pp[u]]; zz[u] = tz[pp[u]]; }
                    }
#pragma unroll
                    for (int u = 0; u < SEL_UT; u++) {
                        bool sel = false;
                        if (pp[u] < end) {
                            double ax, ay, az;
                            if (nowrap) { ax = xx[u] - cx; ay = yy[u] - cy; az = zz[u] - cz; }
                            else { ax = min_image(xx[u], cx, L); ay = min_image(yy[u], cy, L); az = min_image(zz[u], cz, L); }
                            const double r2 = (ax * ax + ay * ay) + az * az;
                            sel = r2 <= R2;
                        }
                        const unsigned long long b = __ballot(sel);
                        if (b) {
                            const int slot = nsel + __popcll(b & ((1ull << LANE) - 1ull));
                            if (sel && slot < SNS) g_W.pidx[slot] = ((unsigned)ts << PIDX_BITS) | pp[u];
                            nsel += __popcll(b);
                        }
                    }
                    if (nsel > SNS) return -1;
                }
            }
        }
    }
    return nsel;
}

// number of particles in the mesh cells a sphere of radius R overlaps (n_cand of the particle-aperture-evaluation
// accounting); no particle data is touched
__device__ __forceinline__ long long small_count_candidates(const CAS RunParams &P, double R) {
    double tot = 0.0;
#pragma unroll 1
    for (int ts = 0; ts < NT_ALL; ts++) {
        const CAS TypeDev &T = P.t[ts];
        if (T.n == 0) continue;
        const int res = T.res;
        Mask256 mx, my, mz;
        cell_masks(T, R, mx, my, mz);
        unsigned long long run_lo, run_hi;
        const int nrun = small_x_runs(mx, res, run_lo, run_hi);
        const int nk = mz.count(), nj = my.count();
        const int ntrip = nk * nj * nrun;
        const GAS unsigned *toff = (const GAS unsigned *)T.cell_offset;
        for (int tl = LANE; tl < ntrip; tl += 64) {
            const int rr = tl % nrun, pj = (tl / nrun) % nj, pk = tl / (nrun * nj);
            const int k = mz.nth(pk), j = my.nth(pj);
            const long long rowbase = (long long)res * j + (long long)res * res * k;
            tot += (double)(toff[rowbase + pick_run(run_hi, rr) + 1] - toff[rowbase + pick_run(run_lo, rr)]);
        }
    }
    return (long long)wave_sum(tot);   // (integers below 2^53: exact)
}

// ---- (key, index) sort in registers ------------------------------------------------------------------------------------
// Element i = 64 e + lane of a bitonic network over npad = 2^lg elements: key k[e] (radius), v[e] (packed particle index,
// the tie-break: pairs are distinct) and s[e] (a payload that travels along).  Stages whose partner lies in another lane
// exchange through the cross-lane network (ds_bpermute: no LDS memory involved), the others stay in the lane's registers.
SOAP_DEV void sort_exchange(double &k, unsigned &v, unsigned &s, int j, bool want_min) {
    const int lo = __double2loint(k), hi = __double2hiint(k);
    const int plo = __shfl_xor(lo, j, 64), phi = __shfl_xor(hi, j, 64);
    const unsigned pv = (unsigned)__shfl_xor((int)v, j, 64), ps = (unsigned)__shfl_xor((int)s, j, 64);
    const double pk = __hiloint2double(phi, plo);
    const bool partner_less = pk < k || (pk == k && pv < v);
    const bool take = want_min ? partner_less : !partner_less;
    // (equal pairs are padding: taking or keeping is the same)
    if (take && !(pk == k && pv == v)) { k = pk; v = pv; s = ps; }
}
__device__ __forceinline__ void small_sort(double (&k)[SE], unsigned (&v)[SE], unsigned (&s)[SE], int lg) {
    const int ne = lg <= 6 ? 1 : (1 << (lg - 6));   // register slots that hold elements below npad
#pragma unroll 1
    for (int p = 1; p <= lg; p++) {
        const int kk = 1 << p;
#pragma unroll 1
        for (int jb = p - 1; jb >= 0; jb--) {
            const int j = 1 << jb;
            if (jb < 6) {
#pragma unroll
                for (int e = 0; e < SE; e++) {
                    if (e >= ne) continue;
                    const int i = 64 * e + LANE;
                    const bool asc = (i & kk) == 0, lower = (i & j) == 0;
                    sort_exchange(k[e], v[e], s[e], j, lower == asc);
                }
            } else {
                const int je = j >> 6;
#pragma unroll
                for (int e = 0; e < SE; e++) {
#pragma unroll
                    for (int f = 0; f < SE; f++) {
                        // pairs (e, f = e ^ je) with e < f, both below ne
                        if (f <= e || (e ^ f) != je || f >= ne) continue;
                        const int i = 64 * e + LANE;
                        const bool asc = (i & kk) == 0;
                        const bool gt = k[e] > k[f] || (k[e] == k[f] && v[e] > v[f]);
                        if (gt == asc) {
                            const double tk = k[e]; k[e] = k[f]; k[f] = tk;
                            const unsigned tv = v[e]; v[e] = v[f]; v[f] = tv;
                            const unsigned tsv = s[e]; s[e] = s[f]; s[f] = tsv;
                        }
                    }
                }
            }
        }
    }
}

// ---- build: the sorted superset of everything within R, in LDS ----------------------------------------------------------
// Returns the number of entries (>= 0), -1 if the sphere does not fit the slab.  q[e] receives the minimum-image r^2 of the
// lane's entries (for the spheres of the radius rounds); g_W.shift / *fof0 the velocity and the FOF id of the innermost
// particle (SO_properties.py:407-409).
#if SGLOBAL
// The variant with the streamed columns in global memory: the selected particles' coordinates are read once more from the
// cell-ordered arrays (the lines the selection has just streamed), the (radius, index) pairs are sorted in registers, and the
// payload records are then gathered IN SORTED ORDER, every column written once at its final place.
__device__ __forceinline__ int small_build(const CAS RunParams &P, double R, long long hindex, double (&q)[SE], long long *fof0) {
    const int n = small_select(P, R);
    if (n < 0) return -1;
    WSYNC();
    const double cx = g_W.c[0], cy = g_W.c[1], cz = g_W.c[2], L = g_W.L;
    double key[SE];
    unsigned pv[SE], sl[SE];
    // ---- radii of the selected particles
    {
#pragma unroll
        for (int e = 0; e < SE; e++) {
            const int i = 64 * e + LANE;
            key[e] = INFINITY; pv[e] = 0xffffffffu; sl[e] = 0u;
            if (i < n) pv[e] = g_W.pidx[i];
        }
        double xx[SE], yy[SE], zz[SE];
#pragma unroll
        for (int e = 0; e < SE; e++) {
            xx[e] = yy[e] = zz[e] = 0.0;
            if (64 * e + LANE < n) {
                const unsigned p = pv[e] & PIDX_MASK;
#if SHYDRO
                const int ts = (int)(pv[e] >> PIDX_BITS);
                const CAS TypeDev &T = P.t[ts];
#else
                const CAS TypeDev &T = P.t[1];
#endif
                xx[e] = ((const GAS double *)T.x)[p]; yy[e] = ((const GAS double *)T.y)[p]; zz[e] = ((const GAS double *)T.z)[p];
            }
        }
#pragma unroll
        for (int e = 0; e < SE; e++) {
            if (64 * e + LANE < n) {
                const double wx = wrap_rel(xx[e], cx, L), wy = wrap_rel(yy[e], cy, L), wz = wrap_rel(zz[e], cz, L);
                key[e] = sqrt((wx * wx + wy * wy) + wz * wz);
            }
        }
    }
    int lg = 0;
    while ((1 << lg) < n) lg++;
    small_sort(key, pv, sl, lg);
    WSYNC();
#pragma unroll
    for (int e = 0; e < SE; e++) if (64 * e + LANE < n) { g_W.r[64 * e + LANE] = key[e]; g_W.pidx[64 * e + LANE] = pv[e]; }
    WSYNC();
    const GAS Pay *pay0 = (const GAS Pay *)P.t[0].pay, *pay1 = (const GAS Pay *)P.t[1].pay, *pay2 = (const GAS Pay *)P.t[2].pay,
                  *pay3 = (const GAS Pay *)P.t[3].pay;
    // ---- the innermost particle (no neutrinos in this class: entry 0): FOF id, velocity shift -- every lane reads its record
    long long f0 = -1;
    double sh0 = 0.0, sh1 = 0.0, sh2 = 0.0;
    {
        const unsigned pi0 = n > 0 ? g_W.pidx[0] : 0u;
#if SHYDRO
        const int ts0 = (int)(pi0 >> PIDX_BITS);
        const GAS Pay *base0 = ts0 == 0 ? pay0 : (ts0 == 1 ? pay1 : (ts0 == 2 ? pay2 : pay3));
#else
        const GAS Pay *base0 = pay1;
#endif
        if (n > 0) {
            const Pay r0 = load_pay(base0 + (pi0 & PIDX_MASK));
            f0 = uni(r0.fof); sh0 = uni(r0.vx); sh1 = uni(r0.vy); sh2 = uni(r0.vz);
        }
    }
    *fof0 = f0;
    if (LANE == 0) { g_W.shift[0] = sh0; g_W.shift[1] = sh1; g_W.shift[2] = sh2; }
    // ---- the payload records in sorted order, two entries per lane in flight
    GAS double *c_m = CX(m), *c_dx = CX(dx), *c_dy = CX(dy), *c_dz = CX(dz), *c_vx = CX(vx), *c_vy = CX(vy), *c_vz = CX(vz);
#pragma unroll
    for (int e0 = 0; e0 < SE; e0 += 2) {
        Pay rec[2];
        int ts[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int e = e0 + u, i = 64 * e + LANE;
            ts[u] = 1;
            if (i < n) {
                ts[u] = (int)(pv[e] >> PIDX_BITS);
#if SHYDRO
                const GAS Pay *base = ts[u] == 0 ? pay0 : (ts[u] == 1 ? pay1 : (ts[u] == 2 ? pay2 : pay3));
#else
                const GAS Pay *base = pay1;
#endif
                rec[u] = load_pay(base + (pv[e] & PIDX_MASK));
            }
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int e = e0 + u, i = 64 * e + LANE;
            q[e] = INFINITY;
            if (i < n) {
                const Pay &R_ = rec[u];
                const double wx = wrap_rel(R_.x, cx, L), wy = wrap_rel(R_.y, cy, L), wz = wrap_rel(R_.z, cz, L);
                const double ax = min_image(R_.x, cx, L), ay = min_image(R_.y, cy, L), az = min_image(R_.z, cz, L);
                q[e] = (ax * ax + ay * ay) + az * az;
                c_dx[i] = wx; c_dy[i] = wy; c_dz[i] = wz;
                c_vx[i] = R_.vx - sh0; c_vy[i] = R_.vy - sh1; c_vz[i] = R_.vz - sh2;
                c_m[i] = R_.m;
                g_W.cum[i] = R_.m;   // (the cumulative-mass chain below runs in place)
                unsigned char fl = (unsigned char)ts[u];
                if (R_.grnr == hindex) fl |= FL_BOUND;
                if (R_.grnr >= 0 && R_.grnr != hindex) fl |= (R_.fof == f0) ? FL_SAT : FL_EXT;
                g_W.flags[i] = fl;
            }
        }
    }
    (void)pay0; (void)pay2; (void)pay3;
    // the columns are read back by other lanes of the wave (shell ranges of the flow rates): global memory, workgroup scope
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    WSYNC();
    // ---- cumulative mass in numpy.cumsum order: ONE sequential chain, in place over the masses parked in the column
    if (LANE == 0) {
        double run = 0.0;
        int j = 0;
        for (; j + 16 <= n; j += 16) {
            double a_[16];
#pragma unroll
            for (int u = 0; u < 16; u++) a_[u] = g_W.cum[j + u];
#pragma unroll
            for (int u = 0; u < 16; u++) { run += a_[u]; a_[u] = run; }
#pragma unroll
            for (int u = 0; u < 16; u++) g_W.cum[j + u] = a_[u];
        }
        for (; j < n; j++) { run += g_W.cum[j]; g_W.cum[j] = run; }
    }
    WSYNC();
    return n;
}
#else
__device__ __forceinline__ int small_build(const CAS RunParams &P, double R, long long hindex, double (&q)[SE], long long *fof0) {
#ifndef NO_SEL
    const int n = small_select(P, R);
#else
    const int n = 77;
#endif
    if (n < 0) return -1;
    WSYNC();
    const double cx = g_W.c[0], cy = g_W.c[1], cz = g_W.c[2], L = g_W.L;
    // the five payload pointers of the staged types (uniform: scalar registers)
    const GAS Pay *pay0 = (const GAS Pay *)P.t[0].pay, *pay1 = (const GAS Pay *)P.t[1].pay, *pay2 = (const GAS Pay *)P.t[2].pay,
                  *pay3 = (const GAS Pay *)P.t[3].pay;
    double key[SE];
    unsigned pv[SE], sl[SE];
    // ---- gather in selection order: entry 64 e + lane
    {
        Pay rec[SE];
        int ts[SE];
#pragma unroll
        for (int e = 0; e < SE; e++) {
            const int i = 64 * e + LANE;
            pv[e] = 0xffffffffu; ts[e] = 1;
            if (i < n) {
                pv[e] = g_W.pidx[i];
                ts[e] = (int)(pv[e] >> PIDX_BITS);
            }
        }
#pragma unroll
        for (int e = 0; e < SE; e++) {
            const int i = 64 * e + LANE;
            if (i < n) {
                const unsigned p = pv[e] & PIDX_MASK;
#if SHYDRO
                const GAS Pay *base = ts[e] == 0 ? pay0 : (ts[e] == 1 ? pay1 : (ts[e] == 2 ? pay2 : pay3));
#else
                const GAS Pay *base = pay1;
                (void)pay0; (void)pay2; (void)pay3;
#endif
                if (SCHECK((long long)p < P.t[1].n, 3, p)) rec[e] = load_pay(base + p);
            }
        }
#pragma unroll
        for (int e = 0; e < SE; e++) {
            const int i = 64 * e + LANE;
            key[e] = INFINITY; sl[e] = (unsigned)i; q[e] = INFINITY;
            if (i < n) {
                const Pay &R_ = rec[e];
                const double wx = wrap_rel(R_.x, cx, L), wy = wrap_rel(R_.y, cy, L), wz = wrap_rel(R_.z, cz, L);
                key[e] = sqrt((wx * wx + wy * wy) + wz * wz);
                const double ax = min_image(R_.x, cx, L), ay = min_image(R_.y, cy, L), az = min_image(R_.z, cz, L);
                g_W.dx[i] = wx; g_W.dy[i] = wy; g_W.dz[i] = wz;
                g_W.vx[i] = R_.vx; g_W.vy[i] = R_.vy; g_W.vz[i] = R_.vz;
                g_W.m[i] = R_.m;
                g_W.r[i] = (ax * ax + ay * ay) + az * az;          // parked: the r column receives the sorted radii below
                g_W.cum[i] = __longlong_as_double(R_.fof);         // parked: the cumulative mass is computed below
                unsigned char fl = (unsigned char)ts[e];
                if (R_.grnr == hindex) fl |= FL_BOUND;
                if (R_.grnr >= 0 && R_.grnr != hindex) fl |= FL_SAT;   // "bound to another subhalo": satellite or external once the FOF id of the centre is known
                g_W.flags[i] = fl;
            }
        }
    }
    // ---- sort by (r, packed index)
    int lg = 0;
    while ((1 << lg) < n) lg++;
#ifndef NO_SORT
    small_sort(key, pv, sl, lg);
#endif
    WSYNC();
    // ---- the columns into sorted order: entry i takes what selection slot sl[e] holds (read everything, then write)
#define PERMUTE_COL(col, T) { T t_[SE]; \
        _Pragma("unroll") for (int e = 0; e < SE; e++) { t_[e] = (T)0; if (64 * e + LANE < n) t_[e] = g_W.col[sl[e]]; } \
        WSYNC(); \
        _Pragma("unroll") for (int e = 0; e < SE; e++) if (64 * e + LANE < n) g_W.col[64 * e + LANE] = t_[e]; \
        WSYNC(); }
#ifndef NO_PERM
    PERMUTE_COL(dx, double) PERMUTE_COL(dy, double) PERMUTE_COL(dz, double)
    PERMUTE_COL(vx, double) PERMUTE_COL(vy, double) PERMUTE_COL(vz, double)
    PERMUTE_COL(m, double)
    PERMUTE_COL(flags, unsigned char)
#endif
    long long fof[SE];
#pragma unroll
    for (int e = 0; e < SE; e++) { q[e] = INFINITY; fof[e] = 0; if (64 * e + LANE < n) { q[e] = g_W.r[sl[e]]; fof[e] = __double_as_longlong(g_W.cum[sl[e]]); } }
    WSYNC();
#pragma unroll
    for (int e = 0; e < SE; e++) if (64 * e + LANE < n) { g_W.r[64 * e + LANE] = key[e]; g_W.pidx[64 * e + LANE] = pv[e]; }
#undef PERMUTE_COL
    // ---- the innermost particle: FOF id, velocity shift (no neutrinos in this class: it is entry 0)
    const long long f0u = uni(fof[0]);
    const long long f0 = n > 0 ? f0u : -1ll;
    *fof0 = f0;
    WSYNC();
    double sh0 = 0.0, sh1 = 0.0, sh2 = 0.0;
    if (n > 0) { sh0 = g_W.vx[0]; sh1 = g_W.vy[0]; sh2 = g_W.vz[0]; }
    WSYNC();
    if (LANE == 0) { g_W.shift[0] = sh0; g_W.shift[1] = sh1; g_W.shift[2] = sh2; }
#pragma unroll
    for (int e = 0; e < SE; e++) {
        const int i = 64 * e + LANE;
        if (i < n) {
            g_W.vx[i] -= sh0; g_W.vy[i] -= sh1; g_W.vz[i] -= sh2;
            unsigned char fl = g_W.flags[i];
            if (fl & FL_SAT) { if (fof[e] != f0) fl = (unsigned char)((fl & ~FL_SAT) | FL_EXT); g_W.flags[i] = fl; }
        }
    }
    WSYNC();
    // ---- cumulative mass in numpy.cumsum order: ONE sequential chain (bit-identical to the reference, so that the SO
    // radii and with them the SO memberships are); 16 LDS reads are in flight ahead of the chain
#ifndef NO_CUM
    if (LANE == 0) {
        double run = 0.0;
        int j = 0;
        for (; j + 16 <= n; j += 16) {
            double a_[16];
#pragma unroll
            for (int u = 0; u < 16; u++) a_[u] = g_W.m[j + u];
#pragma unroll
            for (int u = 0; u < 16; u++) { run += a_[u]; a_[u] = run; }
#pragma unroll
            for (int u = 0; u < 16; u++) g_W.cum[j + u] = a_[u];
        }
        for (; j < n; j++) { run += g_W.m[j]; g_W.cum[j] = run; }
    }
#endif
    WSYNC();
    return n;
}

#endif   // SGLOBAL

// order the calculations of one family by prefix length (lanes = calculations): returns how many there are, *rank this
// lane's position (or -1); g_W.bnd[k] the k-th prefix length
__device__ __forceinline__ int small_family_order(bool in, int nc, int *rank_out) {
    const unsigned long long members = __ballot(in);
    const int m = __popcll(members);
    int rank = 0;
    unsigned long long rest = members;
    while (rest) {
        const int c2 = __ffsll((long long)rest) - 1;
        rest &= rest - 1;
        const int n2 = rl(nc, c2);
        if (n2 < nc || (n2 == nc && c2 < LANE)) rank++;
    }
    if (in) g_W.bnd[rank] = nc;
    *rank_out = in ? rank : -1;
    WSYNC();
    return m;
}

// per-lane (= per-calculation) state of the radius loop and of stage B
struct CalcState {
    int state;        // 0 pending, 1 selection known, 2 done with zero columns, 3 copy of calculation n's columns
    int n;            // prefix length of the selection (or the parent for copies)
    int ev;           // >= 0: evaluated successfully in an earlier round (state to commit); after the commit: size of that round's sphere
    double R, M;      // SO radius and mass
};

// ======================= stage B =========================================================================================
// prefix moments of one family of calculations (bound_family: bound subhalo + exclusive spheres, else SO + inclusive
// spheres): ONE pass over the slab, every block of 64 entries turned into inclusive wave scans, the lane that holds the last
// member of a selection stores that selection's prefix moments (fused_moments_small of halo_kernel.inc).  Returns the
// family's A_NFAR total through *nfar (bound family: np.isclose(r, 0) test of the unsoftened v_max).
struct MomOut {       // what the later passes need of a calculation's prefix moments (this lane's calculation)
    double M[4];      // mass per type slot
    double N1;        // dark matter particles
    double MV[3];     // sum m v' (slab frame)
};
__device__ __forceinline__ void small_moments(GAS double *rowbase, bool bound_family, const soap_calc_desc &cd, int coloff, const CalcState &cs,
                                              bool in, int rank, int m, const double (&soft)[4], double *nfar, MomOut *mo) {
    const double soft1 = soft[1]; (void)soft1;
    *nfar = 0.0;
    for (int t = 0; t < 4; t++) mo->M[t] = 0.0;
    mo->N1 = 0.0; mo->MV[0] = mo->MV[1] = mo->MV[2] = 0.0;
    if (m == 0) return;
    const int nmax = rl(LANE < m ? g_W.bnd[LANE < m ? LANE : 0] : 0, m - 1);
    const int my_nc = (LANE < m) ? g_W.bnd[LANE] : -1;
#pragma unroll 1
    for (int k0 = 0; k0 < m; k0 += SPREFB) {
        const int k1 = (k0 + SPREFB < m) ? k0 + SPREFB : m;
        for (int e = LANE; e < (k1 - k0) * SMOM; e += 64) (&g_W.prefc[0][0])[e] = 0.0;
        WSYNC();
#if !SHYDRO
        // dark matter only: the per-type columns of the absent types stay zero and the dark matter sums ARE the all-particle
        // sums -- 23 scans instead of 38, in two sweeps over the slab (12 + 11 running sums in registers instead of 23)
        {
            constexpr int NL = 12;
            double carry[NL];
#pragma unroll
            for (int a = 0; a < NL; a++) carry[a] = 0.0;
#pragma unroll 1
            for (int base = 0; base < nmax; base += 64) {
                const int i = base + LANE;
                double c[NL];
#pragma unroll
                for (int a = 0; a < NL; a++) c[a] = 0.0;
                if (i < nmax) {
                    const unsigned char fl = g_W.flags[i];
                    if (!bound_family || (fl & FL_BOUND)) {
                        const double mm = CX(m)[i], r = g_W.r[i];
                        const double x = CX(dx)[i], y = CX(dy)[i], z = CX(dz)[i];
                        const double u = CX(vx)[i], v = CX(vy)[i], w = CX(vz)[i];
                        c[0] = 1.0; c[1] = mm;
                        c[2] = mm * x; c[3] = mm * y; c[4] = mm * z;
                        c[5] = mm * u; c[6] = mm * v; c[7] = mm * w;
                        c[8] = mm * (y * w - z * v); c[9] = mm * (z * u - x * w); c[10] = mm * (x * v - y * u);
                        c[11] = mm * r;
                    }
                }
#pragma unroll
                for (int a = 0; a < NL; a++) { c[a] = wave_scan(c[a], LANE) + carry[a]; if ((a & 3) == 3) __builtin_amdgcn_sched_barrier(0); }
                for (int k = k0; k < k1; k++) {
                    const int nc = rl(my_nc, k);
                    if (nc > base && nc <= base + 64 && LANE == nc - 1 - base) {
                        double *dst = g_W.prefc[k - k0];
                        dst[A_N1] = c[0]; dst[A_M1] = c[1];
                        dst[A_MX] = c[2]; dst[A_MX + 1] = c[3]; dst[A_MX + 2] = c[4];
                        dst[A_MV] = c[5]; dst[A_MV + 1] = c[6]; dst[A_MV + 2] = c[7];
                        dst[A_MVD] = c[5]; dst[A_MVD + 1] = c[6]; dst[A_MVD + 2] = c[7];
                        dst[A_MXD] = c[2]; dst[A_MXD + 1] = c[3]; dst[A_MXD + 2] = c[4];
                        dst[A_LA] = c[8]; dst[A_LA + 1] = c[9]; dst[A_LA + 2] = c[10];
                        dst[A_LD] = c[8]; dst[A_LD + 1] = c[9]; dst[A_LD + 2] = c[10];
                        dst[A_MR] = c[11];
                    }
                }
#pragma unroll
                for (int a = 0; a < NL; a++) carry[a] = rl64(c[a], 63);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            constexpr int NL = 11;
            double carry[NL];
#pragma unroll
            for (int a = 0; a < NL; a++) carry[a] = 0.0;
#pragma unroll 1
            for (int base = 0; base < nmax; base += 64) {
                const int i = base + LANE;
                double c[NL];
#pragma unroll
                for (int a = 0; a < NL; a++) c[a] = 0.0;
                if (i < nmax) {
                    const unsigned char fl = g_W.flags[i];
                    if (!bound_family || (fl & FL_BOUND)) {
                        const double mm = CX(m)[i], r = g_W.r[i];
                        const double u = CX(vx)[i], v = CX(vy)[i], w = CX(vz)[i];
                        c[0] = mm * u * u; c[1] = mm * v * v; c[2] = mm * w * w;
                        c[3] = mm * u * v; c[4] = mm * u * w; c[5] = mm * v * w;
                        c[6] = mm * fmax(soft1, r);
                        c[7] = (fl & FL_SAT) ? mm : 0.0;
                        c[8] = (fl & FL_EXT) ? mm : 0.0;
                        c[9] = (r > 1e-8) ? 1.0 : 0.0;
                        c[10] = r;
                    }
                }
#pragma unroll
                for (int a = 0; a < NL - 1; a++) { c[a] = wave_scan(c[a], LANE) + carry[a]; if ((a & 3) == 3) __builtin_amdgcn_sched_barrier(0); }
                c[10] = fmax(wave_scan_max(c[10]), carry[10]);
                for (int k = k0; k < k1; k++) {
                    const int nc = rl(my_nc, k);
                    if (nc > base && nc <= base + 64 && LANE == nc - 1 - base) {
                        double *dst = g_W.prefc[k - k0];
                        dst[A_VV] = c[0]; dst[A_VV + 1] = c[1]; dst[A_VV + 2] = c[2]; dst[A_VV + 3] = c[3]; dst[A_VV + 4] = c[4]; dst[A_VV + 5] = c[5];
                        dst[A_MRS] = c[6]; dst[A_MSAT] = c[7]; dst[A_MEXT] = c[8]; dst[A_NFAR] = c[9]; dst[A_RMAX] = c[10];
                    }
                }
#pragma unroll
                for (int a = 0; a < NL; a++) carry[a] = rl64(c[a], 63);
            }
            *nfar = carry[9];
        }
        __builtin_amdgcn_sched_barrier(0);
#else
        // chunks with gas / stars / black holes: the full moment set of halo_kernel.inc's reduce_range (counts and masses per type
        // slot, the all-particle sums, then momentum / position / angular momentum / velocity products of the dark matter, the gas
        // and the stars), as inclusive wave scans in five sweeps of at most 15 running sums
#define SMALL_SWEEP(NL, IMAX, FILL, STORE)                                                                                   \
        {                                                                                                                        \
            double carry[NL];                                                                                                    \
            _Pragma("unroll") for (int a = 0; a < NL; a++) carry[a] = 0.0;                                                       \
            _Pragma("unroll 1") for (int base = 0; base < nmax; base += 64) {                                                    \
                const int i = base + LANE;                                                                                       \
                double c[NL];                                                                                                    \
                _Pragma("unroll") for (int a = 0; a < NL; a++) c[a] = 0.0;                                                       \
                if (i < nmax) {                                                                                                  \
                    const unsigned char fl = g_W.flags[i];                                                                       \
                    const int t = fl & FL_TYPE;                                                                                  \
                    if (!bound_family || (fl & FL_BOUND)) { FILL }                                                               \
                }                                                                                                                \
                _Pragma("unroll") for (int a = 0; a < NL; a++) {                                                                 \
                    if (a == IMAX) c[a] = fmax(wave_scan_max(c[a]), carry[a]);                                                   \
                    else c[a] = wave_scan(c[a], LANE) + carry[a];                                                                \
                    if ((a & 3) == 3) __builtin_amdgcn_sched_barrier(0);                                                         \
                }                                                                                                                \
                for (int k = k0; k < k1; k++) {                                                                                  \
                    const int nc = rl(my_nc, k);                                                                                 \
                    if (nc > base && nc <= base + 64 && LANE == nc - 1 - base) { double *dst = g_W.prefc[k - k0]; STORE }        \
                }                                                                                                                \
                _Pragma("unroll") for (int a = 0; a < NL; a++) carry[a] = rl64(c[a], 63);                                        \
            }                                                                                                                    \
            LAST                                                                                                                 \
        }                                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);
#define LAST
#define SW_FILL_ { const double mm = CX(m)[i], r = g_W.r[i], x = CX(dx)[i], y = CX(dy)[i], z = CX(dz)[i]; \
              c[0] = t == 0 ? 1.0 : 0.0; c[1] = t == 1 ? 1.0 : 0.0; c[2] = t == 2 ? 1.0 : 0.0; c[3] = t == 3 ? 1.0 : 0.0; \
              c[4] = t == 0 ? mm : 0.0; c[5] = t == 1 ? mm : 0.0; c[6] = t == 2 ? mm : 0.0; c[7] = t == 3 ? mm : 0.0; \
              c[8] = mm * x; c[9] = mm * y; c[10] = mm * z; c[11] = mm * r; }
#define SW_STORE_ { dst[A_N0] = c[0]; dst[A_N1] = c[1]; dst[A_N2] = c[2]; dst[A_N3] = c[3]; \
              dst[A_M0] = c[4]; dst[A_M1] = c[5]; dst[A_M2] = c[6]; dst[A_M3] = c[7]; \
              dst[A_MX] = c[8]; dst[A_MX + 1] = c[9]; dst[A_MX + 2] = c[10]; dst[A_MR] = c[11]; }
        SMALL_SWEEP(12, -1, SW_FILL_, SW_STORE_)
#undef SW_FILL_
#undef SW_STORE_
#undef LAST
#define LAST *nfar = carry[9];
#define SW_FILL_ { const double mm = CX(m)[i], r = g_W.r[i], x = CX(dx)[i], y = CX(dy)[i], z = CX(dz)[i]; \
              const double u = CX(vx)[i], v = CX(vy)[i], w = CX(vz)[i]; \
              c[0] = mm * u; c[1] = mm * v; c[2] = mm * w; \
              c[3] = mm * (y * w - z * v); c[4] = mm * (z * u - x * w); c[5] = mm * (x * v - y * u); \
              c[6] = mm * fmax(t == 0 ? soft[0] : (t == 1 ? soft[1] : (t == 2 ? soft[2] : soft[3])), r); \
              c[7] = (fl & FL_SAT) ? mm : 0.0; c[8] = (fl & FL_EXT) ? mm : 0.0; c[9] = (r > 1e-8) ? 1.0 : 0.0; c[10] = r; }
#define SW_STORE_ { dst[A_MV] = c[0]; dst[A_MV + 1] = c[1]; dst[A_MV + 2] = c[2]; \
              dst[A_LA] = c[3]; dst[A_LA + 1] = c[4]; dst[A_LA + 2] = c[5]; \
              dst[A_MRS] = c[6]; dst[A_MSAT] = c[7]; dst[A_MEXT] = c[8]; dst[A_NFAR] = c[9]; dst[A_RMAX] = c[10]; }
        SMALL_SWEEP(11, 10, SW_FILL_, SW_STORE_)
#undef SW_FILL_
#undef SW_STORE_
#undef LAST
#define LAST
#pragma unroll 1
        for (int pass = 0; pass < 3; pass++) {
            // momentum, position, angular momentum and velocity products of one component: dark matter, gas, stars
            const int want_t = pass == 0 ? 1 : (pass == 1 ? 0 : 2);
            const int b_mv = pass == 0 ? A_MVD : (pass == 1 ? A_GMV : A_SMV), b_mx = pass == 0 ? A_MXD : (pass == 1 ? A_GMX : A_SMX);
            const int b_l = pass == 0 ? A_LD : (pass == 1 ? A_GL : A_SL), b_vv = pass == 0 ? A_VV : (pass == 1 ? A_GVV : A_SVV);
#define SW_FILL_ { if (t == want_t) { \
                    const double mm = CX(m)[i], x = CX(dx)[i], y = CX(dy)[i], z = CX(dz)[i]; \
                    const double u = CX(vx)[i], v = CX(vy)[i], w = CX(vz)[i]; \
                    c[0] = mm * u; c[1] = mm * v; c[2] = mm * w; c[3] = mm * x; c[4] = mm * y; c[5] = mm * z; \
                    c[6] = mm * (y * w - z * v); c[7] = mm * (z * u - x * w); c[8] = mm * (x * v - y * u); \
                    c[9] = mm * u * u; c[10] = mm * v * v; c[11] = mm * w * w; c[12] = mm * u * v; c[13] = mm * u * w; c[14] = mm * v * w; } }
#define SW_STORE_ { for (int d = 0; d < 3; d++) { dst[b_mv + d] = c[d]; dst[b_mx + d] = c[3 + d]; dst[b_l + d] = c[6 + d]; } \
                  for (int d = 0; d < 6; d++) dst[b_vv + d] = c[9 + d]; }
            SMALL_SWEEP(15, -1, SW_FILL_, SW_STORE_)
#undef SW_FILL_
#undef SW_STORE_
        }
#undef LAST
#undef SMALL_SWEEP
#endif
        WSYNC();
        if (in && rank >= k0 && rank < k1) {
            const volatile double *a = g_W.prefc[rank - k0];
#ifndef NO_FIN
            finalize_moments(LANE, cd, rowbase + coloff, a, cs.R, cs.M);
#endif
#if SHYDRO
            for (int t = 0; t < 4; t++) mo->M[t] = a[A_M0 + t];
#else
            mo->M[1] = a[A_M1];
#endif
            mo->N1 = a[A_N1];
            for (int d = 0; d < 3; d++) mo->MV[d] = a[A_MV + d];
        }
        WSYNC();
    }
}

// radial scan of one family (fused_scan of halo_kernel.inc on a single wave): v_max (kinematic_properties.py:555-593: arg-max
// of M(<r) / max(r, softening), first maximum) at the end of every selection, the unsoftened v_max of the bound subhalo
// (subhalo_properties.py:994-996), half-mass radii (half_mass_radius.py:40-80) -- then finalize_scan with lanes = calculations
__device__ __forceinline__ void small_scan(GAS double *rowbase, bool bound_family, const soap_calc_desc &cd, int coloff, const CalcState &cs,
                                           bool in, int m, const MomOut &mo, const double (&soft)[4], double nfar_total, double nbound_total,
                                           double *vs_out, double *rs_out) {
    const double soft1 = soft[1]; (void)soft1;
    *vs_out = -1.0; *rs_out = 0.0;
    if (m == 0) return;
    const int nmax = rl(LANE < m ? g_W.bnd[LANE < m ? LANE : 0] : 0, m - 1);
    // half-mass targets of this lane's calculation: [0] all members (bound subhalo only), [1] dark matter
    double tg[SNHM];
#pragma unroll
    for (int w = 0; w < SNHM; w++) tg[w] = 0.0;
    if (in) {
        const double Mtot = ((mo.M[0] + mo.M[1]) + mo.M[2]) + mo.M[3];
        if (cd.kind == SOAP_CALC_SUBHALO) { tg[0] = 0.5 * Mtot; tg[1] = 0.5 * mo.M[1]; }
        else if (cd.kind == SOAP_CALC_APERTURE_INCL || cd.kind == SOAP_CALC_APERTURE_EXCL) tg[1] = 0.5 * mo.M[1];
#if SHYDRO
        // ... [2] gas, [3] stars, [4] baryons (aperture_properties.py:3469-3548, subhalo_properties.py:2265-2334)
        if (cd.kind == SOAP_CALC_SUBHALO || cd.kind == SOAP_CALC_APERTURE_INCL || cd.kind == SOAP_CALC_APERTURE_EXCL) {
            tg[2] = 0.5 * mo.M[0]; tg[3] = 0.5 * mo.M[2]; tg[4] = 0.5 * (mo.M[0] + mo.M[2]);
        }
#endif
    }
    double hm[SNHM];
#pragma unroll
    for (int w = 0; w < SNHM; w++) hm[w] = 0.0;
    const bool want_unsoft = bound_family;
    const bool allclose = bound_family && nbound_total > 0.0 && nfar_total == 0.0;
    double carryA = 0.0, carryB = 0.0, carryC = 0.0, carryV = -1.0, carryS = 0.0;
#if SHYDRO
    double carryG = 0.0, carryT = 0.0;   // cumulative gas / stellar mass
#endif
    double bestU = -1.0, bestUr = 0.0;
    long long bestUi = 0x7fffffffffffffffLL;
    double my_vs = -1.0, my_rs = 0.0;
    const int my_n = in ? cs.n : -1;
#pragma unroll 1
    for (int base = 0; base < nmax; base += 64) {
        const int i = base + LANE;
        bool mem = false;
        int t = 1;
        double mm = 0.0, r = 0.0;
        if (i < nmax) {
            const unsigned char fl = g_W.flags[i];
            mem = !bound_family || (fl & FL_BOUND);
            t = fl & FL_TYPE;
            if (mem) { mm = CX(m)[i]; r = g_W.r[i]; }
        }
        const double sA = wave_scan(mm, LANE) + carryA;
#if SHYDRO
        const double sB = wave_scan((t == 1) ? mm : 0.0, LANE) + carryB;
        const double sG = wave_scan((t == 0) ? mm : 0.0, LANE) + carryG, sT = wave_scan((t == 2) ? mm : 0.0, LANE) + carryT;
#else
        const double sB = sA;
#endif
        const double sC = wave_scan(mem ? 1.0 : 0.0, LANE) + carryC;
        // ---- v_max: this particle's candidate, the running arg-max up to it (ties keep the lower index)
        double wv = -1.0, ws = 0.0;
        if (mem) {
#if SHYDRO
            const double sr = fmax(t == 0 ? soft[0] : (t == 1 ? soft[1] : (t == 2 ? soft[2] : soft[3])), r);   // the softening of the particle's type
#else
            const double sr = fmax(soft1, r);
#endif
            if (sr > 1e-8) { wv = sA / sr; ws = sr; }
            if (want_unsoft) {
                const double rank = sC - 1.0;   // members before this one; nskip = 1 (subhalo_properties.py:994-996)
                if (rank >= 1.0 && (r > 1e-8 || allclose) && r != 0.0) {
                    const double val = sA / r;
                    if (val > bestU) { bestU = val; bestUr = r; bestUi = i; }
                }
            }
        }
        wave_scan_argmax(wv, ws);
        if (!(wv > carryV)) { wv = carryV; ws = carryS; }
        // ---- the running maximum at the end of the selections that end in this block: fetched by the calculation's lane
        {
            const int j = my_n - 1 - base;
            const int src = (j >= 0 && j < 64) ? j : 0;
            const double bv = __shfl(wv, src, 64), bs = __shfl(ws, src, 64);
            if (j >= 0 && j < 64) { my_vs = bv; my_rs = bs; }
        }
        // ---- half-mass crossings
        // (the shifts BEFORE the lane-dependent choice: inside its branch lane 0 is switched off, and a DPP move whose source lane
        // is switched off delivers zero -- lane 1 would see no mass before it)
        const double shA = dpp_f64<DPP_WAVE_SHR1>(sA), shB = dpp_f64<DPP_WAVE_SHR1>(sB);
        const double pA = LANE == 0 ? carryA : shA, pB = LANE == 0 ? carryB : shB;
#if SHYDRO
        const double shG = dpp_f64<DPP_WAVE_SHR1>(sG), shT = dpp_f64<DPP_WAVE_SHR1>(sT);
        const double pG = LANE == 0 ? carryG : shG, pT = LANE == 0 ? carryT : shT;
#define SHM_CV(w) ((w) == 0 ? sA : ((w) == 1 ? sB : ((w) == 2 ? sG : ((w) == 3 ? sT : sG + sT))))
#define SHM_PV(w) ((w) == 0 ? pA : ((w) == 1 ? pB : ((w) == 2 ? pG : ((w) == 3 ? pT : pG + pT))))
#define SHM_CLS(w, tt) ((w) == 0 || ((w) == 1 && (tt) == 1) || ((w) == 2 && (tt) == 0) || ((w) == 3 && (tt) == 2) || ((w) == 4 && ((tt) == 0 || (tt) == 2)))
#else
#define SHM_CV(w) ((w) == 0 ? sA : sB)
#define SHM_PV(w) ((w) == 0 ? pA : pB)
#define SHM_CLS(w, tt) ((w) == 0 || (tt) == 1)
#endif
#pragma unroll
        for (int w = 0; w < SNHM; w++) {
            unsigned long long open = __ballot(in && tg[w] > 0.0 && my_n > base);
            while (open) {
                const int c2 = __ffsll((long long)open) - 1;
                open &= open - 1;
                const double tgt = rl64(tg[w], c2);
                const int n2 = rl(my_n, c2);
                const double cv = SHM_CV(w);
                const bool cls = SHM_CLS(w, t);
                const unsigned long long hit = __ballot(mem && cls && i < n2 && cv >= tgt);
                if (!hit) continue;
                const int owner = __ffsll((long long)hit) - 1;
                // this particle is the first whose cumulative weight reaches the target (half_mass_radius.py:40-80); what is
                // known about it is made uniform and every lane computes the same radius (no value that exists in one lane only)
                const double cv_o = rl64(cv, owner), r_o = rl64(r, owner);
                double Wmin = rl64(SHM_PV(w), owner), rmin = 0.0;
                int j = base + owner - 1;   // previous member of the same class
                while (j >= 0) {
                    const unsigned char fj = g_W.flags[j];
                    if ((!bound_family || (fj & FL_BOUND)) && SHM_CLS(w, fj & FL_TYPE)) break;
                    j--;
                }
                if (j < 0) Wmin = 0.0; else rmin = g_W.r[j];
                const double h = (Wmin == cv_o) ? 0.5 * (rmin + r_o) : rmin + (tgt - Wmin) / (cv_o - Wmin) * (r_o - rmin);
                if (LANE == c2) { hm[w] = h; tg[w] = 0.0; }   // found: the target is closed
            }
        }
        carryA = rl64(sA, 63); carryB = rl64(sB, 63); carryC = rl64(sC, 63);
        carryV = rl64(wv, 63); carryS = rl64(ws, 63);
#if SHYDRO
        carryG = rl64(sG, 63); carryT = rl64(sT, 63);
#endif
    }
#undef SHM_CV
#undef SHM_PV
#undef SHM_CLS
    double vu = -1.0, ru = 0.0;
    if (want_unsoft) {
        // arg-max over all bound members, first maximum: largest value, then the smallest index
        const double vmax = wave_max(bestU);
        const long long icand = (bestU == vmax) ? bestUi : 0x7fffffffffffffffLL;
        const long long imin = wave_min_ll(icand);
        const unsigned long long who = __ballot(bestU == vmax && bestUi == imin);
        const int src = __ffsll((long long)who) - 1;
        vu = vmax; ru = rl64(bestUr, src < 0 ? 0 : src);
    }
    if (in) {
        ScanResult sr;
        sr.vs = my_vs; sr.rs = my_rs; sr.vu = vu; sr.ru = ru;
#pragma unroll
        for (int w = 0; w < NHM; w++) sr.hm[w] = w < SNHM ? hm[w < SNHM ? w : 0] : 0.0;
        sr.cs_R = cs.R;
        finalize_scan(cd, rowbase + coloff, sr);
    }
    *vs_out = my_vs; *rs_out = my_rs;
}

// BoundSubhalo spin: bound particles with r <= R_vmax_soft about the subhalo's vcom (subhalo_properties.py:1046-1073);
// vs / rs: the soft v_max result of calculation 0, Mtot / mv: its total mass and sum m v'
__device__ __forceinline__ void small_subhalo_spin(GAS double *row, int n0, double vs, double rs, double Mtot, const double *mv) {
    const double vmax_s = (Mtot != 0.0 && vs > 0.0) ? sqrt(vs * g_S.G_eff) : 0.0;
    const double rmax_s = (Mtot != 0.0 && vs > 0.0) ? rs : 0.0;
    if (!(Mtot != 0.0 && rmax_s > 0.0 && vmax_s > 0.0)) return;
    double sm = 0.0, mx0 = 0.0, mx1 = 0.0, mx2 = 0.0, l0 = 0.0, l1 = 0.0, l2 = 0.0;
#pragma unroll
    for (int e = 0; e < SE; e++) {
        const int i = 64 * e + LANE;
        if (i < n0 && (g_W.flags[i] & FL_BOUND) && g_W.r[i] <= rmax_s) {
            const double mm = CX(m)[i], x = CX(dx)[i], y = CX(dy)[i], z = CX(dz)[i], u = CX(vx)[i], v = CX(vy)[i], w = CX(vz)[i];
            sm += mm; mx0 += mm * x; mx1 += mm * y; mx2 += mm * z;
            l0 += mm * (y * w - z * v); l1 += mm * (z * u - x * w); l2 += mm * (x * v - y * u);
        }
    }
    const double p1 = wave_sum4(sm, mx0, mx1, mx2), p2 = wave_sum4(l0, l1, l2, 0.0);
    const double Mr = rl64(p1, 0);
    const double bx0 = rl64(p1, 16), bx1 = rl64(p1, 32), bx2 = rl64(p1, 48);
    const double bl0 = rl64(p2, 0), bl1 = rl64(p2, 16), bl2 = rl64(p2, 32);
    if (LANE == 0 && Mr > 0.0) {
        // row[SH_vcom] - shift, as the general path reads it back
        const double iMt = 1.0 / Mtot;
        const double dv0 = (mv[0] * iMt + g_S.shift[0]) - g_S.shift[0], dv1 = (mv[1] * iMt + g_S.shift[1]) - g_S.shift[1],
                     dv2 = (mv[2] * iMt + g_S.shift[2]) - g_S.shift[2];
        // norm_L_about of halo_finalize.inc on registers (its moment array is read as volatile memory in this file: a local array
        // would live in scratch)
        const double lx = bl0 - (bx1 * dv2 - bx2 * dv1), ly = bl1 - (bx2 * dv0 - bx0 * dv2), lz = bl2 - (bx0 * dv1 - bx1 * dv0);
        row[SH_spin] = sqrt((lx * lx + ly * ly) + lz * lz) / (sqrt(2.0) * Mr * vmax_s * rmax_s);
    }
}

// ======================= dark matter mass flow rates (SO_properties.py:2791-2932; flow_batch of halo_kernel.inc) =========
// Mass flux of the dark matter through shells of width dR = 0.1 R centred on R = 0.1, 0.3, 1 R_SO: radial velocity relative
// to the centre-of-mass velocity of ALL particles inside R and to the pseudo-evolution of the radius; sum of m |v_r| / dR
// over the inflowing and the outflowing particles.  One calculation per call: SO_r / SO_mass its radius and mass, vc_own the
// centre-of-mass velocity of its sphere in the frame of the slab velocities, n the size of the search sphere.
__device__ __forceinline__ void small_flow_one(const CAS RunParams &P, GAS double *crow, double SO_r, double SO_mass, const double *vc_own, int n) {
    // ---- bounds: lane b < 8 (two inner spheres, three shells)
    int bnd = 0;
    {
        const int b = LANE & 7;
        double T;
        bool strict = true;
        if (b < 2) T = (b == 0 ? 0.1 : 0.3) * SO_r;        // radius < R_frac * SO_r (:2796, :2808)
        else {
            const int sh = (b - 2) >> 1;
            const double R = (sh == 0 ? 0.1 : sh == 1 ? 0.3 : 1.0) * SO_r, dR = 0.1 * R;
            if ((b & 1) == 0) { T = R - (dR / 2); strict = false; }   // first index with r > lo
            else T = R + (dR / 2);                                    // first index with r >= hi
        }
        int lo = 0, hi = n;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const double rv = g_W.r[mid];
            if (strict ? (rv < T) : (rv <= T)) lo = mid + 1; else hi = mid;
        }
        bnd = lo;
    }
    const int k10 = rl(bnd, 0), k30 = rl(bnd, 1);
    // ---- pass 1: sum m, sum m v' and the particle count inside 0.1 and 0.3 R_SO (prefixes of the sorted slab)
    double vc[3][3];
    {
        double a[10];
#pragma unroll
        for (int k = 0; k < 10; k++) a[k] = 0.0;
        for (int i = LANE; i < k30; i += 64) {
            const double mm = CX(m)[i], ux = CX(vx)[i], uy = CX(vy)[i], uz = CX(vz)[i];
            const double mx = mm * ux, my = mm * uy, mz = mm * uz;
            const bool in10 = i < k10;
            a[0] += in10 ? mm : 0.0; a[1] += in10 ? mx : 0.0; a[2] += in10 ? my : 0.0; a[3] += in10 ? mz : 0.0;
            a[4] += mm; a[5] += mx; a[6] += my; a[7] += mz;
            a[8] += in10 ? 1.0 : 0.0; a[9] += 1.0;
        }
        const double p1 = wave_sum4(a[0], a[1], a[2], a[3]), p2 = wave_sum4(a[4], a[5], a[6], a[7]), p3 = wave_sum4(a[8], a[9], 0.0, 0.0);
        const double cnt10 = rl64(p3, 0), cnt30 = rl64(p3, 16);
        const double m10 = rl64(p1, 0), m30 = rl64(p2, 0);
        const double s10x = rl64(p1, 16), s10y = rl64(p1, 32), s10z = rl64(p1, 48), s30x = rl64(p2, 16), s30y = rl64(p2, 32), s30z = rl64(p2, 48);
        // no particle inside: zero velocity ("HaloCentre doesn't actually need to be on top of a particle", :2860-2863)
        vc[0][0] = cnt10 == 0.0 ? -g_S.shift[0] : s10x / m10;
        vc[0][1] = cnt10 == 0.0 ? -g_S.shift[1] : s10y / m10;
        vc[0][2] = cnt10 == 0.0 ? -g_S.shift[2] : s10z / m10;
        vc[1][0] = cnt30 == 0.0 ? -g_S.shift[0] : s30x / m30;
        vc[1][1] = cnt30 == 0.0 ? -g_S.shift[1] : s30y / m30;
        vc[1][2] = cnt30 == 0.0 ? -g_S.shift[2] : s30z / m30;
        vc[2][0] = vc_own[0]; vc[2][1] = vc_own[1]; vc[2][2] = vc_own[2];
    }
    // ---- pass 2: the shells, their index ranges laid end to end
    int base[3], off[4];
    off[0] = 0;
#pragma unroll
    for (int sh = 0; sh < 3; sh++) {
        const int i0 = rl(bnd, 2 + 2 * sh), i1 = rl(bnd, 3 + 2 * sh);
        off[sh + 1] = off[sh] + (i1 > i0 ? i1 - i0 : 0);
        base[sh] = i0 - off[sh];
    }
    // R_dot = (2/3) (G M H / 100)^(1/3) (2 Omega_g + 3/2 Omega_m) R_frac   (:2875-2884)
    double Rd0 = (2.0 / 3.0) * cbrt(P.G_phys * SO_mass * P.hubble / 100.0);
    Rd0 *= P.rdot_omega;
    double fin[3] = {0.0, 0.0, 0.0}, fout[3] = {0.0, 0.0, 0.0};
    const int total = off[3];
    for (int j = LANE; j < total; j += 64) {
        const int sh = (j >= off[1] ? 1 : 0) + (j >= off[2] ? 1 : 0);
        const int i = j + (sh == 0 ? base[0] : (sh == 1 ? base[1] : base[2]));
        const unsigned char tf = g_W.flags[i];
        const double rr = g_W.r[i], mm = CX(m)[i];
        const double x = CX(dx)[i], y = CX(dy)[i], z = CX(dz)[i];
        const double ux = CX(vx)[i], uy = CX(vy)[i], uz = CX(vz)[i];
        const double vc0 = sh == 0 ? vc[0][0] : (sh == 1 ? vc[1][0] : vc[2][0]);
        const double vc1 = sh == 0 ? vc[0][1] : (sh == 1 ? vc[1][1] : vc[2][1]);
        const double vc2 = sh == 0 ? vc[0][2] : (sh == 1 ? vc[1][2] : vc[2][2]);
        const double R_dot = Rd0 * (sh == 0 ? 0.1 : sh == 1 ? 0.3 : 1.0);
        const bool dmp = (tf & FL_TYPE) == 1;   // dark matter only
        const double ir = 1.0 / rr;
        double v_r = ((ux - vc0) * (x * ir) + (uy - vc1) * (y * ir)) + (uz - vc2) * (z * ir);
        v_r -= R_dot;
        const double fl = mm * fabs(v_r);
        const bool in = dmp && v_r < 0.0, out = dmp && v_r > 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            fin[k] += (in && sh == k) ? fl : 0.0;
            fout[k] += (out && sh == k) ? fl : 0.0;
        }
    }
    const double q1 = wave_sum4(fin[0], fin[1], fin[2], fout[0]), q2 = wave_sum4(fout[1], fout[2], 0.0, 0.0);
    // lane k < 6: inflow at 0.1 / 0.3 / 1 R_SO, outflow at the same (the six totals are read across the lanes BEFORE the
    // lane-dependent choice: a cross-lane read inside a lane-dependent branch sees only that branch's lanes)
    const double t0 = rl64(q1, 0), t1 = rl64(q1, 16), t2 = rl64(q1, 32), t3 = rl64(q1, 48), t4 = rl64(q2, 0), t5 = rl64(q2, 16);
    const double v = LANE == 0 ? t0 : LANE == 1 ? t1 : LANE == 2 ? t2 : LANE == 3 ? t3 : LANE == 4 ? t4 : t5;
    if (LANE < 6) {
        const double R = (LANE % 3 == 0 ? 0.1 : LANE % 3 == 1 ? 0.3 : 1.0) * SO_r;
        crow[SOC_flow + LANE] = v / (0.1 * R);
    }
}

#if SINERTIA
// ======================= inertia tensors (property_calculation/inertia_tensors.py:19-132) ================================
// As inertia_pass of halo_kernel.inc, for a slab that lives in LDS: a STATE is (calculation, iterative /
// fixed sphere, plain / reduced) -- dark-matter-only chunks, so the only component is "all particles" and the dark matter
// tensors are copies -- and state (group g, row r) belongs to lane 16 r + g.  The four states of a group (two calculations x
// plain / reduced, all iterative or all fixed) sweep the particles together, lanes = particles; wave_sum4 leaves the total of
// the r-th argument in row r, i.e. exactly in the state's own lane: no hand-over.  Shapes, convergence tests and the 3x3
// eigenproblems of all 64 states run side by side, one per lane.  Up to 16 calculations with tensors (engine.hip checks).

// eigenvalues (ascending) and eigenvectors of a symmetric 3x3 matrix, cyclic Jacobi rotations as eigh3 of halo_kernel.inc
// (same operations in the same order), without indexed arrays: every lane solves its own matrix.  V[3 j + k] = component j of
// eigenvector k.
SOAP_DEV void small_eigh3(const double t6[6], double ev[3], double V[9]) {
    double a00 = t6[0], a11 = t6[1], a22 = t6[2], a01 = t6[3], a02 = t6[4], a12 = t6[5];
    double v00 = 1.0, v01 = 0.0, v02 = 0.0, v10 = 0.0, v11 = 1.0, v12 = 0.0, v20 = 0.0, v21 = 0.0, v22 = 1.0;
#pragma unroll 1
    for (int sweep = 0; sweep < 16; sweep++) {
        const double off = (a01 * a01 + a02 * a02) + a12 * a12;
        const double dia = (a00 * a00 + a11 * a11) + a22 * a22;
        const bool go = off > 1e-60 * dia;
        if (!__ballot(go)) break;
#define JACOBI_ROT(app, aqq, apq, arp, arq, v0p, v0q, v1p, v1q, v2p, v2q)                      \
        if (go && apq != 0.0) {                                                              \
            const double theta = (aqq - app) / (2.0 * apq);                                  \
            const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0)); \
            const double cc = 1.0 / sqrt(tt * tt + 1.0), ss = tt * cc;                       \
            app -= tt * apq; aqq += tt * apq; apq = 0.0;                                     \
            const double rp = arp, rq = arq;                                                 \
            arp = cc * rp - ss * rq; arq = ss * rp + cc * rq;                                \
            double vp, vq;                                                                   \
            vp = v0p; vq = v0q; v0p = cc * vp - ss * vq; v0q = ss * vp + cc * vq;            \
            vp = v1p; vq = v1q; v1p = cc * vp - ss * vq; v1q = ss * vp + cc * vq;            \
            vp = v2p; vq = v2q; v2p = cc * vp - ss * vq; v2q = ss * vp + cc * vq;            \
        }
        JACOBI_ROT(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21)
        JACOBI_ROT(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22)
        JACOBI_ROT(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22)
#undef JACOBI_ROT
    }
    // ascending eigenvalues, columns follow: the three compare-and-swap steps of eigh3's index sort
#define EIG_SWAP(ea, eb, xa, xb, ya, yb, za, zb)                                                          \
    if (ea > eb) { double t_; t_ = ea; ea = eb; eb = t_; t_ = xa; xa = xb; xb = t_; t_ = ya; ya = yb; yb = t_; t_ = za; za = zb; zb = t_; }
    EIG_SWAP(a00, a11, v00, v01, v10, v11, v20, v21)
    EIG_SWAP(a11, a22, v01, v02, v11, v12, v21, v22)
    EIG_SWAP(a00, a11, v00, v01, v10, v11, v20, v21)
#undef EIG_SWAP
    ev[0] = a00; ev[1] = a11; ev[2] = a22;
    V[0] = v00; V[1] = v01; V[2] = v02; V[3] = v10; V[4] = v11; V[5] = v12; V[6] = v20; V[7] = v21; V[8] = v22;
}

__device__ __attribute__((noinline)) void small_inertia(GAS double *row, int kind, int flags, int coloff, int cstate, int cn, int cev, double cR,
                                                        int ncalc, double nbound_total) {
    const bool isin = LANE < ncalc && (flags & SOAP_CALC_FLAG_INERTIA) && cstate == 1 && (kind == SOAP_CALC_SUBHALO || (kind >= 0 && kind <= SOAP_CALC_SO_MULTIPLE));
    const unsigned long long imask = __ballot(isin);
    const int nic = __popcll(imask);
    if (nic == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the half-mass radii and masses of the rows were written by other lanes
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- this lane's state
    // (groups 0..7 hold the iterative states; the fixed-sphere tensor of the same calculation and weighting IS the tensor of their first
    // iteration -- same sphere, same particles inside it, the same sums -- and is handed out with it: lanes of groups 8..15 idle)
    const int g = LANE & 15, r4 = LANE >> 4;
    const int mode = 0;
    const int o = 2 * (g & 7) + (r4 >> 1);         // ordinal of the calculation among those with tensors
    const bool reduced = r4 & 1;
    const bool mine = o < nic && g < 8;
    int c = 0;
    {
        unsigned long long mm = imask;
        for (int k = 0; k < 16; k++) if (k < o) mm &= mm - 1;
        if (mine && mm) c = __ffsll((long long)mm) - 1;
    }
    const int c_kind = __shfl(kind, c, 64), c_n = __shfl(cn, c, 64), c_ev = __shfl(cev, c, 64), c_col = __shfl(coloff, c, 64);
    const double c_R = __shfl(cR, c, 64);
    const bool bound_only = c_kind == SOAP_CALC_SUBHALO;
    int n_lim = 0, out = 0;
    double R = 1.0, nall = 0.0;
    bool has = false;
    if (mine) {
        const GAS double *crow = row + c_col;
        if (bound_only) {
            // bound particles, spheres of 10 half-mass radii, no search radius (subhalo_properties.py:1075-1125, :1206-1256)
            n_lim = c_n; R = 10 * crow[SH_HMR_tot]; has = crow[SH_Mtot] != 0.0; nall = nbound_total;
            out = c_col + SH_inertia + mode * 12 + (reduced ? 6 : 0);
        } else {
            // members and surrounding particles for the iterative tensors, members for the fixed sphere (SO_properties.py:620-672, :798-857)
            n_lim = mode == 0 ? c_ev : c_n; R = c_R; has = crow[SOC_Mdm] != 0.0; nall = (double)n_lim;
            out = c_col + SOC_inertia + mode * 12 + (reduced ? 6 : 0);
        }
    }
    const int max_iter = mode == 0 ? 20 : 1;
    const int sflags = (bound_only ? 1 : 0) | (reduced ? 2 : 0);
    double ev0 = 1.0, ev1 = 1.0, ev2 = 1.0, q = 1000.0;
    double T[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double T0[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // the first iteration's tensor: the fixed-sphere one
    bool active = mine && has, valid = true, valid0 = true;
    {
        double *S = g_W.it_par[LANE];
        for (int k = 0; k < 9; k++) S[k] = (k % 4 == 0) ? 1.0 : 0.0;
        S[9] = S[10] = S[11] = 0.0;
    }
#pragma unroll 1
    for (int iter = 0; iter < 20; iter++) {
        // shape of this iteration's ellipsoid, convergence test (:82-96)
        double rlim = 0.0;
        if (active && iter >= max_iter) active = false;
        if (active) {
            const double old_q = q;
            q = sqrt(ev1 / ev2);
            const double s = sqrt(ev0 / ev2);
            const double p = sqrt(ev0 / ev1);
            if (fabs((old_q - q) / q) < 0.0001) active = false;
            else {
                // reciprocals of the semi-axes R cbrt(s p), R cbrt(q / p), R / cbrt(q s)  (:99-101)
                const double ax0 = 1.0 / (R * (1.0 * cbrt(s * p)));
                const double ax1 = 1.0 / (R * (1.0 * cbrt(q / p)));
                const double ax2 = 1.0 / (R * (1.0 / cbrt(q * s)));
                double *S = g_W.it_par[LANE];
                S[9] = ax0; S[10] = ax1; S[11] = ax2;
                rlim = 1.0001 / fmin(fmin(ax0, ax1), ax2);   // nothing beyond the longest semi-axis is inside (the slab is sorted by radius)
            }
        }
        WSYNC();
        const unsigned long long amask = __ballot(active);
        if (!amask) break;
        double s_n = 0.0, s_w = 0.0, s_xx = 0.0, s_yy = 0.0, s_zz = 0.0, s_xy = 0.0, s_xz = 0.0, s_yz = 0.0;
        unsigned gmask = (unsigned)((amask | (amask >> 16) | (amask >> 32) | (amask >> 48)) & 0xffffull);
#pragma unroll 1
        while (gmask) {
            const int gg = __ffs((int)gmask) - 1;
            gmask &= gmask - 1;
            bool act[4];
            int nl[4], sf[4];
            int nmax = 0;
            double rmax = 0.0;
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int L = 16 * rr + gg;
                act[rr] = (amask >> L) & 1ull;
                nl[rr] = rl(n_lim, L); sf[rr] = rl(sflags, L);
                const double rl_ = rl64(rlim, L);
                if (act[rr]) { nmax = max(nmax, nl[rr]); rmax = fmax(rmax, rl_); }
            }
            const int ncut = min(nmax, count_r_le(nmax, rmax));
            double acc[4][8];
#pragma unroll
            for (int rr = 0; rr < 4; rr++)
#pragma unroll
                for (int k = 0; k < 8; k++) acc[rr][k] = 0.0;
#pragma unroll 1
            for (int e = 0; e < SE; e++) {
                if (64 * e >= ncut) break;
                const int i = 64 * e + LANE;
                const bool ok = i < ncut;
                const int ii = ok ? i : 0;
                const int fl = g_W.flags[ii];
                const double x = CX(dx)[ii], y = CX(dy)[ii], z = CX(dz)[ii], w = CX(m)[ii];
                const double rr_ = g_W.r[ii];
                const double norm = rr_ * rr_;              // np.linalg.norm(x, axis=1) ** 2  (:63)
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    if (!act[rr]) continue;                  // (uniform)
                    const double *S = g_W.it_par[16 * rr + gg];
                    const double p0 = ((x * S[0] + y * S[3]) + z * S[6]) * S[9];
                    const double p1 = ((x * S[1] + y * S[4]) + z * S[7]) * S[10];
                    const double p2 = ((x * S[2] + y * S[5]) + z * S[8]) * S[11];
                    const double re = sqrt((p0 * p0 + p1 * p1) + p2 * p2);
                    const bool in = ok && i < nl[rr] && !((sf[rr] & 1) && !(fl & FL_BOUND)) && !((sf[rr] & 2) && norm <= 1e-8) && re <= 1.0;   // np.isclose(norm, 0)  (:64)
                    const double ww = in ? ((sf[rr] & 2) ? w / norm : w) : 0.0;
                    acc[rr][0] += in ? 1.0 : 0.0; acc[rr][1] += in ? w : 0.0;
                    acc[rr][2] += ww * x * x; acc[rr][3] += ww * y * y; acc[rr][4] += ww * z * z;
                    acc[rr][5] += ww * x * y; acc[rr][6] += ww * x * z; acc[rr][7] += ww * y * z;
                }
            }
            // totals: row r of each packed register holds the sum of state (gg, r), i.e. its own lane's
            const double t0 = wave_sum4(acc[0][0], acc[1][0], acc[2][0], acc[3][0]), t1 = wave_sum4(acc[0][1], acc[1][1], acc[2][1], acc[3][1]);
            const double t2 = wave_sum4(acc[0][2], acc[1][2], acc[2][2], acc[3][2]), t3 = wave_sum4(acc[0][3], acc[1][3], acc[2][3], acc[3][3]);
            const double t4 = wave_sum4(acc[0][4], acc[1][4], acc[2][4], acc[3][4]), t5 = wave_sum4(acc[0][5], acc[1][5], acc[2][5], acc[3][5]);
            const double t6 = wave_sum4(acc[0][6], acc[1][6], acc[2][6], acc[3][6]), t7 = wave_sum4(acc[0][7], acc[1][7], acc[2][7], acc[3][7]);
            if (g == gg) { s_n = t0; s_w = t1; s_xx = t2; s_yy = t3; s_zz = t4; s_xy = t5; s_xz = t6; s_yz = t7; }
        }
        bool fresh = false;
        if (active) {
            // fewer than 20 particles in all, or inside the initial sphere: no tensor (:57-58, :106-107)
            if (iter == 0 && (nall < 20.0 || s_n < 20.0)) { valid = false; active = false; }
            else if (!(s_w != 0.0)) { for (int k = 0; k < 6; k++) T[k] = 0.0; active = false; }
            else {
                T[0] = s_xx / s_w; T[1] = s_yy / s_w; T[2] = s_zz / s_w; T[3] = s_xy / s_w; T[4] = s_xz / s_w; T[5] = s_yz / s_w;
                fresh = true;
            }
        }
        {
            // (the eigenproblems of all lanes together: lanes without a fresh tensor carry theirs along and keep their state)
            double ev[3], Vn[9];
            small_eigh3(T, ev, Vn);
            if (fresh) {
                double *S = g_W.it_par[LANE];
                for (int k = 0; k < 9; k++) S[k] = Vn[k];
                ev0 = fabs(ev[0]); ev1 = fabs(ev[1]); ev2 = fabs(ev[2]);
                if (q == 0.0) { for (int k = 0; k < 6; k++) T[k] = 0.0; active = false; }   // :127-129
            }
        }
        if (iter == 0) { for (int k = 0; k < 6; k++) T0[k] = T[k]; valid0 = valid; }
        WSYNC();
    }
    // columns: mode * 12 + reduced * 6; the dark matter tensors of a dark-matter-only chunk are the total ones
    if (mine) {
        for (int k = 0; k < 6; k++) {
            const double val = (valid && has) ? T[k] : 0.0, val0 = (valid0 && has) ? T0[k] : 0.0;
            row[out + k] = val;
            row[out + 12 + k] = val0;
            if (has) { row[out + 24 + k] = val; row[out + 36 + k] = val0; }
        }
    }
}
#endif   // SINERTIA

#if SHYDRO
// ======================= ordered rotation of gas / stars / baryons (rotation_pass of halo_kernel.inc on one wave) ========
// kappa_corot and the disc fractions (kinematic_properties.py:266-425): per flagged calculation one sweep over its members
// with lanes = particles; the components' centre-of-mass velocities were parked by finalize_moments, their angular momenta
// are in the row.
__device__ __forceinline__ void small_rotation(GAS double *row, const soap_calc_desc &cd, int coloff, const CalcState &cs, int ncalc) {
    unsigned long long todo = __ballot(LANE < ncalc && (cd.flags & SOAP_CALC_FLAG_ROTATION) && cs.state == 1 && cd.kind != SOAP_CALC_PROJECTED);
    if (!todo) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the rows were written by other lanes
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    WSYNC();
#pragma unroll 1
    while (todo) {
        const int c2 = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int kind = rl(cd.kind, c2), n_in = rl(cs.n, c2);
        int cMg, cMs, cNg, cNs, cLg, cLs, cLb, cKap, cDt;
        bool bound_only = false;
        if (kind == SOAP_CALC_SUBHALO) { cNg = SH_Ngas; cNs = SH_Nstar; cMg = SH_Mgas; cMs = SH_Mstar; cLg = SH_Lgas; cLs = SH_Lstar; cLb = SH_Lbaryons; cKap = SH_kappa; cDt = SH_DtoT; bound_only = true; }
        else if (kind == SOAP_CALC_APERTURE_INCL || kind == SOAP_CALC_APERTURE_EXCL) {
            cNg = AP_Ngas; cNs = AP_Nstar; cMg = AP_Mgas; cMs = AP_Mstar; cLg = AP_Lgas; cLs = AP_Lstar; cLb = AP_Lbaryons; cKap = AP_kappa; cDt = AP_DtoT;
            bound_only = kind == SOAP_CALC_APERTURE_EXCL;
        } else { cNg = SOC_Ngas; cNs = SOC_Nstar; cMg = SOC_Mgas; cMs = SOC_Mstar; cLg = SOC_Lgas; cLs = SOC_Lstar; cLb = SOC_Lbaryons; cKap = -1; cDt = SOC_DtoT; }
        GAS double *crow = row + rl(coloff, c2);
        // (everything this calculation reads of its row in ONE round trip: the loads are issued before the first value is looked at)
        const double Mg_ = crow[cMg], Ms_ = crow[cMs], Ng_ = crow[cNg], Ns_ = crow[cNs];
        double lv_[3][3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int cl = q == 0 ? cLg : (q == 1 ? cLs : cLb);
#pragma unroll
            for (int d = 0; d < 3; d++) lv_[q][d] = (q == 2 && cKap < 0) ? 0.0 : crow[cl + d];
        }
        const double Mg = uni(Mg_), Ms = uni(Ms_), Ng = uni(Ng_), Ns = uni(Ns_);
        if (Mg == 0.0 && Ms == 0.0) continue;
        // centre-of-mass velocities (slab frame), unit vectors along L and |L| of gas, stars, baryons
        double vc[3][3], hh[3][3], ln[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            if (q == 2 && cKap < 0) { ln[q] = 0.0; for (int d = 0; d < 3; d++) { hh[q][d] = 0.0; vc[q][d] = 0.0; } continue; }   // (no baryon pass without kappa_corot)
            const double lx = uni(lv_[q][0]), ly = uni(lv_[q][1]), lz = uni(lv_[q][2]);
            const double nn = sqrt((lx * lx + ly * ly) + lz * lz);
            ln[q] = nn;
            hh[q][0] = nn > 0.0 ? lx / nn : 0.0; hh[q][1] = nn > 0.0 ? ly / nn : 0.0; hh[q][2] = nn > 0.0 ? lz / nn : 0.0;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const double vg = g_W.rot_v[c2][d], vs = g_W.rot_v[c2][3 + d];
                vc[q][d] = q == 0 ? vg : (q == 1 ? vs : (Mg * vg + Ms * vs) / (Mg + Ms));
            }
        }
        double sg[3] = {0.0, 0.0, 0.0}, ss[3] = {0.0, 0.0, 0.0}, sb[2] = {0.0, 0.0};
#pragma unroll 1
        for (int base = 0; base < n_in; base += 64) {
            const int i = base + LANE;
            if (i >= n_in) continue;
            const int fl = g_W.flags[i];
            const int t = fl & FL_TYPE;
            if ((t != 0 && t != 2) || (bound_only && !(fl & FL_BOUND))) continue;
            const double m = CX(m)[i], x = CX(dx)[i], y = CX(dy)[i], z = CX(dz)[i];
            const double u0 = CX(vx)[i], u1 = CX(vy)[i], u2 = CX(vz)[i];
            if (cKap < 0) {
                // spherical overdensities: the disc fractions only -- the sign of L_i about the particle's own component (as in rotation_pass)
                const bool gas = t == 0;
                const double a = u0 - (gas ? vc[0][0] : vc[1][0]), b = u1 - (gas ? vc[0][1] : vc[1][1]), w = u2 - (gas ? vc[0][2] : vc[1][2]);
                const double h0 = gas ? hh[0][0] : hh[1][0], h1 = gas ? hh[0][1] : hh[1][1], h2 = gas ? hh[0][2] : hh[1][2];
                const double l0 = m * (y * w - z * b), l1 = m * (z * a - x * w), l2 = m * (x * b - y * a);
                const double Li = (l0 * h0 + l1 * h1) + l2 * h2;
                const double mc = Li < 0.0 ? m : 0.0;
                if (gas) sg[2] += mc; else ss[2] += mc;
                continue;
            }
            const double r2 = (x * x + y * y) + z * z;
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
                const int q = pass == 0 ? (t == 0 ? 0 : 1) : 2;
                const double v0 = q == 0 ? vc[0][0] : (q == 1 ? vc[1][0] : vc[2][0]), v1 = q == 0 ? vc[0][1] : (q == 1 ? vc[1][1] : vc[2][1]),
                             v2 = q == 0 ? vc[0][2] : (q == 1 ? vc[1][2] : vc[2][2]);
                const double h0 = q == 0 ? hh[0][0] : (q == 1 ? hh[1][0] : hh[2][0]), h1 = q == 0 ? hh[0][1] : (q == 1 ? hh[1][1] : hh[2][1]),
                             h2 = q == 0 ? hh[0][2] : (q == 1 ? hh[1][2] : hh[2][2]);
                const double a = u0 - v0, b = u1 - v1, w = u2 - v2;
                const double l0 = m * (y * w - z * b), l1 = m * (z * a - x * w), l2 = m * (x * b - y * a);
                const double Li = (l0 * h0 + l1 * h1) + l2 * h2;
                const double rdotL = (x * h0 + y * h1) + z * h2;
                const double Ri2 = r2 - rdotL * rdotL;
                const double ke = (m * (a * a) + m * (b * b)) + m * (w * w);
                const double kco = (Ri2 != 0.0 && Li > 0.0) ? 0.5 * ((Li * Li) / (m * Ri2)) : 0.0;
                const double mc = Li < 0.0 ? m : 0.0;
                if (pass == 1) { sb[0] += ke; sb[1] += kco; }
                else if (t == 0) { sg[0] += ke; sg[1] += kco; sg[2] += mc; }
                else { ss[0] += ke; ss[1] += kco; ss[2] += mc; }
            }
        }
        const double p1 = wave_sum4(sg[0], sg[1], sg[2], ss[0]), p2 = wave_sum4(ss[1], ss[2], sb[0], sb[1]);
        const double t0 = rl64(p1, 0), t1 = rl64(p1, 16), t2 = rl64(p1, 32), t3 = rl64(p1, 48);
        const double t4 = rl64(p2, 0), t5 = rl64(p2, 16), t6 = rl64(p2, 32), t7 = rl64(p2, 48);
        if (LANE == 0) {
            // a single member moves with its own centre-of-mass velocity: L = 0 exactly in the reference
            const double cnt[3] = {Ng, Ns, Ng + Ns};
            const double Mq[3] = {Mg, Ms, Mg + Ms};
            const double Kq[3] = {0.5 * t0, 0.5 * t3, 0.5 * t6};
            const double Cq[3] = {t1, t4, t7};
            const double Cm[3] = {t2, t5, 0.0};
#pragma unroll
            for (int q = 0; q < 3; q++) {
                if (Mq[q] == 0.0) continue;
                const bool rot = cnt[q] > 1.0 && ln[q] > 0.0;
                if (cKap >= 0) crow[cKap + q] = (rot && Kq[q] > 0.0) ? Cq[q] / Kq[q] : 0.0;
                if (q < 2) crow[cDt + q] = 1.0 - 2.0 * (rot ? Cm[q] : 0.0) / Mq[q];
            }
        }
    }
}

// ======================= projected apertures (projected_aperture_properties.py:98-198; projected_apertures of halo_kernel.inc) =====
// Bound members only.  Per projection axis the members inside the largest requested aperture are ordered by their 2D radius
// (register bitonic network on (radius, slab index), as the build's), so the apertures are nested prefixes: prefix sums by
// wave scans with the lane that holds a prefix' last member storing them, finalisation with lanes = apertures, then one scan
// for the half-mass radii in projection (:1491-1577).  The sorted radii / slab indices live in the cum / pidx columns of the
// LDS slab, both dead after the radius loop.
enum { SPJ_N0 = 0, SPJ_M0 = 4, SPJ_MX = 8, SPJ_MV = 11, SPJ_VL = 14, SPJ_VL2 = 15, SPJ_SMX = 16, SPJ_VLG = 19, SPJ_VL2G = 20, SPJ_VLS = 21, SPJ_VL2S = 22 };
__device__ __forceinline__ void small_projected(GAS double *row, const soap_calc_desc &cd, int coloff, const CalcState &cs, int ncalc, int nsub) {
    const bool isp = LANE < ncalc && cd.kind == SOAP_CALC_PROJECTED && cs.state == 1;
    const unsigned long long pm = __ballot(isp);
    if (!pm) return;
    const double myR = isp ? cs.R : 0.0;
    const double Rmax = wave_max(myR);
    // position of this lane's aperture in order of radius
    int rank = 0;
    {
        unsigned long long rest = pm;
        while (rest) {
            const int c2 = __ffsll((long long)rest) - 1;
            rest &= rest - 1;
            const double R2 = rl64(myR, c2);
            if (R2 < myR || (R2 == myR && c2 < LANE)) rank++;
        }
    }
    int lg = 1;
    while ((1 << lg) < nsub) lg++;
#pragma unroll 1
    for (int axis = 0; axis < 3; axis++) {
        double key[SE];
        unsigned idx[SE], dmy[SE];
#pragma unroll
        for (int e = 0; e < SE; e++) {
            const int i = 64 * e + LANE;
            key[e] = INFINITY; idx[e] = (unsigned)i; dmy[e] = 0u;
            if (i < nsub && (g_W.flags[i] & FL_BOUND)) {
                const double x = CX(dx)[i], y = CX(dy)[i], z = CX(dz)[i];
                const double rp = axis == 0 ? sqrt(y * y + z * z) : (axis == 1 ? sqrt(x * x + z * z) : sqrt(x * x + y * y));
                if (rp <= Rmax) key[e] = rp;
            }
        }
        int nmem = 0;
#pragma unroll
        for (int e = 0; e < SE; e++) nmem += __popcll(__ballot(key[e] < INFINITY));
        if (nmem == 0) continue;
        small_sort(key, idx, dmy, lg);
        WSYNC();
#pragma unroll
        for (int e = 0; e < SE; e++) { g_W.cum[64 * e + LANE] = key[e]; g_W.pidx[64 * e + LANE] = idx[e]; }
        // members with rp <= R of every aperture (projected_aperture_properties.py:140-142)
        int my_nc = 0;
        {
            unsigned long long rest = pm;
            while (rest) {
                const int c2 = __ffsll((long long)rest) - 1;
                rest &= rest - 1;
                const double Rc = rl64(myR, c2);
                int cnt = 0;
#pragma unroll
                for (int e = 0; e < SE; e++) cnt += __popcll(__ballot(key[e] <= Rc));
                if (LANE == c2) my_nc = cnt;
            }
        }
        for (int e2 = LANE; e2 < SMAXPJ * 24; e2 += 64) (&g_W.pj[0][0])[e2] = 0.0;
        WSYNC();
        const int nmax = nmem;
        // ---- prefix sums in two sweeps of 12 + 11 running sums
#define SPJ_SWEEP(NL, FILL, STORE)                                                                                             \
        {                                                                                                                        \
            double carry[NL];                                                                                                    \
            _Pragma("unroll") for (int a = 0; a < NL; a++) carry[a] = 0.0;                                                       \
            _Pragma("unroll 1") for (int base = 0; base < nmax; base += 64) {                                                    \
                const int p_ = base + LANE;                                                                                      \
                double c[NL];                                                                                                    \
                _Pragma("unroll") for (int a = 0; a < NL; a++) c[a] = 0.0;                                                       \
                if (p_ < nmax) {                                                                                                 \
                    const int i = (int)g_W.pidx[p_];                                                                             \
                    const int t = g_W.flags[i] & FL_TYPE;                                                                        \
                    const double mm = CX(m)[i];                                                                                  \
                    FILL                                                                                                         \
                }                                                                                                                \
                _Pragma("unroll") for (int a = 0; a < NL; a++) { c[a] = wave_scan(c[a], LANE) + carry[a]; if ((a & 3) == 3) __builtin_amdgcn_sched_barrier(0); } \
                {                                                                                                                \
                    unsigned long long rest = pm;                                                                                \
                    while (rest) {                                                                                               \
                        const int c2 = __ffsll((long long)rest) - 1;                                                             \
                        rest &= rest - 1;                                                                                        \
                        const int nc = rl(my_nc, c2), rk = rl(rank, c2);                                                         \
                        if (nc > base && nc <= base + 64 && LANE == nc - 1 - base) { double *dst = g_W.pj[rk]; STORE }           \
                    }                                                                                                            \
                }                                                                                                                \
                _Pragma("unroll") for (int a = 0; a < NL; a++) carry[a] = rl64(c[a], 63);                                        \
            }                                                                                                                    \
        }                                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);
#define SW_FILL_ { const double x = CX(dx)[i], y = CX(dy)[i], z = CX(dz)[i]; \
              const double u = CX(vx)[i], v = CX(vy)[i], w = CX(vz)[i]; \
              const double vl = axis == 0 ? u : (axis == 1 ? v : w); \
              c[0] = t == 0 ? 1.0 : 0.0; c[1] = t == 1 ? 1.0 : 0.0; c[2] = t == 2 ? 1.0 : 0.0; c[3] = t == 3 ? 1.0 : 0.0; \
              c[4] = t == 0 ? mm : 0.0; c[5] = t == 1 ? mm : 0.0; c[6] = t == 2 ? mm : 0.0; c[7] = t == 3 ? mm : 0.0; \
              c[8] = mm * x; c[9] = mm * y; c[10] = mm * z; c[11] = t == 1 ? mm * vl : 0.0; }
#define SW_STORE_ { for (int d = 0; d < 8; d++) dst[SPJ_N0 + d] = c[d]; \
              dst[SPJ_MX] = c[8]; dst[SPJ_MX + 1] = c[9]; dst[SPJ_MX + 2] = c[10]; dst[SPJ_VL] = c[11]; }
        SPJ_SWEEP(12, SW_FILL_, SW_STORE_)
#undef SW_FILL_
#undef SW_STORE_
#define SW_FILL_ { const double x = CX(dx)[i], y = CX(dy)[i], z = CX(dz)[i]; \
              const double u = CX(vx)[i], v = CX(vy)[i], w = CX(vz)[i]; \
              const double vl = axis == 0 ? u : (axis == 1 ? v : w); \
              c[0] = mm * u; c[1] = mm * v; c[2] = mm * w; c[3] = t == 1 ? mm * vl * vl : 0.0; \
              c[4] = t == 2 ? mm * x : 0.0; c[5] = t == 2 ? mm * y : 0.0; c[6] = t == 2 ? mm * z : 0.0; \
              c[7] = t == 0 ? mm * vl : 0.0; c[8] = t == 0 ? mm * vl * vl : 0.0; c[9] = t == 2 ? mm * vl : 0.0; c[10] = t == 2 ? mm * vl * vl : 0.0; }
#define SW_STORE_ { dst[SPJ_MV] = c[0]; dst[SPJ_MV + 1] = c[1]; dst[SPJ_MV + 2] = c[2]; dst[SPJ_VL2] = c[3]; \
              dst[SPJ_SMX] = c[4]; dst[SPJ_SMX + 1] = c[5]; dst[SPJ_SMX + 2] = c[6]; \
              dst[SPJ_VLG] = c[7]; dst[SPJ_VL2G] = c[8]; dst[SPJ_VLS] = c[9]; dst[SPJ_VL2S] = c[10]; }
        SPJ_SWEEP(11, SW_FILL_, SW_STORE_)
#undef SW_FILL_
#undef SW_STORE_
#undef SPJ_SWEEP
        WSYNC();
        // ---- lanes = apertures: the columns that follow from the sums, and the half-mass targets [1] dark matter, [2] gas, [3] stars, [4] baryons
        double tg[NHM] = {0.0, 0.0, 0.0, 0.0, 0.0};
        if (isp) {
            const volatile double *a = g_W.pj[rank];
            GAS double *prow = row + coloff + axis * PJ_NCOLS;
            const double Mtot = ((a[SPJ_M0] + a[SPJ_M0 + 1]) + a[SPJ_M0 + 2]) + a[SPJ_M0 + 3];
            prow[PJ_Mtot] = Mtot;
            prow[PJ_Mgas] = a[SPJ_M0]; prow[PJ_Mdm] = a[SPJ_M0 + 1]; prow[PJ_Mstar] = a[SPJ_M0 + 2]; prow[PJ_Mbh] = a[SPJ_M0 + 3];
            prow[PJ_Ngas] = a[SPJ_N0]; prow[PJ_Ndm] = a[SPJ_N0 + 1]; prow[PJ_Nstar] = a[SPJ_N0 + 2]; prow[PJ_Nbh] = a[SPJ_N0 + 3];
            if (Mtot != 0.0) {
                write_com(prow + PJ_com, a + SPJ_MX, Mtot, g_S.c, g_S.L);
                for (int d = 0; d < 3; d++) prow[PJ_vcom + d] = a[SPJ_MV + d] / Mtot + g_S.shift[d];
            }
            const double Md = a[SPJ_M0 + 1], Mg = a[SPJ_M0], Ms = a[SPJ_M0 + 2];
            if (Md != 0.0) {
                // sqrt(sum w (v - <v>)^2), shifted form (projected_aperture_properties.py:946-957); a single particle has no dispersion
                const double d1 = a[SPJ_VL] / Md, var = a[SPJ_VL2] / Md - d1 * d1;
                prow[PJ_veldisp_dm] = (a[SPJ_N0 + 1] == 1.0) ? 0.0 : sqrt(var > 0.0 ? var : 0.0);
            }
            if (Ms != 0.0) write_com(prow + PJ_com_star, a + SPJ_SMX, Ms, g_S.c, g_S.L);
            if (Mg != 0.0) {
                const double d1 = a[SPJ_VLG] / Mg, var = a[SPJ_VL2G] / Mg - d1 * d1;
                prow[PJ_veldisp_gas] = (a[SPJ_N0] == 1.0) ? 0.0 : sqrt(var > 0.0 ? var : 0.0);
            }
            if (Ms != 0.0) {
                const double d1 = a[SPJ_VLS] / Ms, var = a[SPJ_VL2S] / Ms - d1 * d1;
                prow[PJ_veldisp_star] = (a[SPJ_N0 + 2] == 1.0) ? 0.0 : sqrt(var > 0.0 ? var : 0.0);
            }
            tg[1] = 0.5 * Md; tg[2] = 0.5 * Mg; tg[3] = 0.5 * Ms; tg[4] = 0.5 * (Mg + Ms);
        }
        // ---- half-mass radii in projection: scan of the dark matter / gas / stellar mass in 2D-radius order
        const int scan_n = rl(my_nc, __ffsll((long long)__ballot(isp && rank == __popcll(pm) - 1)) - 1);   // the largest aperture's members
        double carryD = 0.0, carryG = 0.0, carryT = 0.0;
#define SPJ_CLS(w, tt) (((w) == 1 && (tt) == 1) || ((w) == 2 && (tt) == 0) || ((w) == 3 && (tt) == 2) || ((w) == 4 && ((tt) == 0 || (tt) == 2)))
#pragma unroll 1
        for (int base = 0; base < scan_n; base += 64) {
            const int p_ = base + LANE;
            int t = -1;
            double mm = 0.0, rp = 0.0;
            if (p_ < scan_n) {
                const int i = (int)g_W.pidx[p_];
                t = g_W.flags[i] & FL_TYPE;
                mm = CX(m)[i]; rp = g_W.cum[p_];
            }
            const double sD = wave_scan(t == 1 ? mm : 0.0, LANE) + carryD, sG = wave_scan(t == 0 ? mm : 0.0, LANE) + carryG,
                         sT = wave_scan(t == 2 ? mm : 0.0, LANE) + carryT;
            const double shD = dpp_f64<DPP_WAVE_SHR1>(sD), shG = dpp_f64<DPP_WAVE_SHR1>(sG), shT = dpp_f64<DPP_WAVE_SHR1>(sT);
            const double pD = LANE == 0 ? carryD : shD, pG = LANE == 0 ? carryG : shG, pT = LANE == 0 ? carryT : shT;
#pragma unroll
            for (int w = 1; w < NHM; w++) {
                unsigned long long open = __ballot(isp && tg[w] > 0.0 && my_nc > base);
                while (open) {
                    const int c2 = __ffsll((long long)open) - 1;
                    open &= open - 1;
                    const double tgt = rl64(tg[w], c2);
                    const int n2 = rl(my_nc, c2);
                    const double cv = w == 1 ? sD : (w == 2 ? sG : (w == 3 ? sT : sG + sT));
                    const double pv = w == 1 ? pD : (w == 2 ? pG : (w == 3 ? pT : pG + pT));
                    const unsigned long long hit = __ballot(t >= 0 && SPJ_CLS(w, t) && p_ < n2 && cv >= tgt);
                    if (!hit) continue;
                    const int owner = __ffsll((long long)hit) - 1;
                    const double cv_o = rl64(cv, owner), r_o = rl64(rp, owner);
                    double Wmin = rl64(pv, owner), rmin = 0.0;
                    int j = base + owner - 1;   // previous member of the same class
                    while (j >= 0) {
                        const int tj = g_W.flags[g_W.pidx[j]] & FL_TYPE;
                        if (SPJ_CLS(w, tj)) break;
                        j--;
                    }
                    if (j < 0) Wmin = 0.0; else rmin = g_W.cum[j];
                    const double h = (Wmin == cv_o) ? 0.5 * (rmin + r_o) : rmin + (tgt - Wmin) / (cv_o - Wmin) * (r_o - rmin);
                    const int col = w == 1 ? PJ_HMR_dm : (w == 2 ? PJ_HMR_gas : (w == 3 ? PJ_HMR_star : PJ_HMR_baryon));
                    if (LANE == c2) { row[coloff + axis * PJ_NCOLS + col] = h; tg[w] = 0.0; }   // found: the target is closed
                }
            }
            carryD = rl64(sD, 63); carryG = rl64(sG, 63); carryT = rl64(sT, 63);
        }
#undef SPJ_CLS
        WSYNC();
    }
}
#endif   // SHYDRO

// ======================= the kernel =======================================================================================
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SWPE, 8)))
SOAP_CAT4(k_small, SNS, SVAR, )(const RunParams *Pp, unsigned *queue, const int *todo, long long n_todo) {   // k_small128d / k_small512d: the variants show under their own names in a kernel trace
    const CAS RunParams &P = *(const CAS RunParams *)Pp;
    const int ncalc = P.ncalc;
    const long long ncols = P.ncols;
    // ---- this lane's calculation: kind, flags and virial definition stay in a register, the rest of the descriptor is
    // fetched again where the radius loop evaluates the calculation (a handful of cached loads per halo against ten
    // registers held through every phase)
    int kfv = -1, coloff = 0;
    if (LANE < ncalc) {
        const GAS soap_calc_desc *g = (const GAS soap_calc_desc *)P.calcs + LANE;
        kfv = (g->kind & 0xff) | ((g->flags & 0xff) << 8) | ((g->virial_definition & 0xff) << 16);
        coloff = ((const GAS int *)P.coloff)[LANE];
    }
    soap_calc_desc cd;   // kind / flags / virial definition of this lane's calculation (what stage B needs)
    cd.kind = kfv < 0 ? -1 : (kfv & 0xff); cd.flags = kfv < 0 ? 0 : ((kfv >> 8) & 0xff); cd.virial_definition = kfv < 0 ? 0 : ((kfv >> 16) & 0xff);
    cd.parent = -1; cd.reference_density = 0.0; cd.radius = 0.0; cd.physical_radius_mpc = 0.0; cd.prev_radius = 0.0; cd.filter_limit = 0; cd.filter_types = 0;
    const double subhalo_required = ((const CAS soap_calc_desc *)P.calcs)[0].physical_radius_mpc;
    const unsigned long long so_lanes = __ballot(LANE < ncalc && cd.kind == SOAP_CALC_SO_DENSITY);
    const unsigned long long flow_lanes = __ballot(LANE < ncalc && cd.kind == SOAP_CALC_SO_DENSITY && (cd.flags & SOAP_CALC_FLAG_DM_FLOW));
    if (LANE == 0) {
        g_W.L = P.boxsize; g_W.nu_density = P.nu_density; g_W.a = P.a; g_W.mpc = P.mpc; g_W.mpc20 = P.mpc20; g_W.G_eff = P.G_eff;
        g_W.hydro = P.hydro; g_W.has_nu = P.has_nu; g_W.mass_f32 = P.mass_f32;
#if SGLOBAL
        g_W.cols = (GAS double *)(SNS > 512 ? P.small_cols2 : (SNS > 128 ? P.small_cols : P.small_cols0)) + (long long)blockIdx.x * (SC_NCOLS * SNS);
#endif
#ifdef SMALL_CHECK
        g_W.dbg = (unsigned long long *)P.phase;
#endif
    }
    WSYNC();
    const double soft[4] = {P.soft[0], P.soft[1], P.soft[2], P.soft[3]};   // (SHYDRO 0: only the dark matter's is used)
    const int dbg_stop = P.profile >= 100 ? P.profile - 100 : 0;   // diagnostics: SOAP_ENGINE_PROFILE=100+k ends every halo after phase k
    if (dbg_stop == 9) return;
    const double target_density = P.target_density, first_factor = P.first_factor;
    const GAS double *h_cofp = (const GAS double *)P.halo.cofp;
    GAS double *h_search = (GAS double *)P.halo.search_radius, *h_read = (GAS double *)P.halo.read_radius;
    const GAS long long *h_index = (const GAS long long *)P.halo.index, *h_nrbound = (const GAS long long *)P.halo.nr_bound;
    const GAS int *h_central = (const GAS int *)P.halo.is_central;
    const GAS double *h_resident = (const GAS double *)P.halo.resident;
    GAS int *h_status = (GAS int *)P.halo.status;
    GAS long long *h_stats = (GAS long long *)P.halo.stats;
    GAS double *table = (GAS double *)P.table;
    GAS unsigned *gqueue = (GAS unsigned *)queue;
    const GAS int *gtodo = (const GAS int *)todo;
    const unsigned long long alldone = (ncalc >= 64) ? ~0ull : ((1ull << ncalc) - 1ull);

#pragma unroll 1
    while (true) {
        // (the ticket travels through LDS, as in halo_kernel.inc: a value defined in lane 0's branch and read back with
        // readfirstlane was compiled into a loop exit that only lane 0 takes -- the other lanes spun on ticket 0)
        WSYNC();
        if (LANE == 0) g_W.ticket = atomicAdd((unsigned *)gqueue, 1u);
        WSYNC();
        const long long task = (long long)(unsigned)uni((int)g_W.ticket);
        if (task >= n_todo) break;
        const long long hrow = (long long)uni(gtodo[task]);
        if (!SCHECK(hrow >= 0 && hrow < P.halo.n, 0, hrow)) continue;
        const unsigned long long t_begin = __builtin_readcyclecounter();
        const double c0 = uni(h_cofp[3 * hrow]), c1 = uni(h_cofp[3 * hrow + 1]), c2 = uni(h_cofp[3 * hrow + 2]);
        const long long hindex = uni(h_index[hrow]);
        double Rread = uni(h_read[hrow]);
        double Rsearch = uni(h_search[hrow]);
        const double Rresident = h_resident ? uni(h_resident[hrow]) : Rread;
        const int central = uni(h_central[hrow]);
        const long long nr_bound = uni(h_nrbound[hrow]);
        WSYNC();
        if (LANE == 0) { g_W.c[0] = c0; g_W.c[1] = c1; g_W.c[2] = c2; }
        WSYNC();
        const bool use_target = central == 1 && target_density > 0.0;
        GAS double *row = table + hrow * ncols;
        // (columns nobody writes stay zero: the table was cleared by one memset when the run began; nothing is written before the property passes)
        if (dbg_stop == 8) { if (LANE == 0) h_status[hrow] = SOAP_HALO_OK; continue; }
#ifdef SMALL_STUB8
        { if (LANE == 0) h_status[hrow] = SOAP_HALO_OK; continue; }
#endif
        CalcState cs;
        cs.state = 0; cs.n = 0; cs.ev = -1; cs.R = 0.0; cs.M = 0.0;
        double counts[4] = {0.0, 0.0, 0.0, 0.0};
        double encl = 0.0;
        int last_bound = -1;   // last slab entry bound to the halo: the passes over the bound members end there
        unsigned long long done = 0ull;
        double Rcur = Rsearch;
        int status = SOAP_HALO_OK;
        long long n_loop = 0, n_loop_all = 0, ncand = 0;
        int n_restart = 0, nsel_final = 0;
        double R_have = -1.0;
        int n_have = 0;
        double q[SE];
        long long fof0 = -1;
#pragma unroll
        for (int e = 0; e < SE; e++) q[e] = INFINITY;
#pragma unroll 1
        while (true) {
            n_loop++; n_loop_all++;
            if (!(Rcur <= Rread) || n_loop_all > 100000) { status = SOAP_HALO_FATAL_RADIUS_ASSERT; break; }   // (halo_tasks.py:78; the count: a radius that cannot grow, e.g. 0)
            // the sorted superset has to cover Rcur: built a few radius steps ahead of the loop, built again further out
            // when the loop outruns it (a rebuild from scratch: everything fits the registers and LDS of the wave)
            if (Rcur > R_have) {
                const double R_new = uni(fmin(Rread, fmax(Rcur, (R_have < 0.0 ? Rsearch * first_factor : R_have * 1.728))));
#ifndef NO_BUILD
                const int nb_ = small_build(P, R_new, hindex, q, &fof0);
#else
                const int nb_ = 5;
#endif
                if (nb_ < 0) { status = SOAP_HALO_FATAL_SLAB_OVERFLOW; break; }
                n_have = nb_;
                R_have = R_new;
                if (dbg_stop == 1) break;
            }
            // the sphere of this round: entries with minimum-image r^2 <= Rcur^2 (shared_mesh.py:191-194); they have to be a
            // prefix of the slab (they are, unless two particles straddle the radius within rounding: general path)
            int n = n_have;
            if (Rcur != R_have) {
                const double R2 = uni(Rcur * Rcur);
                int cnt = 0;
#pragma unroll
                for (int e = 0; e < SE; e++) cnt += __popcll(__ballot(64 * e + LANE < n_have && q[e] <= R2));
                int chk = 0;
#pragma unroll
                for (int e = 0; e < SE; e++) chk += __popcll(__ballot(64 * e + LANE < cnt && q[e] <= R2));
                if (chk != cnt) { status = SOAP_HALO_FATAL_SLAB_OVERFLOW; break; }
                n = cnt;
            }
            nsel_final = n;
            if (dbg_stop == 2) break;
            double required = 0.0;
            // density = mass_total / (4/3 pi R^3) <= target  (halo_tasks.py:97-103)
            const double mass_total = n > 0 ? g_W.cum[n - 1] : 0.0;
            const double density = uni(mass_total / (4.0 / 3.0 * M_PI * cube(Rcur)));
            if (!use_target || density <= target_density) {
                int fatal = 0;
                bool stop = false;
                if (!(done & 1ull)) {
                    // BoundSubhalo gate (subhalo_properties.py:2618-2646): bound-particle counts per type + enclose radius
                    int cb[4] = {0, 0, 0, 0};
                    int last = -1;
#pragma unroll
                    for (int e = 0; e < SE; e++) {
                        const int i = 64 * e + LANE;
                        const unsigned char fl = i < n ? g_W.flags[i] : (unsigned char)0;
                        const bool bd = (fl & FL_BOUND) != 0;
                        const unsigned long long bm = __ballot(bd);
                        if (bm) { const int hi = 64 * e + 63 - __clzll((long long)bm); last = hi > last ? hi : last; }
#if SHYDRO
#pragma unroll
                        for (int tt = 0; tt < 4; tt++) cb[tt] += __popcll(__ballot(bd && (fl & FL_TYPE) == tt));
#else
                        cb[1] += __popcll(bm);
#endif
                    }
                    const double tot[4] = {uni((double)cb[0]), uni((double)cb[1]), uni((double)cb[2]), uni((double)cb[3])};
                    const double Ntot = ((tot[0] + tot[1]) + tot[2]) + tot[3];
                    if (Ntot < (double)nr_bound) { required = uni(fmax(required, subhalo_required)); stop = true; }   // subhalo_properties.py:2634
                    else if (Ntot > (double)nr_bound) fatal = SOAP_HALO_FATAL_TOO_MANY_BOUND;                                            // :2642
                    else {
                        for (int k = 0; k < 4; k++) counts[k] = tot[k];
                        encl = uni(last >= 0 ? g_W.r[last] : 0.0);
                        last_bound = last;
                        if (LANE == 0) { cs.state = 1; cs.n = n; cs.R = 0.0; cs.M = 0.0; }
                        done |= 1ull;
                    }
                }
                if (!stop && !fatal && done != alldone) {
                    // nskip = max(1, argmax(r > 0))  (SO_properties.py:416): the radii are sorted, the zeros lead
                    int nskip = 1;
                    if (central && n > 0) {
                        int nz = 0;
#pragma unroll
                        for (int e = 0; e < SE; e++) { const int i = 64 * e + LANE; nz += __popcll(__ballot(i < n && !(g_W.r[i < n ? i : 0] > 0.0))); }
                        nskip = nz == n ? 0 : nz;
                        nskip = nskip < 1 ? 1 : nskip;
                    }
                    // first entry >= 1 whose enclosed density is not above the threshold of each SO density calculation
                    // (SO_properties.py:140-146), found with the lanes on the entries; kept by the calculation's lane
                    soap_calc_desc cdf = cd;   // the full descriptor of this lane's calculation
                    if (LANE < ncalc) {
                        const GAS soap_calc_desc *g = (const GAS soap_calc_desc *)P.calcs + LANE;
                        cdf.parent = g->parent; cdf.reference_density = g->reference_density; cdf.radius = g->radius;
                        cdf.physical_radius_mpc = g->physical_radius_mpc; cdf.prev_radius = g->prev_radius;
                        cdf.filter_limit = g->filter_limit; cdf.filter_types = g->filter_types;
                    }
                    int my_fb = -1;
                    {
                        unsigned long long pend = so_lanes & ~done;
                        if (pend && central) {
                            double de[SE];
#pragma unroll
                            for (int e = 0; e < SE; e++) { const int i = 64 * e + LANE; de[e] = (i >= 1 && i < n) ? small_dens(i) : INFINITY; }
                            while (pend) {
                                const int c2 = __ffsll((long long)pend) - 1;
                                pend &= pend - 1;
                                const double rho = rl64(cdf.reference_density, c2);
                                int first = 0x7fffffff;
#pragma unroll
                                for (int e = SE - 1; e >= 0; e--) {
                                    const unsigned long long bm = __ballot(!(de[e] > rho));
                                    if (bm) first = 64 * e + __ffsll((long long)bm) - 1;
                                }
                                if (LANE == c2) my_fb = first;
                            }
                        }
                    }
                    {
                        const int c = LANE;
                        const bool pending = c > 0 && c < ncalc && !((done >> c) & 1ull);
                        int res = RES_OK, st = 2, nin = 0;
                        double Rv = 0.0, Mv = 0.0, creq = 0.0;
                        bool filt = true;
                        // A selection found in an earlier round stays what it is when the sphere grows (it is a prefix of the
                        // sorted slab): such lanes only wait for the calculations before them.
                        const bool cached = pending && cs.ev >= 0;
                        if (cached) { st = cs.ev; nin = cs.n; Rv = cs.R; Mv = cs.M; creq = cdf.physical_radius_mpc; }
                        if (pending && !cached) {
                            creq = cdf.physical_radius_mpc;
                            // category filter of the calculation (category_filter.py:69-106)
                            if (cdf.filter_limit > 0) {
                                double v = 0.0;
                                for (int t = 0; t < 4; t++) if (cdf.filter_types & (1 << t)) v += counts[t];
                                filt = v >= (double)cdf.filter_limit;
                            }
                            if (cdf.kind == SOAP_CALC_SO_DENSITY || cdf.kind == SOAP_CALC_SO_PHYSICAL) {
                                #ifndef NO_SO
                                if (central && filt) res = so_eval_lane(cdf, Rcur, n, nskip, 0.0, my_fb, &creq, &st, &nin, &Rv, &Mv);
#endif
                            } else if (cdf.kind == SOAP_CALC_APERTURE_INCL || cdf.kind == SOAP_CALC_APERTURE_EXCL) {
                                bool skip = false;
                                int copy_from = -1;
                                const double Rap = cdf.radius;
                                if (cdf.parent >= 0) {
                                    // aperture_properties.py:4086-4123, EncloseRadius as stored (float32)
                                    const float enc = (float)encl;
                                    if (cdf.prev_radius > (double)enc) {
                                        skip = true;
                                        if (cdf.kind == SOAP_CALC_APERTURE_EXCL && filt) copy_from = cdf.parent;
                                    }
                                }
                                if (filt && !skip) {
                                    if (Rcur < Rap) res = RES_TOO_SMALL;   // aperture_properties.py:4140-4143
                                    else { st = 1; nin = count_r_le(n, Rap); Rv = Rap; }   // r <= R_ap (:310)
                                } else if (copy_from >= 0) { st = 3; nin = copy_from; }
#if SHYDRO
                            } else if (cdf.kind == SOAP_CALC_PROJECTED) {
                                // bound particles only: no search-radius check (projected_aperture_properties.py:1889-1891);
                                // skip / copy rule as for exclusive spheres (:1827-1884)
                                int copy_from = -1;
                                bool skip = false;
                                if (cdf.parent >= 0) {
                                    const float enc = (float)encl;
                                    if (cdf.prev_radius > (double)enc) { skip = true; if (filt) copy_from = cdf.parent; }
                                }
                                if (filt && !skip) { st = 1; Rv = cdf.radius; }
                                else if (copy_from >= 0) { st = 3; nin = copy_from; }
#endif
                            }
                        }
                        if (pending && !cached && res == RES_OK && (st == 1 || st == 3) && cdf.kind != SOAP_CALC_SO_MULTIPLE) {
                            cs.ev = st; cs.n = nin; cs.R = Rv; cs.M = Mv;
                        }
                        // DarkMatterMassFlowRate: the shell around R_SO reaches R + dR/2 with dR = 0.1 R; beyond the search radius the
                        // reference raises SearchRadiusTooSmallError (SO_properties.py:2846-2850), unless no dark matter lies
                        // inside R_SO (:2923).  Checked every round, also for a cached selection.
                        if (pending && res == RES_OK && st == 1 && cdf.kind == SOAP_CALC_SO_DENSITY && (cdf.flags & SOAP_CALC_FLAG_DM_FLOW)) {
                            const double dR = 0.1 * Rv;
                            if (Rv + (dR / 2) > Rcur) {
                                bool has_dm = false;
                                for (int i = 0; i < nin && !has_dm; i++) has_dm = (g_W.flags[i] & FL_TYPE) == 1;
                                if (has_dm) res = RES_TOO_SMALL;
                            }
                        }
                        // radius multiples need their parent: finished in an earlier round or evaluated just now (its lane)
                        const bool is_mult = pending && cdf.kind == SOAP_CALC_SO_MULTIPLE;
                        const int par = is_mult ? cdf.parent : 0;
                        const int p_res = __shfl(res, par, 64), p_st = __shfl(st, par, 64);
                        const double p_R = __shfl(Rv, par, 64);
                        const int p_cstate = __shfl(cs.state, par, 64);
                        const double p_cR = __shfl(cs.R, par, 64);
                        if (is_mult && central && filt) {
                            double rp;
                            if ((done >> par) & 1ull) rp = (p_cstate == 1) ? p_cR : 0.0;
                            else rp = (p_res == RES_OK && p_st == 1) ? p_R : 0.0;
#ifndef NO_SO2
                            res = so_eval_lane(cdf, Rcur, n, nskip, rp, -1, &creq, &st, &nin, &Rv, &Mv);
#endif
                        }
#if SINERTIA
                        // iterative inertia tensors: "if np.max(R) > search_radius: raise" (inertia_tensors.py:110-111) is reached when the
                        // sphere holds at least 20 particles, all of which then lie inside R
                        if (pending && res == RES_OK && st == 1 && (cdf.flags & SOAP_CALC_FLAG_INERTIA) && cdf.kind != SOAP_CALC_SUBHALO && Rv > Rcur && n >= 20) res = RES_TOO_SMALL;
#endif
                        const unsigned long long failm = __ballot(pending && res != RES_OK);
                        const int f = failm ? __ffsll((long long)failm) - 1 : 64;
                        const bool commit = pending && c < f;
                        // (ev keeps the size of the search sphere of the round in which the calculation was committed)
                        if (commit) { cs.state = st; cs.n = nin; cs.R = Rv; cs.M = Mv; cs.ev = n; }
                        done |= __ballot(commit);
                        const int fres = __shfl(res, f & 63, 64);
                        const double freq = __shfl(creq, f & 63, 64);
                        if (f < 64) {
                            if (fres == RES_TOO_SMALL) required = uni(fmax(required, freq));
                            else if (fres < 0) fatal = fres;
                        }
                    }
                }
                if (fatal) { status = fatal; break; }
                #ifndef NO_CC
                if (done == alldone) { ncand = small_count_candidates(P, Rcur); break; }
#else
                if (done == alldone) break;
#endif
            }
            // halo_tasks.py:167-187
            if (required > Rread || Rcur >= Rread) {
                Rsearch = uni(fmax(Rsearch, required > Rread ? required : Rcur));
                const double Rread_new = uni(fmax(Rread * 1.5, Rsearch));   // halo_tasks.py:393-402
                if (Rread_new <= Rresident && n_restart < 64) {
                    // the enlarged region is already resident: what the next process_halos call would do with the halo -- start
                    // again from the new search radius with every calculation open (halo_tasks.py:23-74) -- happens here
                    Rread = Rread_new;
                    n_restart++;
                    for (int k = 0; k < 4; k++) counts[k] = 0.0;
                    cs.state = 0; cs.ev = -1;
                    encl = 0.0; last_bound = -1;
                    done = 0ull;
                    Rcur = Rsearch;
                    n_loop = 0;
                    continue;
                }
                status = SOAP_HALO_NEED_LARGER_READ_RADIUS;
                break;
            } else {
                Rcur = fmin(Rcur * 1.2, Rread);
                Rcur = uni(fmax(Rcur, required));
            }
        }
        if (status == SOAP_HALO_OK && dbg_stop != 1 && dbg_stop != 2 && dbg_stop != 3) {
            // ---- stage B: every property of every calculation in a few fused passes over the sorted slab
            if (LANE == 0 && cs.state == 1) cs.n = last_bound + 1 < nsel_final ? last_bound + 1 : nsel_final;   // all bound particles lie inside any later sphere; the passes over them end at the last one
            const int kind = cd.kind;
            const bool isb = kind == SOAP_CALC_SUBHALO || kind == SOAP_CALC_APERTURE_EXCL;
            const bool fam = LANE < ncalc && cs.state == 1 && kind != SOAP_CALC_PROJECTED;
            const double nbound_total = uni(((counts[0] + counts[1]) + counts[2]) + counts[3]);
            int rank;
            // ---- all particles: SO + inclusive spheres (moments, radial scan, flow rates), then the bound particles: subhalo +
            // exclusive spheres (moments, radial scan, spin).  The two families write disjoint columns.
            {
                MomOut mo;
                double nfar, vs, rs;
                const int m = small_family_order(fam && !isb, cs.n, &rank);
#ifndef NO_MOM
                small_moments(row, false, cd, coloff, cs, fam && !isb, rank, m, soft, &nfar, &mo);
#endif
#ifndef NO_SCAN
                if (dbg_stop != 4)
                small_scan(row, false, cd, coloff, cs, fam && !isb, m, mo, soft, nfar, nbound_total, &vs, &rs);
#endif
#ifndef NO_FLOW
                if (dbg_stop != 4 && dbg_stop != 5) {
                // dark matter mass flow rates of the virial SO definitions that exist and hold dark matter (:2923)
                unsigned long long fl = flow_lanes & __ballot(cs.state == 1 && mo.N1 != 0.0);
                // the sphere's own centre-of-mass velocity in the frame of the slab velocities: row[SOC_vcom] - shift, as the
                // general path reads it back
                double vc_own[3];
                {
                    const double Mtot = ((mo.M[0] + mo.M[1]) + mo.M[2]) + mo.M[3];
                    const double iMt = 1.0 / Mtot;
                    for (int d = 0; d < 3; d++) vc_own[d] = Mtot != 0.0 ? (mo.MV[d] * iMt + g_S.shift[d]) - g_S.shift[d] : -g_S.shift[d];
                }
                while (fl) {
                    const int c2 = __ffsll((long long)fl) - 1;
                    fl &= fl - 1;
                    const double vo[3] = {rl64(vc_own[0], c2), rl64(vc_own[1], c2), rl64(vc_own[2], c2)};
                    small_flow_one(P, row + rl(coloff, c2), rl64(cs.R, c2), rl64(cs.M, c2), vo, nsel_final);
                }
                }
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
            if (dbg_stop == 0 || dbg_stop > 6) {
                MomOut mo;
                double nfar, vs, rs;
                const int m = small_family_order(fam && isb, cs.n, &rank);
#ifndef NO_MOM
                small_moments(row, true, cd, coloff, cs, fam && isb, rank, m, soft, &nfar, &mo);
#endif
#ifndef NO_SCAN
                small_scan(row, true, cd, coloff, cs, fam && isb, m, mo, soft, nfar, nbound_total, &vs, &rs);
                if (rl(cs.state, 0) == 1) {
                    const double Mt0 = rl64(((mo.M[0] + mo.M[1]) + mo.M[2]) + mo.M[3], 0);
                    const double mv0[3] = {rl64(mo.MV[0], 0), rl64(mo.MV[1], 0), rl64(mo.MV[2], 0)};
                    small_subhalo_spin(row + rl(coloff, 0), rl(cs.n, 0), rl64(vs, 0), rl64(rs, 0), Mt0, mv0);
                }
#endif
            }
#if SINERTIA
            __builtin_amdgcn_sched_barrier(0);
            small_inertia(row, cd.kind, cd.flags, coloff, cs.state, cs.n, cs.ev, cs.R, ncalc, nbound_total);
#endif
#if SHYDRO
            // (diagnostics: SOAP_ENGINE_PROFILE=110 everything but these two passes, 111 with the rotation pass)
            __builtin_amdgcn_sched_barrier(0);
            if (dbg_stop == 0 || dbg_stop == 11) small_rotation(row, cd, coloff, cs, ncalc);
            __builtin_amdgcn_sched_barrier(0);
            if (dbg_stop == 0 && rl(cs.state, 0) == 1) small_projected(row, cd, coloff, cs, ncalc, rl(cs.n, 0));
#endif
            // exclusive spheres that repeat the next-smaller one (aperture_properties.py:4117-4123): copies of its columns
            {
                unsigned long long cp = __ballot(LANE < ncalc && cs.state == 3);
                if (cp) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the parents' columns were written by other lanes
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    while (cp) {
                        const int c2 = __ffsll((long long)cp) - 1;
                        cp &= cp - 1;
                        const int par = rl(cs.n, c2);
                        const GAS double *prow = row + rl(coloff, par);
                        GAS double *crow = row + rl(coloff, c2);
                        // (chains: a copy of a copy reads what the earlier iteration of this loop wrote)
                        const int ncopy = rl(cd.kind, c2) == SOAP_CALC_PROJECTED ? 3 * PJ_NCOLS : AP_NCOLS;
                        for (int k0 = 0; k0 < ncopy; k0 += 64) {
                            double v = 0.0;
                            if (k0 + LANE < ncopy) v = prow[k0 + LANE];
                            if (k0 + LANE < ncopy) crow[k0 + LANE] = v;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    }
                }
            }
        }
        if (status == SOAP_HALO_FATAL_SLAB_OVERFLOW && LANE == 0) {
            // handed on: the wavefront class of halo_kernel.inc is launched behind this kernel with the list these rows go to
            GAS int *esc = (GAS int *)P.small_esc;
            if (esc != nullptr) { const int slot = atomicAdd((int *)esc, 1); esc[1 + slot] = (int)hrow; }
        }
        if (LANE == 0) {
            h_status[hrow] = status;
            if (status == SOAP_HALO_NEED_LARGER_READ_RADIUS) {
                h_read[hrow] = fmax(Rread * 1.5, Rsearch);   // halo_tasks.py:393-402
                h_search[hrow] = Rsearch;
            } else if (n_restart > 0 && status != SOAP_HALO_FATAL_SLAB_OVERFLOW) {
                h_read[hrow] = Rread;   // the radii the reference's halo_arrays hold after the retries
                h_search[hrow] = Rsearch;
            }
            h_stats[SOAP_STATS_COLUMNS * hrow + 4] = n_loop_all;
            h_stats[SOAP_STATS_COLUMNS * hrow + 5] = n_restart + 1;
            h_stats[SOAP_STATS_COLUMNS * hrow] = n_loop;
            h_stats[SOAP_STATS_COLUMNS * hrow + 1] = nsel_final;
            h_stats[SOAP_STATS_COLUMNS * hrow + 2] = ncand;
            h_stats[SOAP_STATS_COLUMNS * hrow + 3] = P.profile == 2 ? (long long)n_have : (long long)(__builtin_readcyclecounter() - t_begin);
        }
    }
}
}  // namespace

extern "C" void SOAP_CAT4(soap_launch_small_, SNS, SVAR, )(int groups, hipStream_t stream, const RunParams *params, unsigned *queue, const int *todo, long long n_todo) {
    hipLaunchKernelGGL(SOAP_CAT4(k_small, SNS, SVAR, ), dim3(groups), dim3(64), 0, stream, params, queue, todo, n_todo);
}
// 1: this variant keeps its streamed columns in a per-wave slice of global memory (engine.hip provides RunParams.small_cols*)
extern "C" int SOAP_CAT4(soap_small_global_, SNS, SVAR, )() { return SGLOBAL; }
