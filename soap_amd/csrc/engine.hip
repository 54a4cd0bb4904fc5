// MI355X (gfx950) halo-property engine: mesh build + per-halo selection/property kernels
// and the C ABI declared in include/soap_engine.h.
//
// Reference behaviour being replaced (paths relative to the SOAP checkout):
//   SOAP/core/shared_mesh.py:11-200            cell-list mesh build and periodic ball query
//   SOAP/core/halo_tasks.py:23-187, 276-430    search-radius loop and calculation scheduling
//   SOAP/particle_selection/SO_properties.py   :80-217, :313-513, :532-692, :857-913, :2725-2765
//   SOAP/particle_selection/aperture_properties.py :270-317, :1099-1281, :3469-3577, :4024-4244
//   SOAP/particle_selection/subhalo_properties.py  :128-164, :798-846, :982-1073, :2587-2740
//   SOAP/property_calculation/{half_mass_radius,kinematic_properties}.py
//
// Design (DESIGN.md has the long form):
//   * particles of each type are physically reordered into mesh-cell order (SoA in HBM) so the
//     cells a search sphere overlaps are a few contiguous runs -> coalesced reads;
//   * one 256-thread workgroup per halo, halos pulled from a device queue in the staged
//     (largest-first) order by a persistent grid;
//   * per halo: stream candidate runs, exact periodic distance test, compact the selected
//     particles into the workgroup's slab, bitonic sort by radius (LDS for <=4096 particles,
//     LDS+L2 above), ONE sequential fp64 cumulative-mass pass (bit-identical to
//     numpy.cumsum, so SO radii and hence SO membership are reproducible), then every
//     SO / aperture / subhalo calculation is a masked reduction + a masked scan over the
//     sorted slab (wavefront shuffles, LDS across the 4 waves).
//   * no MFMA anywhere: the path is bandwidth/latency bound fp64 vector work.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/soap_engine.h"
#include "device_math.h"
#include "layout.h"

#pragma clang fp contract(off)

#define BLK 256
#define NWAVE (BLK / 64)
#define CH 4096            // LDS sort chunk (elements)
#define PIDX_BITS 30
#define PIDX_MASK ((1u << PIDX_BITS) - 1u)
#define MAXCALC 48

// flags byte of a slab element
#define FL_TYPE 0x3
#define FL_BOUND 0x4
#define FL_SAT 0x8
#define FL_EXT 0x10

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHECK(x)                                                                        \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess)                                                              \
            return fail(-100, std::string(#x) + ": " + hipGetErrorString(e_));             \
    } while (0)

// ------------------------------------------------------------------------------------------
// device-side views
// ------------------------------------------------------------------------------------------
struct TypeDev {
    long long n;
    int res;
    int grnr64;
    const double *x, *y, *z, *vx, *vy, *vz, *m;
    const int *grnr32;
    const long long *grnr64p;
    const long long *fof;
    const unsigned *orig;
    const unsigned *cell_offset;  // res^3 + 1 entries
    double pos_min[3], pos_max[3], cell_size[3];
};

struct SlabDev {
    long long cap;
    double *r, *m, *dx, *dy, *dz, *vx, *vy, *vz, *cum;
    unsigned *pidx;
    unsigned char *flags;
};

struct HaloDev {
    long long n;
    const double *cofp;  // [n][3]
    double *search_radius, *read_radius;
    const long long *index;
    const int *is_central;
    const long long *nr_bound;
    int *status;
    long long *stats;  // [n][4]
    const int *todo;   // optional list of rows to process (retry pass), or null
    long long n_todo;
};

struct RunParams {
    TypeDev t[SOAP_NTYPES];
    SlabDev slab;
    HaloDev halo;
    double boxsize, G_eff, nu_density, mpc20, target_density, a, mpc;
    double soft[SOAP_NTYPES];
    int ncalc;
    long long ncols;
    double *table;
    unsigned *queue;
    unsigned long long *phase;   // [8] accumulated s_memtime cycles of thread 0 per phase
    const soap_calc_desc *calcs;  // device copy
    const int *coloff;            // device: column offset per calc
};

// ------------------------------------------------------------------------------------------
// wave / block primitives
// ------------------------------------------------------------------------------------------
SOAP_DEV double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
SOAP_DEV double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
SOAP_DEV long long wave_min_ll(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        long long u = __shfl_xor(v, o, 64);
        v = u < v ? u : v;
    }
    return v;
}
// inclusive scan inside a wave
SOAP_DEV double wave_scan(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        double u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}

// ------------------------------------------------------------------------------------------
// mesh build kernels  (shared_mesh.py:33-114)
// ------------------------------------------------------------------------------------------
__global__ void k_minmax(const double *pos, long long n, double *partial /*[grid][6]*/) {
    __shared__ double s[NWAVE][6];
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (long long i = blockIdx.x * (long long)BLK + threadIdx.x; i < n;
         i += (long long)gridDim.x * BLK) {
#pragma unroll
        for (int d = 0; d < 3; d++) {
            double v = pos[3 * i + d];
            mn[d] = fmin(mn[d], v);
            mx[d] = fmax(mx[d], v);
        }
    }
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        double a = -wave_max(-mn[d]);
        double b = wave_max(mx[d]);
        if (lane == 0) { s[w][d] = a; s[w][3 + d] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        int d = threadIdx.x;
        double v = s[0][d];
        for (int k = 1; k < NWAVE; k++) v = d < 3 ? fmin(v, s[k][d]) : fmax(v, s[k][d]);
        partial[blockIdx.x * 6 + d] = v;
    }
}

struct MeshGeom {
    double pos_min[3], cell_size[3];
    int res;
};

// cell index of every particle + histogram (shared_mesh.py:69-80)
__global__ void k_cellid(const double *pos, long long n, MeshGeom g, unsigned *cell,
                         unsigned *count) {
    for (long long i = blockIdx.x * (long long)BLK + threadIdx.x; i < n;
         i += (long long)gridDim.x * BLK) {
        int c[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            double q = floor((pos[3 * i + d] - g.pos_min[d]) / g.cell_size[d]);
            // .astype(np.int32) then np.clip(0, res-1)
            int ci = (q >= 2147483647.0) ? 2147483647 : (q <= -2147483648.0 ? (-2147483647 - 1) : (int)q);
            ci = ci < 0 ? 0 : (ci > g.res - 1 ? g.res - 1 : ci);
            c[d] = ci;
        }
        unsigned id = (unsigned)c[0] + (unsigned)g.res * (unsigned)c[1] +
                      (unsigned)g.res * (unsigned)g.res * (unsigned)c[2];
        cell[i] = id;
        atomicAdd(&count[id], 1u);
    }
}

// exclusive scan of `count` (length n) into `offset` (length n+1): three small kernels
#define SCAN_ITEMS 8
#define SCAN_TILE (BLK * SCAN_ITEMS)
__global__ void k_scan_tiles(const unsigned *in, long long n, unsigned *out, unsigned *tilesum) {
    __shared__ unsigned s[BLK];
    long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    unsigned v[SCAN_ITEMS], tot = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        tot += v[k];
    }
    s[threadIdx.x] = tot;
    __syncthreads();
    for (int o = 1; o < BLK; o <<= 1) {
        unsigned u = threadIdx.x >= o ? s[threadIdx.x - o] : 0u;
        __syncthreads();
        s[threadIdx.x] += u;
        __syncthreads();
    }
    unsigned excl = s[threadIdx.x] - tot;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = excl;
        excl += v[k];
    }
    if (threadIdx.x == BLK - 1) tilesum[blockIdx.x] = s[BLK - 1];
}
__global__ void k_scan_sums(unsigned *tilesum, long long ntiles) {
    // single block: sequential over chunks of BLK with carry
    __shared__ unsigned s[BLK];
    __shared__ unsigned carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (long long b = 0; b < ntiles; b += BLK) {
        long long i = b + threadIdx.x;
        unsigned v = i < ntiles ? tilesum[i] : 0u;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < BLK; o <<= 1) {
            unsigned u = threadIdx.x >= o ? s[threadIdx.x - o] : 0u;
            __syncthreads();
            s[threadIdx.x] += u;
            __syncthreads();
        }
        if (i < ntiles) tilesum[i] = carry + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 0) carry += s[BLK - 1];
        __syncthreads();
    }
}
__global__ void k_scan_add(unsigned *out, long long n, const unsigned *tilesum, unsigned total_n) {
    long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    unsigned add = tilesum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++)
        if (base + k < n) out[base + k] += add;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = total_n;
}

// place particle i at its cell's next free position (order inside a cell is arbitrary: it only
// feeds candidate streaming; everything order-sensitive happens after the radial sort)
__global__ void k_scatter(const unsigned *cell, long long n, const unsigned *offset,
                          unsigned *cursor, unsigned *sort_idx) {
    for (long long i = blockIdx.x * (long long)BLK + threadIdx.x; i < n;
         i += (long long)gridDim.x * BLK) {
        unsigned c = cell[i];
        unsigned r = atomicAdd(&cursor[c], 1u);
        sort_idx[offset[c] + r] = (unsigned)i;
    }
}

// physical reorder into cell order: AoS inputs -> SoA outputs
__global__ void k_reorder(const unsigned *sort_idx, long long n, const double *pos,
                          const double *vel, const double *mass, const void *grnr, int grnr_bytes,
                          const long long *fof, double *x, double *y, double *z, double *vx,
                          double *vy, double *vz, double *m, int *g32, long long *g64,
                          long long *f) {
    for (long long p = blockIdx.x * (long long)BLK + threadIdx.x; p < n;
         p += (long long)gridDim.x * BLK) {
        long long i = sort_idx[p];
        x[p] = pos[3 * i];
        y[p] = pos[3 * i + 1];
        z[p] = pos[3 * i + 2];
        vx[p] = vel[3 * i];
        vy[p] = vel[3 * i + 1];
        vz[p] = vel[3 * i + 2];
        m[p] = mass[i];
        if (grnr_bytes == 4) g32[p] = ((const int *)grnr)[i];
        else g64[p] = ((const long long *)grnr)[i];
        f[p] = fof[i];
    }
}

// ------------------------------------------------------------------------------------------
// per-halo kernel
// ------------------------------------------------------------------------------------------
#define NACC 40
enum {
    A_N0 = 0, A_N1, A_N2, A_N3,        // counts per type slot
    A_M0 = 4, A_M1, A_M2, A_M3,        // mass per type slot
    A_MX = 8,                          // sum m*x (3), all types
    A_MV = 11,                         // sum m*v' (3), all types
    A_MVD = 14,                        // sum m*v' (3), dark matter
    A_MXD = 17,                        // sum m*x (3), dark matter
    A_LD = 20,                         // sum m * x cross v' (3), dark matter
    A_VV = 23,                         // sum m v'_a v'_b (6), dark matter: xx yy zz xy xz yz
    A_LA = 29,                         // sum m * x cross v' (3), all types
    A_MR = 32,                         // sum m*r
    A_MRS = 33,                        // sum m*max(r, softening)
    A_MSAT = 34, A_MEXT = 35,          // mass bound to satellites / external
    A_NFAR = 36,                       // members with r > 1e-8 (np.isclose(r, 0) false)
    A_RMAX = 37                        // max r  (max-reduction)
};

struct Shared {
    double keys[CH];
    unsigned vals[CH];
    double red[NWAVE][NACC];
    double mom[NACC];
    double mom2[NACC];
    double wtot[3][NWAVE];
    unsigned long long cellmask[3][4];
    double bc[16];
    long long bi[8];
    int imin[2];
    int nsel;
    int flag;
};

#define PH_SELECT 0
#define PH_SORT 1
#define PH_MATERIALIZE 2
#define PH_REDUCE 3
#define PH_SCAN 4
#define PH_SO 5
#define PH_TOTAL 6

struct Ctx {
    const RunParams *P;
    Shared *S;
    int tid, lane, w;
    // halo
    double c[3], L;
    long long hindex;
    // slab (this workgroup's slice)
    double *r, *m, *dx, *dy, *dz, *vx, *vy, *vz, *cum;
    unsigned *pidx;
    unsigned char *flags;
    long long cap;
    int n;          // selected particles this round
    unsigned long long *phase;
    double shift[3];
};

struct PhaseTimer {
    unsigned long long t0;
    const Ctx &X;
    int ph;
    __device__ PhaseTimer(const Ctx &x, int p) : X(x), ph(p) { t0 = (x.tid == 0) ? __builtin_readcyclecounter() : 0ull; }
    __device__ ~PhaseTimer() { if (X.tid == 0) atomicAdd(&X.phase[ph], (unsigned long long)(__builtin_readcyclecounter() - t0)); }
};

// selection predicate over the sorted slab
struct Pred {
    int nmax;        // only elements [0, nmax)
    int bound_only;  // require FL_BOUND
};

SOAP_DEV bool member(const Ctx &X, const Pred &pr, int i, unsigned char fl) {
    return i < pr.nmax && (!pr.bound_only || (fl & FL_BOUND));
}

// block-wide reduction of the moment set over a predicate; totals land in `dst` (LDS, NACC
// doubles), valid for every thread after the trailing barrier.
__device__ __noinline__ void reduce_moments(const Ctx &X, const Pred &pr, double *dst) {
    PhaseTimer pt_(X, PH_REDUCE);
    double a[NACC];
#pragma unroll
    for (int k = 0; k < NACC; k++) a[k] = 0.0;
    const double *soft = X.P->soft;
    for (int i = X.tid; i < pr.nmax; i += BLK) {
        unsigned char fl = X.flags[i];
        if (!member(X, pr, i, fl)) continue;
        int t = fl & FL_TYPE;
        double m = X.m[i], r = X.r[i];
        double x = X.dx[i], y = X.dy[i], z = X.dz[i];
        double u = X.vx[i], v = X.vy[i], q = X.vz[i];
#pragma unroll
        for (int tt = 0; tt < 4; tt++) {
            a[A_N0 + tt] += (t == tt) ? 1.0 : 0.0;
            a[A_M0 + tt] += (t == tt) ? m : 0.0;
        }
        a[A_MX] += m * x; a[A_MX + 1] += m * y; a[A_MX + 2] += m * z;
        a[A_MV] += m * u; a[A_MV + 1] += m * v; a[A_MV + 2] += m * q;
        double lx = y * q - z * v, ly = z * u - x * q, lz = x * v - y * u;
        a[A_LA] += m * lx; a[A_LA + 1] += m * ly; a[A_LA + 2] += m * lz;
        if (t == 1) {
            a[A_MVD] += m * u; a[A_MVD + 1] += m * v; a[A_MVD + 2] += m * q;
            a[A_MXD] += m * x; a[A_MXD + 1] += m * y; a[A_MXD + 2] += m * z;
            a[A_LD] += m * lx; a[A_LD + 1] += m * ly; a[A_LD + 2] += m * lz;
            a[A_VV] += m * u * u; a[A_VV + 1] += m * v * v; a[A_VV + 2] += m * q * q;
            a[A_VV + 3] += m * u * v; a[A_VV + 4] += m * u * q; a[A_VV + 5] += m * v * q;
        }
        a[A_MR] += m * r;
        a[A_MRS] += m * fmax(soft[t], r);
        if (fl & FL_SAT) a[A_MSAT] += m;
        if (fl & FL_EXT) a[A_MEXT] += m;
        if (r > 1e-8) a[A_NFAR] += 1.0;
        a[A_RMAX] = fmax(a[A_RMAX], r);
    }
#pragma unroll
    for (int k = 0; k < NACC; k++) {
        double v = (k == A_RMAX) ? wave_max(a[k]) : wave_sum(a[k]);
        if (X.lane == 0) X.S->red[X.w][k] = v;
    }
    __syncthreads();
    if (X.tid < NACC) {
        int k = X.tid;
        double v = X.S->red[0][k];
        for (int ww = 1; ww < NWAVE; ww++)
            v = (k == A_RMAX) ? fmax(v, X.S->red[ww][k]) : v + X.S->red[ww][k];
        dst[k] = v;
    }
    __syncthreads();
}

// Result of the scan pass over one selection
struct ScanOut {
    double vmax_soft_val, vmax_soft_r;       // max of M(<r)/max(r,eps) and its radius
    double vmax_unsoft_val, vmax_unsoft_r;   // same on raw radii, first member skipped
    double hmr_all, hmr_dm;                  // half-mass radii (all members / dark matter)
};

// Masked inclusive scan of member masses in radial order.
//  - v_max (kinematic_properties.py:555-593): argmax over members of cum/r, first maximum
//  - half-mass radius (half_mass_radius.py:40-80): first member with cum >= target, linear
//    interpolation from the previous member
__device__ __noinline__ void scan_pass(const Ctx &X, const Pred &pr, double target_all, double target_dm,
                          bool allclose_unsoft, ScanOut *out) {
    PhaseTimer pt_(X, PH_SCAN);
    Shared *S = X.S;
    const double *soft = X.P->soft;
    double carryA = 0.0, carryB = 0.0, carryC = 0.0;
    double bestS = -1.0, bestSr = 0.0, bestU = -1.0, bestUr = 0.0;
    long long bestSi = 0x7fffffffffffffffLL, bestUi = 0x7fffffffffffffffLL;
    if (X.tid == 0) {
        S->bc[0] = 0.0; S->bc[1] = 0.0;   // hmr_all, hmr_dm
        S->imin[0] = 0x7fffffff; S->imin[1] = 0x7fffffff;
    }
    __syncthreads();
    int foundA = 0x7fffffff, foundB = 0x7fffffff;
    for (int base = 0; base < pr.nmax; base += BLK) {
        int i = base + X.tid;
        unsigned char fl = 0;
        bool mem = false;
        double m = 0.0, r = 0.0;
        if (i < pr.nmax) {
            fl = X.flags[i];
            mem = member(X, pr, i, fl);
            if (mem) { m = X.m[i]; r = X.r[i]; }
        }
        int t = fl & FL_TYPE;
        double vA = mem ? m : 0.0;
        double vB = (mem && t == 1) ? m : 0.0;
        double vC = mem ? 1.0 : 0.0;
        double sA = wave_scan(vA, X.lane), sB = wave_scan(vB, X.lane), sC = wave_scan(vC, X.lane);
        if (X.lane == 63) { S->wtot[0][X.w] = sA; S->wtot[1][X.w] = sB; S->wtot[2][X.w] = sC; }
        __syncthreads();
        // offsets: one running chain through the waves, so that "inclusive value of the previous
        // element" is the same number for every thread that looks at it
        double oA = 0, oB = 0, oC = 0, runA = carryA, runB = carryB, runC = carryC;
        for (int ww = 0; ww < NWAVE; ww++) {
            if (ww == X.w) { oA = runA; oB = runB; oC = runC; }
            runA += S->wtot[0][ww]; runB += S->wtot[1][ww]; runC += S->wtot[2][ww];
        }
        double cA = oA + sA, cB = oB + sB, cC = oC + sC;   // inclusive cumulative values
        double pA = __shfl_up(cA, 1, 64), pB = __shfl_up(cB, 1, 64);
        if (X.lane == 0) { pA = oA; pB = oB; }
        if (mem) {
            // v_max, softened radii
            double sr = fmax(soft[t], r);
            if (sr > 1e-8) {
                double val = cA / sr;
                if (val > bestS) { bestS = val; bestSr = sr; bestSi = i; }
            }
            // v_max, raw radii, nskip = 1
            double rank = cC - 1.0;   // members before this one
            if (rank >= 1.0 && (r > 1e-8 || allclose_unsoft) && r != 0.0) {
                double val = cA / r;
                if (val > bestU) { bestU = val; bestUr = r; bestUi = i; }
            }
            // half-mass crossing candidates: first member with cum >= target
            if (target_all > 0.0 && cA >= target_all && i < foundA) atomicMin(&S->imin[0], i);
            if (t == 1 && target_dm > 0.0 && cB >= target_dm && i < foundB) atomicMin(&S->imin[1], i);
        }
        __syncthreads();
        foundA = S->imin[0]; foundB = S->imin[1];
        if (mem && i == foundA && i >= base) {
            double Wmin = pA, rmin = 0.0;
            int j = i - 1;
            while (j >= 0 && !member(X, pr, j, X.flags[j])) j--;
            if (j < 0) Wmin = 0.0; else rmin = X.r[j];
            S->bc[0] = (Wmin == cA) ? 0.5 * (rmin + r)
                                    : rmin + (target_all - Wmin) / (cA - Wmin) * (r - rmin);
        }
        if (mem && t == 1 && i == foundB && i >= base) {
            double Wmin = pB, rmin = 0.0;
            int j = i - 1;
            while (j >= 0) {
                unsigned char fj = X.flags[j];
                if (member(X, pr, j, fj) && (fj & FL_TYPE) == 1) break;
                j--;
            }
            if (j < 0) Wmin = 0.0; else rmin = X.r[j];
            S->bc[1] = (Wmin == cB) ? 0.5 * (rmin + r)
                                    : rmin + (target_dm - Wmin) / (cB - Wmin) * (r - rmin);
        }
        carryA = runA; carryB = runB; carryC = runC;
    }
    __syncthreads();
    // block arg-max (first maximum = smallest index on ties)
    for (int pass = 0; pass < 2; pass++) {
        double v = pass == 0 ? bestS : bestU, rr = pass == 0 ? bestSr : bestUr;
        long long ii = pass == 0 ? bestSi : bestUi;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            double v2 = __shfl_xor(v, o, 64), r2 = __shfl_xor(rr, o, 64);
            long long i2 = __shfl_xor(ii, o, 64);
            if (v2 > v || (v2 == v && i2 < ii)) { v = v2; rr = r2; ii = i2; }
        }
        if (X.lane == 0) { S->red[X.w][0] = v; S->red[X.w][1] = rr; S->red[X.w][2] = (double)ii; }
        __syncthreads();
        if (X.tid == 0) {
            double bv = S->red[0][0], br = S->red[0][1], bi = S->red[0][2];
            for (int ww = 1; ww < NWAVE; ww++) {
                double v2 = S->red[ww][0], i2 = S->red[ww][2];
                if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; br = S->red[ww][1]; bi = i2; }
            }
            S->bc[2 + 2 * pass] = bv;
            S->bc[3 + 2 * pass] = br;
        }
        __syncthreads();
    }
    out->hmr_all = S->bc[0];
    out->hmr_dm = S->bc[1];
    out->vmax_soft_val = S->bc[2]; out->vmax_soft_r = S->bc[3];
    out->vmax_unsoft_val = S->bc[4]; out->vmax_unsoft_r = S->bc[5];
    __syncthreads();
}

// ---- bitonic sort of (r, pidx) ---------------------------------------------------------------
SOAP_DEV void cmpx_lds(Shared *S, int a, int b, bool asc) {
    double ka = S->keys[a], kb = S->keys[b];
    unsigned va = S->vals[a], vb = S->vals[b];
    bool sw = asc ? (ka > kb || (ka == kb && va > vb)) : (ka < kb || (ka == kb && va < vb));
    if (sw) { S->keys[a] = kb; S->keys[b] = ka; S->vals[a] = vb; S->vals[b] = va; }
}

// sort X.r / X.pidx [0, n) ascending; entries [n, npad) are padded with +inf
__device__ __noinline__ void sort_slab(Ctx &X, int n) {
    PhaseTimer pt_(X, PH_SORT);
    Shared *S = X.S;
    int npad = 1;
    while (npad < n) npad <<= 1;
    for (int i = n + X.tid; i < npad; i += BLK) { X.r[i] = INFINITY; X.pidx[i] = 0xffffffffu; }
    __syncthreads();
    int chunk = npad < CH ? npad : CH;
    // phase 1: sort each chunk completely in LDS (direction alternates so that the chunks form
    // bitonic sequences for the global merge)
    for (int cb = 0; cb < npad; cb += chunk) {
        for (int i = X.tid; i < chunk; i += BLK) { S->keys[i] = X.r[cb + i]; S->vals[i] = X.pidx[cb + i]; }
        __syncthreads();
        for (int k = 2; k <= chunk; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = X.tid; t < chunk / 2; t += BLK) {
                    int a = 2 * t - (t & (j - 1));
                    int b = a + j;
                    bool asc = (((cb + a) & k) == 0);
                    cmpx_lds(S, a, b, asc);
                }
                __syncthreads();
            }
        }
        for (int i = X.tid; i < chunk; i += BLK) { X.r[cb + i] = S->keys[i]; X.pidx[cb + i] = S->vals[i]; }
        __syncthreads();
    }
    // phase 2: merges across chunks
    for (int k = 2 * chunk; k <= npad; k <<= 1) {
        for (int j = k >> 1; j >= chunk; j >>= 1) {
            for (int t = X.tid; t < npad / 2; t += BLK) {
                int a = 2 * t - (t & (j - 1));
                int b = a + j;
                bool asc = ((a & k) == 0);
                double ka = X.r[a], kb = X.r[b];
                unsigned va = X.pidx[a], vb = X.pidx[b];
                bool sw = asc ? (ka > kb || (ka == kb && va > vb)) : (ka < kb || (ka == kb && va < vb));
                if (sw) { X.r[a] = kb; X.r[b] = ka; X.pidx[a] = vb; X.pidx[b] = va; }
            }
            __syncthreads();
        }
        for (int cb = 0; cb < npad; cb += chunk) {
            for (int i = X.tid; i < chunk; i += BLK) { S->keys[i] = X.r[cb + i]; S->vals[i] = X.pidx[cb + i]; }
            __syncthreads();
            bool asc = ((cb & k) == 0);
            for (int j = chunk >> 1; j > 0; j >>= 1) {
                for (int t = X.tid; t < chunk / 2; t += BLK) {
                    int a = 2 * t - (t & (j - 1));
                    cmpx_lds(S, a, a + j, asc);
                }
                __syncthreads();
            }
            for (int i = X.tid; i < chunk; i += BLK) { X.r[cb + i] = S->keys[i]; X.pidx[cb + i] = S->vals[i]; }
            __syncthreads();
        }
    }
}

// ---- candidate cells of one particle type (shared_mesh.py:146-179) ------------------------------
__device__ void build_cellmask(const Ctx &X, const TypeDev &T, double R) {
    Shared *S = X.S;
    if (X.tid < 3) {
        int d = X.tid;
        unsigned long long mk[4] = {0, 0, 0, 0};
        double c = X.c[d], L = X.L;
        int min_copy = 0, max_copy = 0, guard = 0;
        while (c + (double)(min_copy - 1) * L + R >= T.pos_min[d] && guard++ < 4096) min_copy--;
        guard = 0;
        while (c + (double)(max_copy + 1) * L - R <= T.pos_max[d] && guard++ < 4096) max_copy++;
        for (int cp = min_copy; cp <= max_copy; cp++) {
            double lo = fmax(T.pos_min[d], c + (double)cp * L - R);
            double hi = fmin(T.pos_max[d], c + (double)cp * L + R);
            double qlo = floor((lo - T.pos_min[d]) / T.cell_size[d]);
            double qhi = floor((hi - T.pos_min[d]) / T.cell_size[d]);
            int ilo = qlo < 0.0 ? 0 : (qlo > 1e9 ? 1000000000 : (int)qlo);
            int ihi = qhi < -1.0 ? -1 : (qhi > (double)(T.res - 1) ? T.res - 1 : (int)qhi);
            for (int q = ilo; q <= ihi; q++) mk[q >> 6] |= 1ull << (q & 63);
        }
        for (int k = 0; k < 4; k++) S->cellmask[d][k] = mk[k];
    }
    __syncthreads();
}

SOAP_DEV bool mask_bit(const unsigned long long *m, int q) { return (m[q >> 6] >> (q & 63)) & 1ull; }

// stream the candidate runs of every staged type, keep particles with r2 <= R^2 (shared_mesh.py:
// 183-194), write (exact wrapped radius, packed index) to the slab.  Returns via LDS: nsel,
// enclosed mass, n_cand.  overflow -> S->flag |= 1
__device__ __noinline__ void select_particles(Ctx &X, double R, double *mass_out, long long *ncand_out) {
    PhaseTimer pt_(X, PH_SELECT);
    Shared *S = X.S;
    const RunParams &P = *X.P;
    if (X.tid == 0) { S->nsel = 0; }
    __syncthreads();
    double msum = 0.0;
    long long ncand = 0;
    const double R2 = R * R;
    for (int ts = 0; ts < SOAP_NTYPES; ts++) {
        const TypeDev &T = P.t[ts];
        if (T.n == 0) continue;
        build_cellmask(X, T, R);
        const int res = T.res;
        for (int k = 0; k < res; k++) {
            if (!mask_bit(S->cellmask[2], k)) continue;
            for (int j = 0; j < res; j++) {
                if (!mask_bit(S->cellmask[1], j)) continue;
                long long rowbase = (long long)res * j + (long long)res * res * k;
                int i0 = 0;
                while (i0 < res) {
                    if (!mask_bit(S->cellmask[0], i0)) { i0++; continue; }
                    int i1 = i0;
                    while (i1 + 1 < res && mask_bit(S->cellmask[0], i1 + 1)) i1++;
                    unsigned beg = T.cell_offset[rowbase + i0];
                    unsigned end = T.cell_offset[rowbase + i1 + 1];
                    ncand += (X.tid == 0) ? (long long)(end - beg) : 0;
                    for (unsigned pb = beg; pb < end; pb += BLK) {
                        unsigned p = pb + X.tid;
                        bool sel = false;
                        double x = 0, y = 0, z = 0;
                        if (p < end) {
                            x = T.x[p]; y = T.y[p]; z = T.z[p];
                            double ax = min_image(x, X.c[0], X.L);
                            double ay = min_image(y, X.c[1], X.L);
                            double az = min_image(z, X.c[2], X.L);
                            double r2 = (ax * ax + ay * ay) + az * az;
                            sel = r2 <= R2;
                        }
                        unsigned long long b = __ballot(sel);
                        if (b) {
                            int cnt = __popcll(b);
                            int leader = __ffsll((long long)b) - 1;
                            int basei = 0;
                            if (X.lane == leader) basei = atomicAdd(&S->nsel, cnt);
                            basei = __shfl(basei, leader, 64);
                            if (sel) {
                                int slot = basei + __popcll(b & ((1ull << X.lane) - 1ull));
                                if (slot < X.cap) {
                                    double wx = wrap_rel(x, X.c[0], X.L);
                                    double wy = wrap_rel(y, X.c[1], X.L);
                                    double wz = wrap_rel(z, X.c[2], X.L);
                                    X.r[slot] = sqrt((wx * wx + wy * wy) + wz * wz);
                                    X.pidx[slot] = ((unsigned)ts << PIDX_BITS) | p;
                                }
                                msum += T.m[p];
                            }
                        }
                    }
                    i0 = i1 + 1;
                }
            }
        }
        __syncthreads();
    }
    double v = wave_sum(msum);
    if (X.lane == 0) S->red[X.w][0] = v;
    __syncthreads();
    if (X.tid == 0) {
        double tot = S->red[0][0];
        for (int ww = 1; ww < NWAVE; ww++) tot += S->red[ww][0];
        S->bc[0] = tot;
        S->bi[0] = ncand;
    }
    __syncthreads();
    *mass_out = S->bc[0];
    *ncand_out = S->bi[0];
    __syncthreads();
}

// gather the per-particle fields into the sorted slab, build the flags and run the sequential
// cumulative-mass pass (numpy.cumsum order: SO_properties.py:400)
__device__ __noinline__ void materialize(Ctx &X, int n) {
    PhaseTimer pt_(X, PH_MATERIALIZE);
    Shared *S = X.S;
    const RunParams &P = *X.P;
    if (X.tid == 0) {
        unsigned pi = X.pidx[0];
        int ts = pi >> PIDX_BITS;
        unsigned p = pi & PIDX_MASK;
        const TypeDev &T = P.t[ts];
        S->bi[1] = T.fof[p];               // FOF id of the innermost particle (SO_properties.py:408-409)
        S->bc[3] = T.vx[p]; S->bc[4] = T.vy[p]; S->bc[5] = T.vz[p];   // velocity shift
        S->bc[6] = 0.0;                    // running cumulative mass
    }
    __syncthreads();
    const long long fof0 = S->bi[1];
    X.shift[0] = S->bc[3]; X.shift[1] = S->bc[4]; X.shift[2] = S->bc[5];
    for (int cb = 0; cb < n; cb += CH) {
        int cn = n - cb < CH ? n - cb : CH;
        for (int j = X.tid; j < cn; j += BLK) {
            int i = cb + j;
            unsigned pi = X.pidx[i];
            int ts = pi >> PIDX_BITS;
            unsigned p = pi & PIDX_MASK;
            const TypeDev &T = P.t[ts];
            double m = T.m[p];
            X.m[i] = m;
            S->keys[j] = m;
            X.dx[i] = wrap_rel(T.x[p], X.c[0], X.L);
            X.dy[i] = wrap_rel(T.y[p], X.c[1], X.L);
            X.dz[i] = wrap_rel(T.z[p], X.c[2], X.L);
            X.vx[i] = T.vx[p] - X.shift[0];
            X.vy[i] = T.vy[p] - X.shift[1];
            X.vz[i] = T.vz[p] - X.shift[2];
            long long g = T.grnr64 ? T.grnr64p[p] : (long long)T.grnr32[p];
            long long f = T.fof[p];
            unsigned char fl = (unsigned char)ts;
            if (g == X.hindex) fl |= FL_BOUND;
            if (g >= 0 && g != X.hindex) fl |= (f == fof0) ? FL_SAT : FL_EXT;
            X.flags[i] = fl;
        }
        __syncthreads();
        if (X.tid == 0) {
            double run = S->bc[6];
            for (int j = 0; j < cn; j++) { run += S->keys[j]; S->keys[j] = run; }
            S->bc[6] = run;
        }
        __syncthreads();
        for (int j = X.tid; j < cn; j += BLK) X.cum[cb + j] = S->keys[j];
        __syncthreads();
    }
}

// block-wide minimum of an index-valued predicate result
__device__ long long block_min_ll(const Ctx &X, long long v) {
    v = wave_min_ll(v);
    if (X.lane == 0) X.S->bi[4 + X.w] = v;
    __syncthreads();
    long long r = X.S->bi[4];
    for (int ww = 1; ww < NWAVE; ww++) r = X.S->bi[4 + ww] < r ? X.S->bi[4 + ww] : r;
    __syncthreads();
    return r;
}

SOAP_DEV double dens_at(const Ctx &X, int i, double nuK, double K43, double *cum_nu) {
    double r3 = cube(X.r[i]);
    double cn = X.cum[i] + nuK * r3;
    *cum_nu = cn;
    return cn / (K43 * r3);
}

// first index >= lo with r[i] >= R (strict) or r[i] > R; the slab is sorted
SOAP_DEV int lower_bound_r(const Ctx &X, int n, double R, bool strict_less_inside) {
    // returns number of elements with r < R (strict_less_inside) or r <= R
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        double v = X.r[mid];
        bool inside = strict_less_inside ? (v < R) : (v <= R);
        if (inside) lo = mid + 1; else hi = mid;
    }
    return lo;
}

#define RES_OK 0
#define RES_TOO_SMALL 1

SOAP_DEV void write_com(double *dst, const double *mx, double M, const double *c, double L) {
    for (int d = 0; d < 3; d++) dst[d] = np_mod(mx[d] / M + c[d], L);
}

// ---- SO calculations --------------------------------------------------------------------------
__device__ __noinline__ int calc_SO(Ctx &X, const soap_calc_desc &cd, double *row, const double *rowbase_all,
                       const int *coloff, double Rcur, double *required) {
    Shared *S = X.S;
    const RunParams &P = *X.P;
    const int n = X.n;
    if (n == 0) return RES_OK;
    const double nuK = ((P.nu_density * 4.0) / 3.0) * M_PI;
    const double K43 = (4.0 / 3.0) * M_PI;
    double R_phys = 0.0;
    if (cd.kind == SOAP_CALC_SO_MULTIPLE) {
        // SO_properties.py:3912-3923: parent radius as stored (float32)
        double rp = rowbase_all[coloff[cd.parent] + SOC_r];
        float r32 = (float)rp;
        R_phys = cd.radius * (double)r32;
        // halo_prop.physical_radius_mpc = multiple * R.to("Mpc"); the scheduler converts it back
        // (SO_properties.py:3917, halo_tasks.py:168): same two roundings as the oracle
        *required = ((R_phys * P.a) / P.mpc) * P.mpc / P.a;
        float chk = (float)cd.radius * r32;
        if ((double)chk > Rcur) return RES_TOO_SMALL;
    } else if (cd.kind == SOAP_CALC_SO_PHYSICAL) {
        R_phys = cd.radius;
    }
    // nskip = max(1, argmax(r > 0))  (SO_properties.py:416)
    long long first_pos = 0x7fffffffffffffffLL;
    for (int i = X.tid; i < n; i += BLK)
        if (X.r[i] > 0.0) { first_pos = i; break; }
    first_pos = block_min_ll(X, first_pos);
    int nskip = (first_pos == 0x7fffffffffffffffLL) ? 0 : (int)first_pos;
    if (nskip < 1) nskip = 1;
    const int nr_parts = n - nskip;
    double SO_r = 0.0, SO_mass = 0.0;
    int have = 0;
    if (cd.kind == SOAP_CALC_SO_DENSITY) {
        const double rho = cd.reference_density;
        if (nr_parts <= 0) return RES_OK;   // reference leaves SO_r undefined; documented divergence
        double cn;
        bool above0 = dens_at(X, nskip, nuK, K43, &cn) > rho;
        int result = 0;   // 0 ok, 1 too small, <0 fatal
        if (above0) {
            long long ib = 0x7fffffffffffffffLL;
            for (int i = nskip + 1 + X.tid; i < n; i += BLK) {
                double c2;
                if (!(dens_at(X, i, nuK, K43, &c2) > rho)) { ib = i; break; }
            }
            ib = block_min_ll(X, ib);
            if (X.tid == 0) {
                int res = 0;
                double so_r = 0, so_m = 0;
                if (ib == 0x7fffffffffffffffLL) {
                    res = (X.r[n - 1] > P.mpc20) ? SOAP_HALO_FATAL_SO_BEYOND_20MPC : 1;
                } else {
                    int i = (int)ib;
                    double r1 = X.r[i - 1], r2 = X.r[i], M1, M2, t;
                    bool a1 = dens_at(X, i - 1, nuK, K43, &M1) > rho;
                    bool a2 = dens_at(X, i, nuK, K43, &M2) > rho;
                    while (r1 == r2 || a1 == a2) {
                        i++;
                        if (i >= n) { res = (X.r[n - 1] > P.mpc20) ? SOAP_HALO_FATAL_SO_BEYOND_20MPC : 1; break; }
                        r1 = r2; r2 = X.r[i];
                        M1 = M2; a1 = a2;
                        a2 = dens_at(X, i, nuK, K43, &t) > rho;
                        M2 = t;
                    }
                    if (res == 0) {
                        double rho_dim = rho * cube(r1) / M1;
                        double slope_dim = (M2 - M1) / (r2 - r1) * (r1 / M1);
                        SOIntersection f;
                        f.c3 = 4.0 * M_PI / 3.0 * rho_dim;
                        f.s = slope_dim;
                        int err;
                        double u = brentq(f, 1.0, r2 / r1, &err);
                        so_r = r1 * u;
                        double vol = 4.0 / 3.0 * M_PI * cube(so_r);
                        so_m = vol * rho;
                    }
                }
                S->bi[0] = res; S->bc[0] = so_r; S->bc[1] = so_m;
            }
        } else {
            long long ip = 0x7fffffffffffffffLL;
            for (int i = nskip + X.tid; i < n; i += BLK) {
                double c2;
                dens_at(X, i, nuK, K43, &c2);
                if (!(c2 < 0.0)) { ip = i; break; }
            }
            ip = block_min_ll(X, ip);
            if (X.tid == 0) {
                int res = 0;
                double so_r = 0, so_m = 0;
                if (ip == 0x7fffffffffffffffLL) res = SOAP_HALO_FATAL_SO_BEYOND_20MPC;  // "Should never happen"
                else {
                    double Mi, ri = X.r[(int)ip];
                    dens_at(X, (int)ip, nuK, K43, &Mi);
                    so_r = sqrt(0.75 * Mi / (M_PI * ri * rho));
                    so_m = Mi * so_r / ri;
                }
                S->bi[0] = res; S->bc[0] = so_r; S->bc[1] = so_m;
            }
        }
        __syncthreads();
        result = (int)S->bi[0];
        SO_r = S->bc[0]; SO_mass = S->bc[1];
        __syncthreads();
        if (result == 1) return RES_TOO_SMALL;
        if (result < 0) return result;
        have = 1;
    } else if (R_phys > 0.0) {
        SO_r = R_phys;
        if (nr_parts > 0) {
            if (X.tid == 0) {
                // SO_properties.py:436-452 on the profile with the first nskip entries dropped
                int nin = lower_bound_r(X, n, SO_r, false);   // elements with r <= SO_r
                int i = nin < nskip ? nskip : nin;            // first retained element with r > SO_r
                double t, so_m;
                if (i >= n) { dens_at(X, n - 1, nuK, K43, &t); so_m = t; }
                else if (i == nskip) { dens_at(X, i, nuK, K43, &t); so_m = t; }
                else {
                    double M1, M2, r1 = X.r[i - 1], r2 = X.r[i];
                    dens_at(X, i - 1, nuK, K43, &M1);
                    dens_at(X, i, nuK, K43, &M2);
                    so_m = M1 + (SO_r - r1) / (r2 - r1) * (M2 - M1);
                }
                S->bc[1] = so_m;
            }
            __syncthreads();
            SO_mass = S->bc[1];
            __syncthreads();
            have = 1;
        }
    }
    if (!have) return RES_OK;
    if (!(SO_r > 0.0 && SO_mass > 0.0)) return RES_OK;

    // particles with r < SO_r (strict, SO_properties.py:485)
    int nso = lower_bound_r(X, n, SO_r, true);
    Pred pr; pr.nmax = nso; pr.bound_only = 0;
    double *a = X.S->mom;
    reduce_moments(X, pr, a);
    ScanOut so;
    scan_pass(X, pr, 0.0, 0.0, false, &so);
    if (X.tid == 0) {
        row[SOC_r] = SO_r;
        row[SOC_Mtot] = SO_mass;
        row[SOC_Ngas] = a[A_N0]; row[SOC_Ndm] = a[A_N1]; row[SOC_Nstar] = a[A_N2]; row[SOC_Nbh] = a[A_N3];
        double Mp = ((a[A_M0] + a[A_M1]) + a[A_M2]) + a[A_M3];
        if (Mp != 0.0) {
            write_com(row + SOC_com, a + A_MX, Mp, X.c, X.L);
            double dv[3];
            for (int d = 0; d < 3; d++) { dv[d] = a[A_MV + d] / Mp; row[SOC_vcom + d] = dv[d] + X.shift[d]; }
            double vmax = so.vmax_soft_val > 0.0 ? sqrt(so.vmax_soft_val * P.G_eff) : 0.0;
            row[SOC_R_vmax_soft] = so.vmax_soft_val > 0.0 ? so.vmax_soft_r : 0.0;
            row[SOC_Vmax_soft] = vmax;
            if (vmax > 0.0) {
                // L about the sphere's own centre-of-mass velocity (SO_properties.py:612-617)
                double lx = a[A_LA] - (a[A_MX + 1] * dv[2] - a[A_MX + 2] * dv[1]);
                double ly = a[A_LA + 1] - (a[A_MX + 2] * dv[0] - a[A_MX] * dv[2]);
                double lz = a[A_LA + 2] - (a[A_MX] * dv[1] - a[A_MX + 1] * dv[0]);
                double Lt = sqrt((lx * lx + ly * ly) + lz * lz);
                row[SOC_spin] = Lt / (sqrt(2.0) * Mp * SO_r * vmax);
            }
        }
        row[SOC_Mfrac_sat] = a[A_MSAT] / SO_mass;
        row[SOC_Mfrac_ext] = a[A_MEXT] / SO_mass;
        row[SOC_Mgas] = a[A_M0]; row[SOC_Mdm] = a[A_M1]; row[SOC_Mstar] = a[A_M2]; row[SOC_Mbh] = a[A_M3];
        if (a[A_M1] != 0.0) {
            double dv[3];
            for (int d = 0; d < 3; d++) dv[d] = a[A_MVD + d] / a[A_M1];
            row[SOC_Ldm] = a[A_LD] - (a[A_MXD + 1] * dv[2] - a[A_MXD + 2] * dv[1]);
            row[SOC_Ldm + 1] = a[A_LD + 1] - (a[A_MXD + 2] * dv[0] - a[A_MXD] * dv[2]);
            row[SOC_Ldm + 2] = a[A_LD + 2] - (a[A_MXD] * dv[1] - a[A_MXD + 1] * dv[0]);
        }
        if (cd.virial_definition && nso >= 10) {
            // SO_properties.py:2725-2765
            for (int s = 0; s < 2; s++) {
                double R1 = s == 0 ? a[A_MR] : a[A_MRS];
                double missed = SO_mass - Mp;
                double sr2 = SO_r * SO_r;
                R1 += M_PI * P.nu_density * (sr2 * sr2);
                missed -= P.nu_density * 4.0 / 3.0 * M_PI * cube(SO_r);
                R1 += missed * SO_r;
                R1 /= SO_r * SO_mass;
                const double poly[6] = {-79.71, -222.46, -250.14, -140.17, -43.59, -5.07};
                double lg = log10(R1), c = 0.0, pw = 1.0;
                for (int i = 0; i < 6; i++) { c += poly[5 - i] * pw; pw *= lg; }
                c = fmax(fmin(c, 3.0), 0.0);
                row[s == 0 ? SOC_conc_unsoft : SOC_conc_soft] = pow(10.0, c);
            }
        }
    }
    __syncthreads();
    return RES_OK;
}

// ---- spherical apertures ---------------------------------------------------------------------
__device__ __noinline__ int calc_aperture(Ctx &X, const soap_calc_desc &cd, double *row, double Rcur) {
    const RunParams &P = *X.P;
    if (Rcur < cd.radius) return RES_TOO_SMALL;   // aperture_properties.py:4140-4143
    int nin = lower_bound_r(X, X.n, cd.radius, false);   // r <= R_ap (aperture_properties.py:310)
    Pred pr; pr.nmax = nin; pr.bound_only = (cd.kind == SOAP_CALC_APERTURE_EXCL);
    double *a = X.S->mom;
    reduce_moments(X, pr, a);
    ScanOut so;
    scan_pass(X, pr, 0.0, 0.5 * a[A_M1], false, &so);
    if (X.tid == 0) {
        double Mtot = ((a[A_M0] + a[A_M1]) + a[A_M2]) + a[A_M3];
        row[AP_Mtot] = Mtot;
        row[AP_Mgas] = a[A_M0]; row[AP_Mdm] = a[A_M1]; row[AP_Mstar] = a[A_M2]; row[AP_Mbh] = a[A_M3];
        row[AP_Ngas] = a[A_N0]; row[AP_Ndm] = a[A_N1]; row[AP_Nstar] = a[A_N2]; row[AP_Nbh] = a[A_N3];
        if (Mtot != 0.0) {
            write_com(row + AP_com, a + A_MX, Mtot, X.c, X.L);
            for (int d = 0; d < 3; d++) row[AP_vcom + d] = a[A_MV + d] / Mtot + X.shift[d];
            row[AP_R_vmax_soft] = so.vmax_soft_val > 0.0 ? so.vmax_soft_r : 0.0;
            row[AP_Vmax_soft] = so.vmax_soft_val > 0.0 ? sqrt(so.vmax_soft_val * P.G_eff) : 0.0;
        }
        if (a[A_M1] != 0.0) {
            double Md = a[A_M1], dv[3];
            for (int d = 0; d < 3; d++) dv[d] = a[A_MVD + d] / Md;
            row[AP_Ldm] = a[A_LD] - (a[A_MXD + 1] * dv[2] - a[A_MXD + 2] * dv[1]);
            row[AP_Ldm + 1] = a[A_LD + 1] - (a[A_MXD + 2] * dv[0] - a[A_MXD] * dv[2]);
            row[AP_Ldm + 2] = a[A_LD + 2] - (a[A_MXD] * dv[1] - a[A_MXD + 1] * dv[0]);
            // V_ab = <v'_a v'_b> - d_a d_b   (kinematic_properties.py:117-127, shifted form)
            row[AP_veldisp_dm + 0] = a[A_VV + 0] / Md - dv[0] * dv[0];
            row[AP_veldisp_dm + 1] = a[A_VV + 1] / Md - dv[1] * dv[1];
            row[AP_veldisp_dm + 2] = a[A_VV + 2] / Md - dv[2] * dv[2];
            row[AP_veldisp_dm + 3] = a[A_VV + 3] / Md - dv[0] * dv[1];
            row[AP_veldisp_dm + 4] = a[A_VV + 4] / Md - dv[0] * dv[2];
            row[AP_veldisp_dm + 5] = a[A_VV + 5] / Md - dv[1] * dv[2];
            row[AP_HMR_dm] = so.hmr_dm;
        }
    }
    __syncthreads();
    return RES_OK;
}

// ---- bound subhalo ------------------------------------------------------------------------------
__device__ __noinline__ int calc_subhalo(Ctx &X, double *row, long long nr_bound, double *counts /*LDS, 4*/) {
    Shared *S = X.S;
    const RunParams &P = *X.P;
    Pred pr; pr.nmax = X.n; pr.bound_only = 1;
    double *a = X.S->mom;
    reduce_moments(X, pr, a);
    double Ntot = ((a[A_N0] + a[A_N1]) + a[A_N2]) + a[A_N3];
    if (Ntot < (double)nr_bound) return RES_TOO_SMALL;               // subhalo_properties.py:2634
    if (Ntot > (double)nr_bound) return SOAP_HALO_FATAL_TOO_MANY_BOUND;  // :2642
    double Mtot = ((a[A_M0] + a[A_M1]) + a[A_M2]) + a[A_M3];
    ScanOut so;
    bool allclose = (Ntot > 0.0 && a[A_NFAR] == 0.0);
    scan_pass(X, pr, 0.5 * Mtot, 0.5 * a[A_M1], allclose, &so);
    double vmax_s = 0.0, rmax_s = 0.0;
    if (Mtot != 0.0 && so.vmax_soft_val > 0.0) { vmax_s = sqrt(so.vmax_soft_val * P.G_eff); rmax_s = so.vmax_soft_r; }
    // spin: second reduction over bound particles with r <= R_vmax_soft (subhalo_properties.py:1059-1072)
    double *b = X.S->mom2;
    bool do_spin = (Mtot != 0.0 && rmax_s > 0.0 && vmax_s > 0.0);
    if (do_spin) {
        Pred p2; p2.nmax = lower_bound_r(X, X.n, rmax_s, false); p2.bound_only = 1;
        reduce_moments(X, p2, b);
    }
    if (X.tid == 0) {
        counts[0] = a[A_N0]; counts[1] = a[A_N1]; counts[2] = a[A_N2]; counts[3] = a[A_N3];
        row[SH_Mtot] = Mtot;
        row[SH_Mgas] = a[A_M0]; row[SH_Mdm] = a[A_M1]; row[SH_Mstar] = a[A_M2]; row[SH_Mbh] = a[A_M3];
        row[SH_Ngas] = a[A_N0]; row[SH_Ndm] = a[A_N1]; row[SH_Nstar] = a[A_N2]; row[SH_Nbh] = a[A_N3];
        double dv[3] = {0, 0, 0};
        if (Mtot != 0.0) {
            write_com(row + SH_com, a + A_MX, Mtot, X.c, X.L);
            for (int d = 0; d < 3; d++) { dv[d] = a[A_MV + d] / Mtot; row[SH_vcom + d] = dv[d] + X.shift[d]; }
            if (so.vmax_unsoft_val > 0.0) {
                row[SH_R_vmax_unsoft] = so.vmax_unsoft_r;
                row[SH_Vmax_unsoft] = sqrt(so.vmax_unsoft_val * P.G_eff);
            }
            row[SH_Vmax_soft] = vmax_s;
            row[SH_EncloseRadius] = a[A_RMAX];
            if (do_spin) {
                double Mr = ((b[A_M0] + b[A_M1]) + b[A_M2]) + b[A_M3];
                if (Mr > 0.0) {
                    double lx = b[A_LA] - (b[A_MX + 1] * dv[2] - b[A_MX + 2] * dv[1]);
                    double ly = b[A_LA + 1] - (b[A_MX + 2] * dv[0] - b[A_MX] * dv[2]);
                    double lz = b[A_LA + 2] - (b[A_MX] * dv[1] - b[A_MX + 1] * dv[0]);
                    double Lt = sqrt((lx * lx + ly * ly) + lz * lz);
                    row[SH_spin] = Lt / (sqrt(2.0) * Mr * vmax_s * rmax_s);
                }
            }
        }
        if (a[A_M1] != 0.0) {
            double Md = a[A_M1], dd[3];
            for (int d = 0; d < 3; d++) dd[d] = a[A_MVD + d] / Md;
            row[SH_Ldm] = a[A_LD] - (a[A_MXD + 1] * dd[2] - a[A_MXD + 2] * dd[1]);
            row[SH_Ldm + 1] = a[A_LD + 1] - (a[A_MXD + 2] * dd[0] - a[A_MXD] * dd[2]);
            row[SH_Ldm + 2] = a[A_LD + 2] - (a[A_MXD] * dd[1] - a[A_MXD + 1] * dd[0]);
            row[SH_veldisp_dm + 0] = a[A_VV + 0] / Md - dd[0] * dd[0];
            row[SH_veldisp_dm + 1] = a[A_VV + 1] / Md - dd[1] * dd[1];
            row[SH_veldisp_dm + 2] = a[A_VV + 2] / Md - dd[2] * dd[2];
            row[SH_veldisp_dm + 3] = a[A_VV + 3] / Md - dd[0] * dd[1];
            row[SH_veldisp_dm + 4] = a[A_VV + 4] / Md - dd[0] * dd[2];
            row[SH_veldisp_dm + 5] = a[A_VV + 5] / Md - dd[1] * dd[2];
        }
        row[SH_HMR_tot] = so.hmr_all;
        row[SH_HMR_dm] = so.hmr_dm;
    }
    __syncthreads();
    return RES_OK;
}

__global__ void __launch_bounds__(BLK) k_halo(const RunParams *Pp) {
    const RunParams &P = *Pp;
    __shared__ Shared S;
    __shared__ double counts[4];
    Ctx X;
    X.P = Pp; X.S = &S;
    X.tid = threadIdx.x; X.lane = threadIdx.x & 63; X.w = threadIdx.x >> 6;
    X.L = P.boxsize;
    X.cap = P.slab.cap;
    X.phase = P.phase;
    const long long so = (long long)blockIdx.x * P.slab.cap;
    X.r = P.slab.r + so; X.m = P.slab.m + so; X.dx = P.slab.dx + so; X.dy = P.slab.dy + so;
    X.dz = P.slab.dz + so; X.vx = P.slab.vx + so; X.vy = P.slab.vy + so; X.vz = P.slab.vz + so;
    X.cum = P.slab.cum + so; X.pidx = P.slab.pidx + so; X.flags = P.slab.flags + so;
    const long long ntask = P.halo.todo ? P.halo.n_todo : P.halo.n;

    while (true) {
        if (X.tid == 0) S.bi[0] = (long long)atomicAdd(P.queue, 1u);
        __syncthreads();
        long long task = S.bi[0];
        __syncthreads();
        if (task >= ntask) break;
        const long long hrow = P.halo.todo ? P.halo.todo[task] : task;
        PhaseTimer pt_total(X, PH_TOTAL);
        X.c[0] = P.halo.cofp[3 * hrow]; X.c[1] = P.halo.cofp[3 * hrow + 1]; X.c[2] = P.halo.cofp[3 * hrow + 2];
        X.hindex = P.halo.index[hrow];
        const double Rread = P.halo.read_radius[hrow];
        double Rsearch = P.halo.search_radius[hrow];
        const int central = P.halo.is_central[hrow];
        const long long nr_bound = P.halo.nr_bound[hrow];
        const bool use_target = central == 1 && P.target_density > 0.0;
        double *row = P.table + hrow * P.ncols;
        if (X.tid < 4) counts[X.tid] = 0.0;
        unsigned long long done = 0ull;
        const unsigned long long alldone = (P.ncalc >= 64) ? ~0ull : ((1ull << P.ncalc) - 1ull);
        double Rcur = Rsearch;
        int status = SOAP_HALO_OK;
        long long n_loop = 0, ncand = 0;
        int nsel_final = 0;
        while (true) {
            n_loop++;
            if (!(Rcur <= Rread)) { status = SOAP_HALO_FATAL_RADIUS_ASSERT; break; }
            double mass_total;
            if (X.tid == 0) S.flag = 0;
            select_particles(X, Rcur, &mass_total, &ncand);
            int n = S.nsel;
            __syncthreads();
            if (n > X.cap) { status = SOAP_HALO_FATAL_SLAB_OVERFLOW; break; }
            X.n = n;
            nsel_final = n;
            double required = 0.0;
            // density = mass_total / (4/3 pi R^3) <= target  (halo_tasks.py:97-103)
            double density = mass_total / (4.0 / 3.0 * M_PI * cube(Rcur));
            if (!use_target || density <= P.target_density) {
                if (n > 0) {
                    sort_slab(X, n);
                    materialize(X, n);
                }
                int fatal = 0;
                for (int c = 0; c < P.ncalc; c++) {
                    if (done & (1ull << c)) continue;
                    const soap_calc_desc cd = P.calcs[c];
                    double *crow = row + P.coloff[c];
                    int res = RES_OK;
                    // category filter of the calculation (category_filter.py:69-106)
                    bool filt = true;
                    if (cd.filter_limit > 0 && cd.kind != SOAP_CALC_SUBHALO) {
                        double v = 0.0;
                        for (int t = 0; t < 4; t++) if (cd.filter_types & (1 << t)) v += counts[t];
                        filt = v >= (double)cd.filter_limit;
                    }
                    double creq = cd.physical_radius_mpc;
                    if (cd.kind == SOAP_CALC_SUBHALO) {
                        res = calc_subhalo(X, crow, nr_bound, counts);
                    } else if (cd.kind <= SOAP_CALC_SO_MULTIPLE) {
                        if (central && filt) res = calc_SO(X, cd, crow, row, P.coloff, Rcur, &creq);
                    } else if (cd.kind == SOAP_CALC_APERTURE_INCL || cd.kind == SOAP_CALC_APERTURE_EXCL) {
                        bool skip = false;
                        if (cd.parent >= 0) {
                            // aperture_properties.py:4086-4123, EncloseRadius as stored (float32)
                            float enc = (float)row[P.coloff[0] + SH_EncloseRadius];
                            if (cd.prev_radius > (double)enc) {
                                skip = true;
                                if (cd.kind == SOAP_CALC_APERTURE_EXCL && filt) {
                                    const double *prow = row + P.coloff[cd.parent];
                                    for (int k = X.tid; k < AP_NCOLS; k += BLK) crow[k] = prow[k];
                                    __syncthreads();
                                }
                            }
                        }
                        if (filt && !skip) res = calc_aperture(X, cd, crow, Rcur);
                    } else {
                        res = RES_OK;   // projected apertures: not built yet (columns stay zero)
                    }
                    if (res == RES_TOO_SMALL) { required = fmax(required, creq); break; }
                    if (res < 0) { fatal = res; break; }
                    done |= (1ull << c);
                }
                if (fatal) { status = fatal; break; }
                if (done == alldone) break;
            }
            // halo_tasks.py:167-187
            if (required > Rread) { Rsearch = fmax(Rsearch, required); status = SOAP_HALO_NEED_LARGER_READ_RADIUS; break; }
            else if (Rcur >= Rread) { Rsearch = fmax(Rsearch, Rcur); status = SOAP_HALO_NEED_LARGER_READ_RADIUS; break; }
            else {
                Rcur = fmin(Rcur * 1.2, Rread);
                Rcur = fmax(Rcur, required);
            }
        }
        __syncthreads();
        if (status != SOAP_HALO_OK) {
            // a halo that is not finished contributes no row (halo_tasks.py:383-402)
            for (long long k = X.tid; k < P.ncols; k += BLK) row[k] = 0.0;
        }
        if (X.tid == 0) {
            P.halo.status[hrow] = status;
            if (status == SOAP_HALO_NEED_LARGER_READ_RADIUS) {
                P.halo.read_radius[hrow] = fmax(Rread * 1.5, Rsearch);   // halo_tasks.py:393-402
                P.halo.search_radius[hrow] = Rsearch;
            }
            P.halo.stats[4 * hrow] = n_loop;
            P.halo.stats[4 * hrow + 1] = nsel_final;
            P.halo.stats[4 * hrow + 2] = ncand;
            P.halo.stats[4 * hrow + 3] = 0;
        }
        __syncthreads();
    }
}

// brute-force membership list for verification (same arithmetic as the property kernels)
__global__ void k_membership(TypeDev T, double cx, double cy, double cz, double L, double R,
                             int strict, int bound_only, long long hindex, unsigned *count,
                             long long *out, long long cap) {
    for (long long p = blockIdx.x * (long long)BLK + threadIdx.x; p < T.n;
         p += (long long)gridDim.x * BLK) {
        double wx = wrap_rel(T.x[p], cx, L), wy = wrap_rel(T.y[p], cy, L), wz = wrap_rel(T.z[p], cz, L);
        double r = sqrt((wx * wx + wy * wy) + wz * wz);
        bool in = strict ? (r < R) : (r <= R);
        if (in && bound_only) {
            long long g = T.grnr64 ? T.grnr64p[p] : (long long)T.grnr32[p];
            in = (g == hindex);
        }
        if (in) {
            unsigned s = atomicAdd(count, 1u);
            if ((long long)s < cap) out[s] = (long long)T.orig[p];
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct TypeHost {
    long long n = 0;
    int res = 0;
    int grnr64 = 0;
    double *x = nullptr, *y = nullptr, *z = nullptr, *vx = nullptr, *vy = nullptr, *vz = nullptr,
           *m = nullptr;
    int *g32 = nullptr;
    long long *g64 = nullptr, *fof = nullptr;
    unsigned *orig = nullptr, *cell_offset = nullptr;
    double pos_min[3], pos_max[3], cell_size[3];
    void release() {
        void *ptrs[] = {x, y, z, vx, vy, vz, m, g32, g64, fof, orig, cell_offset};
        for (void *p : ptrs) if (p) (void)hipFree(p);
        *this = TypeHost();
    }
};

struct soap_engine {
    soap_engine_config cfg;
    hipStream_t stream = nullptr;
    TypeHost t[SOAP_NTYPES];
    // halos
    long long n_halo = 0;
    double *d_cofp = nullptr, *d_search = nullptr, *d_read = nullptr;
    long long *d_index = nullptr, *d_nrbound = nullptr, *d_stats = nullptr;
    int *d_central = nullptr, *d_status = nullptr, *d_todo = nullptr;
    // run
    long long ncols = 0;
    double *d_table = nullptr;
    soap_calc_desc *d_calcs = nullptr;
    int *d_coloff = nullptr;
    unsigned *d_queue = nullptr;
    RunParams *d_params = nullptr;
    unsigned long long *d_phase = nullptr;
    unsigned long long phase[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // slabs
    long long slab_cap = 0;
    int slab_wgs = 0;
    SlabDev slab{};
    // timing
    double ms_halo = -1, ms_mesh = -1, ms_reorder = -1;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

static void free_slabs(soap_engine *h) {
    void *ptrs[] = {h->slab.r, h->slab.m, h->slab.dx, h->slab.dy, h->slab.dz, h->slab.vx,
                    h->slab.vy, h->slab.vz, h->slab.cum, h->slab.pidx, h->slab.flags};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    h->slab = SlabDev{};
    h->slab_cap = 0;
    h->slab_wgs = 0;
}

static int alloc_slabs(soap_engine *h, long long cap, int wgs) {
    free_slabs(h);
    size_t ne = (size_t)cap * (size_t)wgs;
    double **dp[] = {&h->slab.r, &h->slab.m, &h->slab.dx, &h->slab.dy, &h->slab.dz,
                     &h->slab.vx, &h->slab.vy, &h->slab.vz, &h->slab.cum};
    for (double **p : dp) HIPCHECK(hipMalloc((void **)p, ne * sizeof(double)));
    HIPCHECK(hipMalloc((void **)&h->slab.pidx, ne * sizeof(unsigned)));
    HIPCHECK(hipMalloc((void **)&h->slab.flags, ne));
    h->slab.cap = cap;
    h->slab_cap = cap;
    h->slab_wgs = wgs;
    return 0;
}

extern "C" {

int32_t soap_engine_abi_version(void) { return SOAP_ENGINE_ABI_VERSION; }
const char *soap_engine_last_error(void) { return g_err.c_str(); }

int32_t soap_engine_calc_columns(int32_t kind) {
    switch (kind) {
        case SOAP_CALC_SUBHALO: return SH_NCOLS;
        case SOAP_CALC_SO_DENSITY:
        case SOAP_CALC_SO_PHYSICAL:
        case SOAP_CALC_SO_MULTIPLE: return SOC_NCOLS;
        case SOAP_CALC_APERTURE_INCL:
        case SOAP_CALC_APERTURE_EXCL: return AP_NCOLS;
        case SOAP_CALC_PROJECTED: return 3 * PJ_NCOLS;
        default: return -1;
    }
}

int32_t soap_engine_create(const soap_engine_config *cfg, soap_engine **out) {
    if (!cfg || !out) return fail(-1, "null argument");
    if (cfg->abi_version != SOAP_ENGINE_ABI_VERSION) return fail(-2, "ABI version mismatch");
    if (!(cfg->boxsize > 0.0)) return fail(-3, "boxsize must be positive");
    if (!(cfg->a > 0.0) || !(cfg->mpc > 0.0)) return fail(-3, "a and mpc must be positive");
    int ndev = 0;
    HIPCHECK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(-4, "no HIP device: the engine has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(-5, "bad device ordinal");
    HIPCHECK(hipSetDevice(cfg->device));
    soap_engine *h = new soap_engine();
    h->cfg = *cfg;
    HIPCHECK(hipStreamCreate(&h->stream));
    HIPCHECK(hipEventCreate(&h->ev0));
    HIPCHECK(hipEventCreate(&h->ev1));
    HIPCHECK(hipMalloc((void **)&h->d_queue, sizeof(unsigned)));
    HIPCHECK(hipMalloc((void **)&h->d_params, sizeof(RunParams)));
    HIPCHECK(hipMalloc((void **)&h->d_phase, 8 * sizeof(unsigned long long)));
    *out = h;
    return 0;
}

static void free_halos(soap_engine *h) {
    void *ptrs[] = {h->d_cofp, h->d_search, h->d_read, h->d_index, h->d_nrbound, h->d_stats,
                    h->d_central, h->d_status, h->d_todo};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    h->d_cofp = h->d_search = h->d_read = nullptr;
    h->d_index = h->d_nrbound = h->d_stats = nullptr;
    h->d_central = h->d_status = h->d_todo = nullptr;
    h->n_halo = 0;
}

int32_t soap_engine_destroy(soap_engine *h) {
    if (!h) return 0;
    (void)hipSetDevice(h->cfg.device);
    (void)hipStreamSynchronize(h->stream);
    for (int t = 0; t < SOAP_NTYPES; t++) h->t[t].release();
    free_halos(h);
    free_slabs(h);
    if (h->d_table) (void)hipFree(h->d_table);
    if (h->d_calcs) (void)hipFree(h->d_calcs);
    if (h->d_coloff) (void)hipFree(h->d_coloff);
    if (h->d_queue) (void)hipFree(h->d_queue);
    if (h->d_params) (void)hipFree(h->d_params);
    if (h->d_phase) (void)hipFree(h->d_phase);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

int32_t soap_engine_stage_particles(soap_engine *h, int32_t ts, int64_t n, const double *pos,
                                    const double *vel, const double *mass, const void *grnr,
                                    int32_t grnr_bytes, const int64_t *fof, int32_t resolution,
                                    int32_t on_device) {
    if (!h) return fail(-1, "null handle");
    if (ts < 0 || ts >= SOAP_NTYPES) return fail(-2, "type_slot out of range");
    if (n < 0 || n >= (1ll << PIDX_BITS)) return fail(-3, "particle count per type must be < 2^30");
    if (grnr_bytes != 4 && grnr_bytes != 8) return fail(-4, "grnr_bytes must be 4 or 8");
    if (resolution < 1 || resolution > 256) return fail(-5, "mesh resolution must be in [1,256]");
    HIPCHECK(hipSetDevice(h->cfg.device));
    TypeHost &T = h->t[ts];
    T.release();
    if (n == 0) return 0;   // SharedMesh.empty (shared_mesh.py:25-27)
    if (!pos || !vel || !mass || !grnr || !fof) return fail(-6, "null particle array");
    hipStream_t st = h->stream;
    const double *dpos = pos, *dvel = vel, *dmass = mass;
    const void *dgr = grnr;
    const long long *dfof = (const long long *)fof;
    std::vector<void *> tmp;
    if (!on_device) {
        void *a, *b, *c, *d, *e;
        HIPCHECK(hipMalloc(&a, (size_t)n * 24)); tmp.push_back(a);
        HIPCHECK(hipMalloc(&b, (size_t)n * 24)); tmp.push_back(b);
        HIPCHECK(hipMalloc(&c, (size_t)n * 8)); tmp.push_back(c);
        HIPCHECK(hipMalloc(&d, (size_t)n * grnr_bytes)); tmp.push_back(d);
        HIPCHECK(hipMalloc(&e, (size_t)n * 8)); tmp.push_back(e);
        HIPCHECK(hipMemcpyAsync(a, pos, (size_t)n * 24, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(b, vel, (size_t)n * 24, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(c, mass, (size_t)n * 8, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(d, grnr, (size_t)n * grnr_bytes, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(e, fof, (size_t)n * 8, hipMemcpyHostToDevice, st));
        dpos = (const double *)a; dvel = (const double *)b; dmass = (const double *)c;
        dgr = d; dfof = (const long long *)e;
    }
    auto cleanup = [&]() { for (void *p : tmp) (void)hipFree(p); };

    HIPCHECK(hipEventRecord(h->ev0, st));
    // bounding box (shared_mesh.py:33-61)
    const int gmm = 1024;
    double *d_partial;
    HIPCHECK(hipMalloc((void **)&d_partial, gmm * 6 * sizeof(double)));
    hipLaunchKernelGGL(k_minmax, dim3(gmm), dim3(BLK), 0, st, dpos, (long long)n, d_partial);
    std::vector<double> part(gmm * 6);
    HIPCHECK(hipMemcpyAsync(part.data(), d_partial, gmm * 6 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    (void)hipFree(d_partial);
    for (int d = 0; d < 3; d++) { T.pos_min[d] = INFINITY; T.pos_max[d] = -INFINITY; }
    for (int b = 0; b < gmm; b++)
        for (int d = 0; d < 3; d++) {
            T.pos_min[d] = std::min(T.pos_min[d], part[b * 6 + d]);
            T.pos_max[d] = std::max(T.pos_max[d], part[b * 6 + 3 + d]);
        }
    for (int d = 0; d < 3; d++) {
        if (!(T.pos_min[d] <= T.pos_max[d])) { cleanup(); return fail(-7, "non-finite coordinates"); }
        if (T.pos_min[d] == T.pos_max[d]) T.pos_max[d] = T.pos_min[d] + 1.0;
        T.cell_size[d] = (T.pos_max[d] - T.pos_min[d]) / (double)resolution;
    }
    T.res = resolution;
    T.n = n;
    T.grnr64 = grnr_bytes == 8;
    const long long ncell = (long long)resolution * resolution * resolution;

    unsigned *d_cell, *d_count, *d_cursor, *d_sortidx, *d_tilesum;
    HIPCHECK(hipMalloc((void **)&d_cell, (size_t)n * 4));
    HIPCHECK(hipMalloc((void **)&d_count, (size_t)ncell * 4));
    HIPCHECK(hipMalloc((void **)&d_cursor, (size_t)ncell * 4));
    HIPCHECK(hipMalloc((void **)&d_sortidx, (size_t)n * 4));
    HIPCHECK(hipMalloc((void **)&T.cell_offset, (size_t)(ncell + 1) * 4));
    HIPCHECK(hipMemsetAsync(d_count, 0, (size_t)ncell * 4, st));
    HIPCHECK(hipMemsetAsync(d_cursor, 0, (size_t)ncell * 4, st));
    MeshGeom g;
    for (int d = 0; d < 3; d++) { g.pos_min[d] = T.pos_min[d]; g.cell_size[d] = T.cell_size[d]; }
    g.res = resolution;
    int grid = (int)std::min<long long>((n + BLK - 1) / BLK, 256 * 8);
    hipLaunchKernelGGL(k_cellid, dim3(grid), dim3(BLK), 0, st, dpos, (long long)n, g, d_cell, d_count);
    long long ntiles = (ncell + SCAN_TILE - 1) / SCAN_TILE;
    HIPCHECK(hipMalloc((void **)&d_tilesum, (size_t)ntiles * 4));
    hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)ntiles), dim3(BLK), 0, st, d_count, ncell, T.cell_offset, d_tilesum);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(BLK), 0, st, d_tilesum, ntiles);
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)ntiles), dim3(BLK), 0, st, T.cell_offset, ncell, d_tilesum, (unsigned)n);
    hipLaunchKernelGGL(k_scatter, dim3(grid), dim3(BLK), 0, st, d_cell, (long long)n, T.cell_offset, d_cursor, d_sortidx);
    HIPCHECK(hipEventRecord(h->ev1, st));

    // reorder into cell order
    double **dp[] = {&T.x, &T.y, &T.z, &T.vx, &T.vy, &T.vz, &T.m};
    for (double **p : dp) HIPCHECK(hipMalloc((void **)p, (size_t)n * 8));
    if (T.grnr64) HIPCHECK(hipMalloc((void **)&T.g64, (size_t)n * 8));
    else HIPCHECK(hipMalloc((void **)&T.g32, (size_t)n * 4));
    HIPCHECK(hipMalloc((void **)&T.fof, (size_t)n * 8));
    hipEvent_t ev2;
    HIPCHECK(hipEventCreate(&ev2));
    hipLaunchKernelGGL(k_reorder, dim3(grid), dim3(BLK), 0, st, d_sortidx, (long long)n, dpos, dvel,
                       dmass, dgr, (int)grnr_bytes, dfof, T.x, T.y, T.z, T.vx, T.vy, T.vz, T.m,
                       T.g32, T.g64, T.fof);
    HIPCHECK(hipEventRecord(ev2, st));
    HIPCHECK(hipStreamSynchronize(st));
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) { cleanup(); return fail(-101, std::string("mesh build: ") + hipGetErrorString(le)); }
    float ms1 = 0, ms2 = 0;
    (void)hipEventElapsedTime(&ms1, h->ev0, h->ev1);
    (void)hipEventElapsedTime(&ms2, h->ev1, ev2);
    (void)hipEventDestroy(ev2);
    h->ms_mesh = ms1;
    h->ms_reorder = ms2;
    T.orig = d_sortidx;   // original index of the particle now at position p
    (void)hipFree(d_cell); (void)hipFree(d_count); (void)hipFree(d_cursor); (void)hipFree(d_tilesum);
    cleanup();
    return 0;
}

int32_t soap_engine_stage_halos(soap_engine *h, int64_t n, const double *cofp,
                                const double *search_radius, const double *read_radius,
                                const int64_t *index, const int32_t *is_central,
                                const int64_t *nr_bound_part) {
    if (!h) return fail(-1, "null handle");
    if (n < 0) return fail(-2, "negative halo count");
    HIPCHECK(hipSetDevice(h->cfg.device));
    free_halos(h);
    h->n_halo = n;
    if (n == 0) return 0;
    if (!cofp || !search_radius || !read_radius || !index || !is_central || !nr_bound_part)
        return fail(-3, "null halo array");
    for (int64_t i = 0; i < n; i++) {
        if (!(search_radius[i] > 0.0) || !(read_radius[i] >= search_radius[i]))
            return fail(-4, "need 0 < search_radius <= read_radius (halo_tasks.py:78)");
    }
    HIPCHECK(hipMalloc((void **)&h->d_cofp, (size_t)n * 24));
    HIPCHECK(hipMalloc((void **)&h->d_search, (size_t)n * 8));
    HIPCHECK(hipMalloc((void **)&h->d_read, (size_t)n * 8));
    HIPCHECK(hipMalloc((void **)&h->d_index, (size_t)n * 8));
    HIPCHECK(hipMalloc((void **)&h->d_nrbound, (size_t)n * 8));
    HIPCHECK(hipMalloc((void **)&h->d_stats, (size_t)n * 32));
    HIPCHECK(hipMalloc((void **)&h->d_central, (size_t)n * 4));
    HIPCHECK(hipMalloc((void **)&h->d_status, (size_t)n * 4));
    HIPCHECK(hipMalloc((void **)&h->d_todo, (size_t)n * 4));
    hipStream_t st = h->stream;
    HIPCHECK(hipMemcpyAsync(h->d_cofp, cofp, (size_t)n * 24, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(h->d_search, search_radius, (size_t)n * 8, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(h->d_read, read_radius, (size_t)n * 8, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(h->d_index, index, (size_t)n * 8, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(h->d_nrbound, nr_bound_part, (size_t)n * 8, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(h->d_central, is_central, (size_t)n * 4, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemsetAsync(h->d_status, 0, (size_t)n * 4, st));
    HIPCHECK(hipMemsetAsync(h->d_stats, 0, (size_t)n * 32, st));
    HIPCHECK(hipStreamSynchronize(st));
    return 0;
}

static long long next_pow2(long long v) { long long p = 1; while (p < v) p <<= 1; return p; }

int32_t soap_engine_run(soap_engine *h, const soap_calc_desc *calcs, int32_t n_calcs,
                        double target_density) {
    if (!h) return fail(-1, "null handle");
    if (n_calcs < 1 || n_calcs > MAXCALC) return fail(-2, "number of calculations must be in [1,48]");
    if (!calcs) return fail(-3, "null calcs");
    if (calcs[0].kind != SOAP_CALC_SUBHALO)
        return fail(-4, "the first calculation must be the bound subhalo (compute_halo_properties.py:232-247)");
    HIPCHECK(hipSetDevice(h->cfg.device));
    std::vector<int> coloff(n_calcs);
    long long ncols = 0;
    for (int c = 0; c < n_calcs; c++) {
        int nc = soap_engine_calc_columns(calcs[c].kind);
        if (nc < 0) return fail(-5, "unknown calculation kind");
        if (c > 0 && calcs[c].kind == SOAP_CALC_SUBHALO) return fail(-6, "only one subhalo calculation allowed");
        if (calcs[c].kind == SOAP_CALC_SO_MULTIPLE) {
            int p = calcs[c].parent;
            if (p < 1 || p >= c || calcs[p].kind < SOAP_CALC_SO_DENSITY || calcs[p].kind > SOAP_CALC_SO_MULTIPLE)
                return fail(-7, "SO radius multiple needs an earlier SO as parent (SO_properties.py:3912-3916)");
        }
        if ((calcs[c].kind == SOAP_CALC_APERTURE_EXCL || calcs[c].kind == SOAP_CALC_APERTURE_INCL) && calcs[c].parent >= 0) {
            int p = calcs[c].parent;
            if (p < 1 || p >= c || calcs[p].kind != calcs[c].kind) return fail(-8, "aperture parent must be an earlier aperture of the same kind");
        }
        coloff[c] = (int)ncols;
        ncols += nc;
    }
    hipStream_t st = h->stream;
    if (h->d_table) { (void)hipFree(h->d_table); h->d_table = nullptr; }
    if (h->d_calcs) { (void)hipFree(h->d_calcs); h->d_calcs = nullptr; }
    if (h->d_coloff) { (void)hipFree(h->d_coloff); h->d_coloff = nullptr; }
    h->ncols = ncols;
    if (h->n_halo == 0) return 0;
    HIPCHECK(hipMalloc((void **)&h->d_table, (size_t)h->n_halo * ncols * 8));
    HIPCHECK(hipMalloc((void **)&h->d_calcs, (size_t)n_calcs * sizeof(soap_calc_desc)));
    HIPCHECK(hipMalloc((void **)&h->d_coloff, (size_t)n_calcs * 4));
    HIPCHECK(hipMemsetAsync(h->d_table, 0, (size_t)h->n_halo * ncols * 8, st));
    HIPCHECK(hipMemcpyAsync(h->d_calcs, calcs, (size_t)n_calcs * sizeof(soap_calc_desc), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(h->d_coloff, coloff.data(), (size_t)n_calcs * 4, hipMemcpyHostToDevice, st));

    long long ntot = 0;
    for (int t = 0; t < SOAP_NTYPES; t++) ntot += h->t[t].n;
    long long cap_max = next_pow2(std::max<long long>(ntot, 64));
    long long cap = h->cfg.slab_capacity > 0 ? next_pow2(h->cfg.slab_capacity) : (1ll << 18);
    cap = std::min(cap, cap_max);
    int wgs = h->cfg.n_workgroups > 0 ? h->cfg.n_workgroups : 1024;
    wgs = (int)std::min<long long>(wgs, std::max<long long>(h->n_halo, 1));

    RunParams P;
    memset(&P, 0, sizeof(P));
    for (int t = 0; t < SOAP_NTYPES; t++) {
        const TypeHost &T = h->t[t];
        TypeDev &D = P.t[t];
        D.n = T.n; D.res = T.res; D.grnr64 = T.grnr64;
        D.x = T.x; D.y = T.y; D.z = T.z; D.vx = T.vx; D.vy = T.vy; D.vz = T.vz; D.m = T.m;
        D.grnr32 = T.g32; D.grnr64p = T.g64; D.fof = T.fof; D.orig = T.orig; D.cell_offset = T.cell_offset;
        for (int d = 0; d < 3; d++) { D.pos_min[d] = T.pos_min[d]; D.pos_max[d] = T.pos_max[d]; D.cell_size[d] = T.cell_size[d]; }
        P.soft[t] = h->cfg.softening[t];
    }
    P.halo.n = h->n_halo; P.halo.cofp = h->d_cofp; P.halo.search_radius = h->d_search; P.halo.read_radius = h->d_read;
    P.halo.index = h->d_index; P.halo.is_central = h->d_central; P.halo.nr_bound = h->d_nrbound;
    P.halo.status = h->d_status; P.halo.stats = h->d_stats; P.halo.todo = nullptr; P.halo.n_todo = 0;
    P.boxsize = h->cfg.boxsize; P.G_eff = h->cfg.G_eff; P.nu_density = h->cfg.nu_density; P.mpc20 = h->cfg.mpc20;
    P.a = h->cfg.a; P.mpc = h->cfg.mpc;
    P.target_density = target_density;
    P.ncalc = n_calcs; P.ncols = ncols; P.table = h->d_table; P.queue = h->d_queue;
    P.calcs = h->d_calcs; P.coloff = h->d_coloff;
    P.phase = h->d_phase;
    HIPCHECK(hipMemsetAsync(h->d_phase, 0, 8 * sizeof(unsigned long long), st));

    std::vector<int> status(h->n_halo);
    std::vector<int> todo;
    double ms_total = 0.0;
    for (int pass = 0; pass < 8; pass++) {
        if (h->slab_cap != cap || h->slab_wgs != wgs) {
            int rc = alloc_slabs(h, cap, wgs);
            if (rc) return rc;
        }
        P.slab = h->slab;
        HIPCHECK(hipMemsetAsync(h->d_queue, 0, sizeof(unsigned), st));
        HIPCHECK(hipMemcpyAsync(h->d_params, &P, sizeof(RunParams), hipMemcpyHostToDevice, st));
        HIPCHECK(hipEventRecord(h->ev0, st));
        hipLaunchKernelGGL(k_halo, dim3(wgs), dim3(BLK), 0, st, (const RunParams *)h->d_params);
        HIPCHECK(hipEventRecord(h->ev1, st));
        HIPCHECK(hipMemcpyAsync(status.data(), h->d_status, (size_t)h->n_halo * 4, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) return fail(-101, std::string("halo kernel: ") + hipGetErrorString(le));
        float ms = 0;
        (void)hipEventElapsedTime(&ms, h->ev0, h->ev1);
        ms_total += ms;
        // halos whose search sphere did not fit the slab: retry with larger slabs, fewer workgroups
        todo.clear();
        for (long long i = 0; i < h->n_halo; i++)
            if (status[i] == SOAP_HALO_FATAL_SLAB_OVERFLOW) todo.push_back((int)i);
        if (todo.empty() || cap >= cap_max) break;
        cap = std::min(cap * 8, cap_max);
        wgs = std::max(1, std::min<int>((int)todo.size(), wgs / 8));
        HIPCHECK(hipMemcpyAsync(h->d_todo, todo.data(), todo.size() * 4, hipMemcpyHostToDevice, st));
        P.halo.todo = h->d_todo;
        P.halo.n_todo = (long long)todo.size();
    }
    h->ms_halo = ms_total;
    HIPCHECK(hipMemcpy(h->phase, h->d_phase, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}

int64_t soap_engine_row_columns(const soap_engine *h) { return h ? h->ncols : -1; }
const double *soap_engine_table_device(const soap_engine *h) { return h ? h->d_table : nullptr; }

int32_t soap_engine_copy_table_device(soap_engine *h, double *dst) {
    if (!h || !dst) return fail(-1, "null argument");
    if (!h->d_table) return fail(-2, "no run to copy");
    HIPCHECK(hipSetDevice(h->cfg.device));
    HIPCHECK(hipMemcpyAsync(dst, h->d_table, (size_t)h->n_halo * h->ncols * 8, hipMemcpyDeviceToDevice, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    return 0;
}

int32_t soap_engine_fetch(soap_engine *h, double *table, int32_t *status, double *search_radius,
                          double *read_radius, int64_t *stats) {
    if (!h) return fail(-1, "null handle");
    HIPCHECK(hipSetDevice(h->cfg.device));
    size_t n = (size_t)h->n_halo;
    if (n == 0) return 0;
    hipStream_t st = h->stream;
    if (table) {
        if (!h->d_table) return fail(-2, "no run to fetch");
        HIPCHECK(hipMemcpyAsync(table, h->d_table, n * h->ncols * 8, hipMemcpyDeviceToHost, st));
    }
    if (status) HIPCHECK(hipMemcpyAsync(status, h->d_status, n * 4, hipMemcpyDeviceToHost, st));
    if (search_radius) HIPCHECK(hipMemcpyAsync(search_radius, h->d_search, n * 8, hipMemcpyDeviceToHost, st));
    if (read_radius) HIPCHECK(hipMemcpyAsync(read_radius, h->d_read, n * 8, hipMemcpyDeviceToHost, st));
    if (stats) HIPCHECK(hipMemcpyAsync(stats, h->d_stats, n * 32, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    return 0;
}

int64_t soap_engine_membership(soap_engine *h, int64_t row, int32_t ts, double radius,
                               int32_t strict, int32_t bound_only, int64_t *out, int64_t cap) {
    if (!h) return fail(-1, "null handle");
    if (row < 0 || row >= h->n_halo) return fail(-2, "row out of range");
    if (ts < 0 || ts >= SOAP_NTYPES) return fail(-3, "type_slot out of range");
    HIPCHECK(hipSetDevice(h->cfg.device));
    const TypeHost &T = h->t[ts];
    if (T.n == 0) return 0;
    double c[3];
    long long hindex;
    HIPCHECK(hipMemcpy(c, h->d_cofp + 3 * row, 24, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(&hindex, h->d_index + row, 8, hipMemcpyDeviceToHost));
    TypeDev D;
    memset(&D, 0, sizeof(D));
    D.n = T.n; D.grnr64 = T.grnr64; D.x = T.x; D.y = T.y; D.z = T.z; D.grnr32 = T.g32; D.grnr64p = T.g64; D.orig = T.orig;
    unsigned *d_cnt;
    long long *d_out = nullptr;
    HIPCHECK(hipMalloc((void **)&d_cnt, 4));
    HIPCHECK(hipMemset(d_cnt, 0, 4));
    if (cap > 0) HIPCHECK(hipMalloc((void **)&d_out, (size_t)cap * 8));
    int grid = (int)std::min<long long>((T.n + BLK - 1) / BLK, 2048);
    hipLaunchKernelGGL(k_membership, dim3(grid), dim3(BLK), 0, h->stream, D, c[0], c[1], c[2],
                       h->cfg.boxsize, radius, (int)strict, (int)bound_only, hindex, d_cnt, d_out, (long long)cap);
    unsigned cnt = 0;
    HIPCHECK(hipMemcpyAsync(&cnt, d_cnt, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    long long nw = std::min<long long>(cnt, cap);
    if (nw > 0 && out) {
        HIPCHECK(hipMemcpy(out, d_out, (size_t)nw * 8, hipMemcpyDeviceToHost));
        std::sort(out, out + nw);
    }
    (void)hipFree(d_cnt);
    if (d_out) (void)hipFree(d_out);
    return (int64_t)cnt;
}

int32_t soap_engine_phase_cycles(const soap_engine *h, uint64_t *out8) {
    if (!h || !out8) return fail(-1, "null argument");
    for (int i = 0; i < 8; i++) out8[i] = h->phase[i];
    return 0;
}

double soap_engine_kernel_ms(const soap_engine *h, const char *name) {
    if (!h || !name) return -1.0;
    if (!strcmp(name, "halo")) return h->ms_halo;
    if (!strcmp(name, "mesh")) return h->ms_mesh;
    if (!strcmp(name, "reorder")) return h->ms_reorder;
    return -1.0;
}

}  // extern "C"
