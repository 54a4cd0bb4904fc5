// Shared by engine.hip (host side, mesh kernels) and halo_class.hip (the per-halo kernels, one translation unit per
// thread-group class so that they compile in parallel): device-side views of the staged data, wave primitives,
// the moment accumulator layout.
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>
#include <functional>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <algorithm>
#include <string>
#include <vector>
#include <map>

#include "../../include/soap_engine.h"
#include "device_math.h"
#include "layout.h"

#pragma clang fp contract(off)

#define BLK 256
#define NWAVE (BLK / 64)
#define CH 4096            // LDS sort chunk (elements)
#define PIDX_BITS 29   // packed slab index: 3 bits of type slot, 29 bits of cell-ordered position
#define PIDX_MASK ((1u << PIDX_BITS) - 1u)
#define MAXCALC 64   // calculations per run: one lane of a wavefront each in the radius loop, one bit of the 64-bit done mask

// flags byte of a slab element
#define FL_TYPE 0x7    // type slot 0..3, or NU_SLOT
#define FL_BOUND 0x8
#define FL_SAT 0x10
#define FL_EXT 0x20
// Neutrino particles (PartType6) are a fifth, internal type slot: they enter the SO mass profile with their
// weighted mass and nothing else (SO_properties.py:380-395); no velocities, no group membership.
#define NU_SLOT 4
#define NT_ALL 5

// ------------------------------------------------------------------------------------------
// device-side views
// ------------------------------------------------------------------------------------------
// Everything the property passes need of one particle, in ONE record at its mesh-cell-ordered position: the slab
// build fetches five 16-byte words (two cache lines) per selected particle instead of eight scattered 8-byte fields
// from eight arrays.  For the neutrino slot vx holds the raw (unweighted) mass and m the weighted one.
struct alignas(16) Pay {
    double vx, vy, vz, m;
    long long grnr, fof;
    double x, y, z;
    long long orig;     // index of the particle in the caller's arrays (what the membership probe reports)
};
static_assert(sizeof(Pay) == 80, "five 16-byte words");

struct TypeDev {
    long long n;
    int res;
    int grnr64;
    const double *x, *y, *z;      // coordinates in mesh-cell order (SoA): what the candidate stream reads
    const double *vel, *mass;     // the caller's arrays (AoS velocities), indexed through `orig`; for NU_SLOT: vel = raw
                                  // (unweighted) masses, mass = mass * weight
    const int *grnr32;
    const long long *grnr64p;
    const long long *fof;
    const unsigned *orig;         // original index of the particle at cell-ordered position p
    const unsigned *cell_offset;  // res^3 + 1 entries
    const Pay *pay;               // payload records in cell order
    double pos_min[3], pos_max[3], cell_size[3];
    double mean_mass;             // of a thinned sample of the type's particles (the scheduling guess of k_predict)
    // thinned copies of the type's particles for k_predict, levels 0 / 1 = one particle in 64 / in 1024 (chosen by thin_hash of its
    // coordinates): coordinates in mesh-cell order and the cells' offsets into them
    const unsigned *thin_off[2];
    const double *thin_x[2], *thin_y[2], *thin_z[2];
};

struct SlabDev {
    long long cap;
    double *r, *m, *dx, *dy, *dz, *vx, *vy, *vz, *cum, *q, *dens, *pr;
    unsigned *pidx, *pi;
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
    long long *stats;  // [n][SOAP_STATS_COLUMNS]
    const int *todo;   // optional list of rows to process (retry pass), or null
    long long n_todo;
    const double *resident;   // optional: radius inside which the staged particles are complete (>= read_radius)
    const double *first_radius;   // optional: radius the first sorted superset of a halo should reach (k_predict; 0 = the default)
    int *part_count;              // halos that several groups of the single-kernel path share (stage_b's shares): groups that have finished, per row (zeroed per run), or null
};

// Per-halo record of the three-stage pipeline (build -> radius loop -> property passes): what the stages hand to
// each other.  The sorted slab of a halo lives in its own slice [off, off + cap) of the arena (same columns as
// the per-group slabs of the single-kernel path).
#define PIPE_PENDING 0
#define PIPE_BUILT 1         // sorted superset in the arena, header filled
#define PIPE_DONE 2          // radius loop finished: status / selections final
#define PIPE_NEED_SUPERSET 3 // the loop ran past the superset: build again out to R_need
#define PIPE_FALLBACK 4      // this halo goes to the single-kernel path (radius ties, inconsistent bound counts, ...)
struct HaloHdr {
    long long off;          // first arena entry of this halo
    int cap;                // arena entries reserved
    int pipe_state;
    // ---- written by the build stage
    int n_have;             // particles in the sorted superset (every particle with r2 <= R_have^2)
    int q_sorted;           // minimum-image r^2 is non-decreasing in slab order
    double R_have;
    double R_build;         // radius to build the superset out to (0: the default first superset)
    double shift[3];        // velocity of the innermost particle
    long long fof0;         // its FOF id
    int i_firstpos;         // first entry with r > 0 (n_have if none)
    int i_lastbound;        // last entry bound to the halo (-1 if none)
    int i_firstdm;          // first dark matter entry (n_have if none)
    int n_restart;
    double nb[4];           // bound particles per type slot in the superset
    double encl;            // radius of the outermost bound particle
    int cs_fb[MAXCALC];     // per SO density calculation (in list order of those): first entry >= 1 not above its threshold
    // ---- written by the radius loop
    int status;
    int n_final;            // particles in the final search sphere
    long long n_loop, n_loop_all;
    double R_final, R_search, R_read, R_need;
    int cs_state[MAXCALC], cs_n[MAXCALC];
    double cs_R[MAXCALC], cs_M[MAXCALC];
    // ---- written by k_bpartials (large halos): moment sums of whole tiles of the final search sphere, for both families
    // of calculations: bpart[(2 * tile + family) * NACC + k]; null if there are none
    double *bpart;
    int bpart_tiles;        // complete tiles of GTILE entries inside the final search sphere
    // ---- written by k_bscan: per complete tile and family BSCAN_W doubles -- cumulative (mass, dark matter mass, count, gas
    // mass, stellar mass) of the family's members before the tile [0..4] and after it [5..9], the tile's largest
    // M(<r) / max(r, softening) with that radius [10, 11] and, bound family, its largest M(<r) / r with radius and entry [12..14]
    double *bscan;
};
#define BSCAN_W 16

// Scratch of the tile-parallel build of one LARGE halo (several workgroups per halo, kernels k_g* of
// halo_kernel.inc): append cursor of the selection, the sample sort's splitters / bucket fills / bucket groups, and
// the slab facts that are reduced with atomics.
#define GTILE 4096       // slab entries per tile (one 256-thread workgroup each)
#define GNBMAX 2048      // most buckets of the distributed sample sort
struct GBuild {
    int row;             // halo row
    int first_tile, ntile;     // tiles covering the halo's arena slice
    int first_gblock, ngblock; // blocks of the bucket-group sort reserved for it
    int first_sel, nsel;       // parts of the selection (whole candidate runs are dealt round-robin to them)
    unsigned n;          // particles selected (append cursor)
    int nb, ngroup, overflow;
    int q_unsorted, i_firstpos, i_lastbound, i_firstdm;
    int nbi[4];
    int unequal;         // the masses of the slab are not all the same value
    // giants (more entries than GNBMAX buckets of a sample sort can split evenly): buckets are uniform bins in radius,
    // their fills / starts / the starts of the bucket groups live in RunParams.gbins from bin_first on
    int nbin, bin_first;
    double bin_scale;    // bin = min(nbin - 1, int(r * bin_scale))
    // cumulative mass of a slab of equal masses in closed form (k_gcumsum): entry q covers slab entries
    // [cum_k[q], cum_k[q] + cum_cnt[q]) with cum[cum_k[q] + t - 1] = cum_x[q] + t cum_d[q]; the tiles fill them in (k_gpost)
    int cum_ne;
    int cum_k[72], cum_cnt[72];
    double cum_x[72], cum_d[72];
    double splitter[GNBMAX];
    unsigned bcount[GNBMAX];
    int bstart[GNBMAX + 1];
    int gstart[GNBMAX + 1];
};

// Control block of the team kernel (k_team, halo_kernel.inc): the queue of large halos and one slot per team -- the team's
// arrival counter (monotonic over the launch) and the list entry it works on -- each on a 128-byte line of its own.
#define TEAM_MAX 64
struct TeamCtl {
    unsigned next;          // next entry of the list of large halos
    unsigned abort;         // a team barrier timed out (a bug, or a team member that never became resident): every group leaves
    unsigned pad0[30];
    unsigned long long phase[32];   // shader clocks of member 0 per phase, summed over the halos (diagnostic)
    struct Slot { unsigned arrive, task, pad[30]; } team[TEAM_MAX];
};

struct RunParams {
    TypeDev t[NT_ALL];
    SlabDev slab;
    HaloDev halo;
    double boxsize, G_eff, nu_density, mpc20, target_density, a, mpc;
    double G_phys, hubble, rdot_omega;   // pseudo-evolution term of the shell flow rates (soap_engine_config)
    int literal;     // 1: rebuild the search sphere at every radius round (reference-literal path)
    double first_factor;   // first sorted superset reaches first_factor * search_radius
    int profile;           // 1: accumulate per-phase clocks (env SOAP_ENGINE_PROFILE)
    int hydro;             // gas or star particles staged: the baryon kinematics are computed
    int dm_only;           // only dark matter staged
    int has_nu;            // neutrino particles staged
    int mass_f32;          // every staged mass array is float32: the SO mass profile is then the float64 cumulative mass cast back to
                           // float32, as the reference computes it (SO_properties.py:400-402: cumsum(dtype=float64).astype(mass.dtype))
    double soft[SOAP_NTYPES];
    int ncalc;
    long long ncols;
    double *table;
    unsigned *queue;
    unsigned long long *phase;   // [32] accumulated s_memtime cycles of thread 0 per phase
    const soap_calc_desc *calcs;  // device copy
    const int *coloff;            // device: column offset per calc
    // member probe (soap_engine_set_member_probe): halo row / calculation to record, packed (type slot << 56 | original index)
    long long probe_row;
    int probe_calc;
    long long probe_cap;
    unsigned long long *probe_out;
    unsigned *probe_count;
    HaloHdr *hdr;                 // pipeline path: one record per staged halo (null on the single-kernel path)
    GBuild *gb;                   // pipeline path: scratch of the large halos' tile-parallel build
    const int *tile_li, *gblock_li, *sel_li;   // tile / sort block / selection part -> entry of gb
    int *gbins;                   // radial bins of the giants' build (GBuild.bin_first)
    double rebuild_factor;        // a halo whose loop ran past its superset is rebuilt out to this times the radius it asked for
    int g_coarse;                 // test hook (SOAP_ENGINE_G_COARSE=1): four buckets per large halo, so that every bucket outgrows its LDS block
    double *small_cols;           // small-halo class with its streamed columns in global memory: seven columns of slab entries per wave
    int *small_esc;               // small-halo class: [0] number of rows in the list behind it -- halos that outgrew the class' slab are appended
                                  // and done by the wavefront class, launched behind the small-halo kernels with this list (null: status only)
    double *small_cols2;          // ... the streamed columns of the 1024-entry variant (the 512-entry one uses small_cols)
    double *small_cols0;          // ... and of the 128-entry variant when it is built with its columns in global memory
};

// ------------------------------------------------------------------------------------------
// wave / block primitives
// ------------------------------------------------------------------------------------------
// Cross-lane primitives on DPP (data-parallel primitives: VALU-rate lane moves inside rows of 16 lanes,
// row broadcasts across rows).  HIP's __shfl_* lower to ds_bpermute, i.e. a round trip through the LDS
// crossbar per 32-bit half and step -- rocprof showed the halo kernels latency-bound with most LDS
// instructions being exactly those shuffles.
template <int CTRL, int ROW_MASK = 0xf>
SOAP_DEV double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
#define DPP_QUAD_XOR1 0xB1      // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2 0x4E      // quad_perm [2,3,0,1]
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_ROR(n) (0x120 + (n))
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143

SOAP_DEV double readlane_f64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// gfx950 lane swaps across rows: v_permlane16_swap exchanges the odd 16-lane rows of one register with the
// even rows of another, v_permlane32_swap the upper half of one with the lower half of another.  Fed the
// same register twice they return (even-row data everywhere, odd-row data everywhere) resp. (lower-half data
// everywhere, upper-half data everywhere): a cross-row butterfly step in two VALU instructions per dword,
// no LDS crossbar (ds_bpermute) round trip.
SOAP_DEV void rows16_f64(double v, double &even, double &odd) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    even = __hiloint2double(rh[0], rl[0]);
    odd = __hiloint2double(rh[1], rl[1]);
}
SOAP_DEV void halves32_f64(double v, double &lower, double &upper) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    lower = __hiloint2double(rh[0], rl[0]);
    upper = __hiloint2double(rh[1], rl[1]);
}

// Wave totals of FOUR values for the price of one: v_permlane32_swap(x, y) leaves x = {x[0:31], y[0:31]} and
// y = {x[32:63], y[32:63]}, so x + y holds the half-sums of x in the lower and those of y in the upper 32
// lanes; the same step on 16-lane rows packs four values, one per row, and the in-row DPP butterfly
// finishes all four together.  Returns a register whose row r (lanes 16r..16r+15) holds the total of v_r.
SOAP_DEV double swap32_add(double x, double y) {
    auto rl = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(y), false, false);
    auto rh = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(y), false, false);
    return __hiloint2double(rh[0], rl[0]) + __hiloint2double(rh[1], rl[1]);
}
SOAP_DEV double swap16_add(double x, double y) {
    auto rl = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(y), false, false);
    auto rh = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(y), false, false);
    return __hiloint2double(rh[0], rl[0]) + __hiloint2double(rh[1], rl[1]);
}
SOAP_DEV double wave_sum4(double v0, double v1, double v2, double v3) {
    double a = swap32_add(v0, v2);   // lower half: v0, upper half: v2
    double b = swap32_add(v1, v3);   // lower half: v1, upper half: v3
    double c = swap16_add(a, b);     // rows: v0, v1, v2, v3
    c += dpp_f64<DPP_QUAD_XOR1>(c);
    c += dpp_f64<DPP_QUAD_XOR2>(c);
    c += dpp_f64<DPP_ROW_HALF_MIRROR>(c);
    c += dpp_f64<DPP_ROW_MIRROR>(c);
    return c;
}
// store the four totals of wave_sum4 into dst[k0..k3] (one lane per row writes)
SOAP_DEV void put4(double *dst, double packed, int k0, int k1, int k2, int k3) {
    const int lane = threadIdx.x & 63, row = lane >> 4;
    if ((lane & 15) == 0) dst[row == 0 ? k0 : (row == 1 ? k1 : (row == 2 ? k2 : k3))] = packed;
}

// sum over the 64 lanes, same value returned to every lane (fixed association:
// ((row0 + row1) + (row2 + row3)) of the in-row butterflies)
SOAP_DEV double wave_sum(double v) {
    v += dpp_f64<DPP_QUAD_XOR1>(v);
    v += dpp_f64<DPP_QUAD_XOR2>(v);
    v += dpp_f64<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_f64<DPP_ROW_MIRROR>(v);        // every lane of a 16-lane row holds the row sum
    double a, b;
    rows16_f64(v, a, b);
    v = a + b;
    halves32_f64(v, a, b);
    return a + b;
}
SOAP_DEV double wave_max(double v) {
    v = fmax(v, dpp_f64<DPP_QUAD_XOR1>(v));
    v = fmax(v, dpp_f64<DPP_QUAD_XOR2>(v));
    v = fmax(v, dpp_f64<DPP_ROW_HALF_MIRROR>(v));
    v = fmax(v, dpp_f64<DPP_ROW_MIRROR>(v));
    double a, b;
    rows16_f64(v, a, b);
    v = fmax(a, b);
    halves32_f64(v, a, b);
    return fmax(a, b);
}
SOAP_DEV long long wave_min_ll(long long v) {
    // same butterfly on the bit pattern (a double register pair is just 64 bits here)
    auto mn = [](long long x, long long y) { return y < x ? y : x; };
    auto asd = [](long long x) { return __longlong_as_double(x); };
    auto asl = [](double x) { return __double_as_longlong(x); };
    v = mn(v, asl(dpp_f64<DPP_QUAD_XOR1>(asd(v))));
    v = mn(v, asl(dpp_f64<DPP_QUAD_XOR2>(asd(v))));
    v = mn(v, asl(dpp_f64<DPP_ROW_HALF_MIRROR>(asd(v))));
    v = mn(v, asl(dpp_f64<DPP_ROW_MIRROR>(asd(v))));
    double a, b;
    rows16_f64(asd(v), a, b);
    v = mn(asl(a), asl(b));
    halves32_f64(asd(v), a, b);
    return mn(asl(a), asl(b));
}
// DPP move that leaves a lane's own value where the pattern has no source lane (identity for scans with an
// operator other than +)
template <int CTRL, int ROW_MASK = 0xf>
SOAP_DEV double dpp_f64_keep(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
#define DPP_WAVE_SHR1 0x138     // wave_shr:1 (whole wavefront, lane 0 keeps its value)
// inclusive prefix arg-max inside a wave: (v, s) becomes the pair with the largest v among lanes <= this one,
// ties going to the lowest lane ("first maximum")
SOAP_DEV void wave_scan_argmax(double &v, double &s) {
#define ARGMAX_STEP(CTRL, MASK) { double pv = dpp_f64_keep<CTRL, MASK>(v), ps = dpp_f64_keep<CTRL, MASK>(s); if (!(v > pv)) { v = pv; s = ps; } }
    ARGMAX_STEP(DPP_ROW_SHR(1), 0xf)
    ARGMAX_STEP(DPP_ROW_SHR(2), 0xf)
    ARGMAX_STEP(DPP_ROW_SHR(4), 0xf)
    ARGMAX_STEP(DPP_ROW_SHR(8), 0xf)
    ARGMAX_STEP(DPP_ROW_BCAST15, 0xa)
    ARGMAX_STEP(DPP_ROW_BCAST31, 0xc)
#undef ARGMAX_STEP
}

// inclusive scan inside a wave (row_shr with zero fill inside rows of 16, then row broadcasts)
SOAP_DEV double wave_scan(double v, int lane) {
    (void)lane;
    v += dpp_f64<DPP_ROW_SHR(1)>(v);
    v += dpp_f64<DPP_ROW_SHR(2)>(v);
    v += dpp_f64<DPP_ROW_SHR(4)>(v);
    v += dpp_f64<DPP_ROW_SHR(8)>(v);
    v += dpp_f64<DPP_ROW_BCAST15, 0xa>(v);   // last lane of rows 0 and 2 into rows 1 and 3
    v += dpp_f64<DPP_ROW_BCAST31, 0xc>(v);   // lane 31 into rows 2 and 3
    return v;
}

// inclusive running maximum inside a wave (values >= 0: the zero fill of row_shr is the identity)
SOAP_DEV double wave_scan_max(double v) {
    v = fmax(v, dpp_f64<DPP_ROW_SHR(1)>(v));
    v = fmax(v, dpp_f64<DPP_ROW_SHR(2)>(v));
    v = fmax(v, dpp_f64<DPP_ROW_SHR(4)>(v));
    v = fmax(v, dpp_f64<DPP_ROW_SHR(8)>(v));
    v = fmax(v, dpp_f64_keep<DPP_ROW_BCAST15, 0xa>(v));
    v = fmax(v, dpp_f64_keep<DPP_ROW_BCAST31, 0xc>(v));
    return v;
}

// ------------------------------------------------------------------------------------------
// per-halo kernel
// ------------------------------------------------------------------------------------------
#define NACC 76
#ifndef PREFB
#define PREFB 12        // calculations whose prefix moments are finalised together
#endif
#define NHM 5           // half-mass weight classes: all, dark matter, gas, stars, baryons
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
    A_RMAX = 37,                       // max r  (max-reduction)
    // gas (type slot 0) and stars (type slot 2), only accumulated when the chunk has baryons:
    // sum m*x (3), sum m*v' (3), sum m * x cross v' (3), sum m v'_a v'_b (6)
    A_GMX = 38, A_GMV = 41, A_GL = 44, A_GVV = 47,
    A_SMX = 53, A_SMV = 56, A_SL = 59, A_SVV = 62,
    // neutrino particles inside the selection: count, raw mass, weighted mass, weighted mass * radius
    A_NUN = 68, A_NUM = 69, A_NUW = 70, A_NUR = 71,
    A_NSUM = 72                        // accumulators in use (the last slots of NACC are scratch)
};

#define GAS __attribute__((address_space(1)))

// ---- views with pointers typed as global memory (address_space(1): global_load/store instead of flat) ----
struct TypeG {   // same layout as TypeDev, pointers typed as global
    long long n;
    int res;
    int grnr64;
    const GAS double *x, *y, *z;
    const GAS double *vel, *mass;
    const GAS int *grnr32;
    const GAS long long *grnr64p;
    const GAS long long *fof;
    const GAS unsigned *orig;
    const GAS unsigned *cell_offset;
    const GAS Pay *pay;
    double pos_min[3], pos_max[3], cell_size[3];
    double mean_mass;
    const GAS unsigned *thin_off[2];
    const GAS double *thin_x[2], *thin_y[2], *thin_z[2];
};
static_assert(sizeof(TypeG) == sizeof(TypeDev), "TypeG mirrors TypeDev");

// Is the particle at (x, y, z) in the thinned sample?  A hash of the coordinate bits: a property of the particle, not of where the
// placement pass put it inside its cell (that order comes out of atomics and differs from run to run).  Returns the hash; the
// sample 1-in-64 is (h & 63) == 0, 1-in-1024 (h & 1023) == 0.
SOAP_DEV unsigned thin_hash(double x, double y, double z) {
    // (one integer multiply: the placement pass computes this for every particle, and 32-bit multiplies run at a quarter of the rate)
    const unsigned long long bx = (unsigned long long)__double_as_longlong(x), by = (unsigned long long)__double_as_longlong(y), bz = (unsigned long long)__double_as_longlong(z);
    unsigned h = (unsigned)bx ^ (unsigned)(bx >> 32);
    h ^= __builtin_rotateleft32((unsigned)by ^ (unsigned)(by >> 32), 11);
    h ^= __builtin_rotateleft32((unsigned)bz ^ (unsigned)(bz >> 32), 22);
    h ^= h >> 15; h *= 0x9E3779B1u; h ^= h >> 16;
    return h;
}

// one payload record from global memory (field by field: the compiler merges them into 16-byte loads)
SOAP_DEV Pay load_pay(const GAS Pay *p) {
    const GAS double *d = (const GAS double *)p;
    const GAS long long *l = (const GAS long long *)p;
    Pay q;
    q.vx = d[0]; q.vy = d[1]; q.vz = d[2]; q.m = d[3];
    q.grnr = l[4]; q.fof = l[5];
    q.x = d[6]; q.y = d[7]; q.z = d[8]; q.orig = 0;   // (the property passes have no use for the original index)
    return q;
}

struct Mask256 {   // 256-bit cell mask held in registers (no dynamically indexed private array)
    unsigned long long w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    __device__ __forceinline__ void set_range(int lo, int hi) {   // bits [lo, hi]
        for (int k = 0; k < 4; k++) {
            int a = lo - 64 * k, b = hi - 64 * k;
            a = a < 0 ? 0 : a;
            b = b > 63 ? 63 : b;
            if (a > b) continue;
            unsigned long long m = (b - a == 63) ? ~0ull : (((1ull << (b - a + 1)) - 1ull) << a);
            if (k == 0) w0 |= m; else if (k == 1) w1 |= m; else if (k == 2) w2 |= m; else w3 |= m;
        }
    }
    __device__ __forceinline__ int count() const { return __popcll(w0) + __popcll(w1) + __popcll(w2) + __popcll(w3); }
    // position of the n-th set bit (n < count())
    __device__ __forceinline__ int nth(int n) const {
        unsigned long long w = w0;
        int base = 0, c = __popcll(w);
        if (n >= c) { n -= c; w = w1; base = 64; c = __popcll(w);
            if (n >= c) { n -= c; w = w2; base = 128; c = __popcll(w);
                if (n >= c) { n -= c; w = w3; base = 192; } } }
        for (int q = 0; q < n; q++) w &= w - 1;
        return base + __ffsll((long long)w) - 1;
    }
    __device__ __forceinline__ bool get(int q) const {
        const int k = q >> 6;
        const unsigned long long w = k == 0 ? w0 : (k == 1 ? w1 : (k == 2 ? w2 : w3));
        return (w >> (q & 63)) & 1ull;
    }
};


