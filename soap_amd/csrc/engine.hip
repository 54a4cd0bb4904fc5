// MI355X (gfx950) halo-property engine: mesh build + per-halo selection/property kernels
// and the C ABI declared in include/soap_engine.h.
//
// Reference behaviour being replaced (paths relative to the SOAP checkout):
//   SOAP/core/shared_mesh.py:11-200            cell-list mesh build and periodic ball query
//   SOAP/core/halo_tasks.py:23-187, 276-430    search-radius loop and calculation scheduling
//   SOAP/particle_selection/SO_properties.py   :80-217, :313-513, :532-692, :857-913, :2725-2765
//   SOAP/particle_selection/aperture_properties.py :270-317, :1099-1281, :3469-3577, :4024-4244
//   SOAP/particle_selection/subhalo_properties.py  :128-164, :798-846, :982-1073, :2587-2740
//   SOAP/property_calculation/{half_mass_radius,kinematic_properties,inertia_tensors}.py
//
// Design (DESIGN.md has the long form):
//   * the COORDINATES of each type are physically reordered into mesh-cell order (SoA in HBM) so the cells a
//     search sphere overlaps are a few contiguous runs -> coalesced reads; velocities, masses and ids stay in
//     the caller's arrays and are gathered through the permutation for the particles a halo selects;
//   * one thread group per halo -- a wavefront, 256 or 1024 threads, chosen from a cheap work estimate -- in
//     three persistent kernels (halo_kernel.inc compiled per class, once more with the inertia tensor passes)
//     that run concurrently on priority streams and pull halos heaviest-first from device queues;
//   * per halo: stream candidate runs, exact periodic distance test, compact the selected particles into
//     the group's slab, sort by radius (LDS bitonic / sample sort), ONE sequential fp64 cumulative-mass chain
//     (bit-identical to numpy.cumsum, so SO radii and hence SO membership are reproducible); the reference's
//     x1.2 radius rounds are emulated on prefixes of that sorted slab; every SO / sphere / subhalo /
//     projected-aperture calculation is then a prefix of it, evaluated in fused passes with
//     lanes = calculations for everything that is per calculation;
//   * no MFMA anywhere: the path is latency / issue bound fp64 vector work.
#include "engine_common.h"

static thread_local std::string g_err;
// The switches INTEGRATION.md lists are read with getenv() on every run.  Everything else is an EXPERIMENT knob -- group counts, cost
// models, launch orders, variants that were measured and lost -- and is honoured only by a library built with
// -DSOAP_EXPERIMENTS (SOAP_BUILD_EXPERIMENTS=1 for __graft_entry__.build(), scripts/build_variant.sh): the product has one path per class.
#ifdef SOAP_EXPERIMENTS
static const char *expt_env(const char *name) { return getenv(name); }
#else
static const char *expt_env(const char *) { return nullptr; }
#endif
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHECK(x)                                                                        \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess)                                                              \
            return fail(-100, std::string(#x) + ": " + hipGetErrorString(e_));             \
    } while (0)

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

// counter[key] += 1 for every active lane, returning the value each lane would have got from its own
// atomicAdd.  Snapshot particles arrive spatially coherent (halo by halo / cell by cell), so most lanes
// of a wavefront hit the same counter: lanes with equal keys are merged into one atomic (up to 4 distinct
// keys per wavefront; whatever is left adds 1 per lane).  The groups are found first (ballots only); then
// ONE atomic instruction is issued by the group leaders and the leftover lanes together, so the wave pays a
// single L2 round trip per call (and none at all when the result is not used: the histogram pass).
// (Tried and dropped: per-thread runs over 8 consecutive particles to cut same-address atomics further --
// the strided AoS reads cost more than the atomics saved, 9.8 ms instead of 6.2 ms for 10^8 particles.)
template <bool WANT_RESULT>
__device__ __forceinline__ unsigned wave_agg_inc(unsigned *counter, unsigned key, bool active) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(active);
    bool mine = active;
    int my_leader = -1, my_rank = 0, my_cnt = 0;
    for (int round = 0; round < 4 && todo; round++) {
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned k0 = (unsigned)__builtin_amdgcn_readlane((int)key, leader);
        const bool in = mine && key == k0;
        const unsigned long long grp = __ballot(in);
        if (in) {
            my_leader = leader;
            my_rank = __popcll(grp & ((1ull << lane) - 1ull));
            my_cnt = __popcll(grp);
            mine = false;
        }
        todo &= ~grp;
    }
    const bool is_leader = my_leader == lane;
    unsigned base = 0;
    if (is_leader || mine) {
        const unsigned addv = is_leader ? (unsigned)my_cnt : 1u;
        if (WANT_RESULT) base = atomicAdd(&counter[key], addv);
        else (void)atomicAdd(&counter[key], addv);
    }
    if (!WANT_RESULT) return 0u;
    const unsigned lb = (unsigned)__shfl((int)base, my_leader >= 0 ? my_leader : lane, 64);
    return my_leader >= 0 ? lb + (unsigned)my_rank : base;
}

struct MeshGeom {
    double pos_min[3], cell_size[3];
    int res;
};

// cell index of every particle + histogram (shared_mesh.py:69-80).  The histogram's returning add gives the
// number of particles that entered the cell before: that is the particle's rank inside its cell, so the
// placement pass needs no second round of atomics.  Adds are aggregated twice: lanes of a wavefront with the
// same cell share one LDS add, and the cells touched by a block's 256 consecutive particles (snapshots are
// stored by top-level cell, so they are few) share one global add each through a small LDS hash table.
#define CELL_SLOTS 512
#define CELL_EMPTY 0xffffffffu
__global__ void k_cellid(const double *pos, long long n, MeshGeom g, unsigned *cell,
                         unsigned *count, unsigned *rank) {
    __shared__ unsigned s_key[CELL_SLOTS], s_cnt[CELL_SLOTS], s_base[CELL_SLOTS];
    for (int t = threadIdx.x; t < CELL_SLOTS; t += BLK) { s_key[t] = CELL_EMPTY; s_cnt[t] = 0u; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long long stride = (long long)gridDim.x * BLK;
    const long long niter = (n + stride - 1) / stride;
    for (long long it = 0; it < niter; it++) {
        const long long i = it * stride + blockIdx.x * (long long)BLK + threadIdx.x;
        const bool active = i < n;
        unsigned id = 0;
        if (active) {
            int c[3];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                double q = floor((pos[3 * i + d] - g.pos_min[d]) / g.cell_size[d]);
                // .astype(np.int32) then np.clip(0, res-1)
                int ci = (q >= 2147483647.0) ? 2147483647 : (q <= -2147483648.0 ? (-2147483647 - 1) : (int)q);
                ci = ci < 0 ? 0 : (ci > g.res - 1 ? g.res - 1 : ci);
                c[d] = ci;
            }
            id = (unsigned)c[0] + (unsigned)g.res * (unsigned)c[1] + (unsigned)g.res * (unsigned)g.res * (unsigned)c[2];
            cell[i] = id;
        }
        // ---- wavefront level: groups of lanes with the same cell (up to 4 groups, the rest go alone)
        unsigned long long todo = __ballot(active);
        bool mine = active;
        int my_leader = -1, my_rank = 0, my_cnt = 0;
        for (int round = 0; round < 4 && todo; round++) {
            const int leader = __ffsll((long long)todo) - 1;
            const unsigned k0 = (unsigned)__builtin_amdgcn_readlane((int)id, leader);
            const bool in = mine && id == k0;
            const unsigned long long grp = __ballot(in);
            if (in) {
                my_leader = leader;
                my_rank = __popcll(grp & ((1ull << lane) - 1ull));
                my_cnt = __popcll(grp);
                mine = false;
            }
            todo &= ~grp;
        }
        const bool adds = my_leader == lane || mine;   // this lane adds for its group (or for itself)
        // ---- block level: one slot per distinct cell in an open-addressing LDS table, LDS add returns the group's base
        unsigned slot = 0, lbase = 0;
        if (adds) {
            unsigned hsh = (id * 2654435761u) >> 23;   // 9 bits
            while (true) {
                const unsigned prev = atomicCAS(&s_key[hsh], CELL_EMPTY, id);
                if (prev == CELL_EMPTY || prev == id) break;
                hsh = (hsh + 1u) & (CELL_SLOTS - 1);
            }
            slot = hsh;
            lbase = atomicAdd(&s_cnt[slot], my_leader == lane ? (unsigned)my_cnt : 1u);
        }
        __syncthreads();
        // ---- one global add per distinct cell of the block
        for (int t = threadIdx.x; t < CELL_SLOTS; t += BLK)
            if (s_key[t] != CELL_EMPTY) s_base[t] = atomicAdd(&count[s_key[t]], s_cnt[t]);
        __syncthreads();
        unsigned base = adds ? s_base[slot] + lbase : 0u;
        const unsigned lb = (unsigned)__shfl((int)base, my_leader >= 0 ? my_leader : lane, 64);
        if (active) rank[i] = my_leader >= 0 ? lb + (unsigned)my_rank : base;
        __syncthreads();
        for (int t = threadIdx.x; t < CELL_SLOTS; t += BLK) { s_key[t] = CELL_EMPTY; s_cnt[t] = 0u; }
        __syncthreads();
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

// the thinned samples of a type while they are built (stage_type): append cursors, per-cell counts and the unsorted lists, per level
struct ThinBuild {
    unsigned *cursor;            // [blocks of the placement pass][2]: entries of the block's sub-list, per level
    unsigned cap[2];             // entries a sub-list holds, per level
    unsigned *count[2], *lc[2];
    double *lx[2], *ly[2], *lz[2];
};
// ... into cell order: slot = offset of the cell + a rank from an atomic (the order inside a cell does not matter: k_predict counts)
__global__ void k_thin_place(const unsigned *cursor, int lv, unsigned cap, int nsub, const unsigned *lc, const double *lx, const double *ly, const double *lz,
                             unsigned *off, long long ncell, const unsigned *count, unsigned *fill, double *x, double *y, double *z) {
    if (blockIdx.x == 0 && threadIdx.x == 0) off[ncell] = off[ncell - 1] + count[ncell - 1];
    for (int sub = blockIdx.x; sub < nsub; sub += gridDim.x) {
        const unsigned n = cursor[2 * sub + lv];
        for (unsigned e = threadIdx.x; e < n; e += BLK) {
            const unsigned s = (unsigned)sub * cap + e;
            const unsigned c = lc[s];
            const unsigned pos = off[c] + atomicAdd(&fill[c], 1u);
            x[pos] = lx[s]; y[pos] = ly[s]; z[pos] = lz[s];
        }
    }
}

// Placement: particle i goes to slot p = offset[cell] + rank of its cell (order inside a cell is arbitrary: it only
// feeds candidate streaming; everything order-sensitive happens after the radial sort) and its position is
// written there (AoS input read in order -> SoA output in cell order), so no second pass has to gather the
// coordinates back through sort_idx.  Velocities, masses and ids stay where the caller put them: only
// selected particles ever need them, and those are gathered through `orig` when a halo's slab is built.
// k_cellid gave the particles of one cell inside one block iteration CONSECUTIVE ranks (one global add per
// cell and block), so their destinations are one run: the block first regroups its 256 particles by
// destination run in LDS (hash table: run start and length per cell, prefix over the table), then consecutive
// threads store to consecutive addresses -- four coalesced streams instead of four scattered 8-byte stores
// per particle.  Launched with the grid of k_cellid (same particles per block iteration).
// (NARROW / THIN are template parameters: a run-time select between the float32 and the float64 load made the compiler issue both,
// and the pass 0.8 ms slower on 1e8 particles)
template <bool NARROW, bool THIN>
__global__ void k_scatter_pos(const unsigned *cell, const unsigned *rank, long long n, const unsigned *offset,
                              const double *pos, double *x, double *y, double *z,
                              Pay *pay, const void *vel_v, const void *mass_v, int value_bytes, const int *g32, const long long *g64,
                              const long long *fof, int nu, ThinBuild thin) {
    // Velocities / Masses arrive as float64 or as float32 (what SWIFT snapshots hold, tests/dummy_halo_generator.py:806-811):
    // widened here, on the way into the payload record; every sum downstream is fp64
    const double *vel = (const double *)vel_v, *mass = (const double *)mass_v;
    const float *vel4 = (const float *)vel_v, *mass4 = (const float *)mass_v;
    constexpr bool narrow = NARROW;
    (void)value_bytes;
    __shared__ unsigned s_key[CELL_SLOTS], s_cnt[CELL_SLOTS], s_min[CELL_SLOTS], s_off[CELL_SLOTS];
    __shared__ unsigned s_wsum[BLK / 64], s_total;
    __shared__ double s_x[BLK], s_y[BLK], s_z[BLK];
    __shared__ unsigned s_i[BLK], s_p[BLK], s_c[BLK];
    __shared__ unsigned s_thin_n[2];   // this block's thinned particles so far, per level (its own sub-list: no global cursor)
    __shared__ double2 s_pay[BLK * 5];   // the block's payload records, written out as consecutive 16-byte words
    static_assert(sizeof(Pay) == 80, "payload record is five 16-byte words");
    for (int t = threadIdx.x; t < CELL_SLOTS; t += BLK) { s_key[t] = CELL_EMPTY; s_cnt[t] = 0u; s_min[t] = 0xffffffffu; }
    if (threadIdx.x < 2) s_thin_n[threadIdx.x] = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long stride = (long long)gridDim.x * BLK;
    const long long niter = (n + stride - 1) / stride;
    for (long long it = 0; it < niter; it++) {
        const long long i = it * stride + blockIdx.x * (long long)BLK + threadIdx.x;
        const bool active = i < n;
        unsigned c = 0, p = 0, slot = 0;
        double px = 0.0, py = 0.0, pz = 0.0;
        if (active) {
            c = cell[i];
            px = pos[3 * i]; py = pos[3 * i + 1]; pz = pos[3 * i + 2];
            p = offset[c] + rank[i];
            unsigned hsh = (c * 2654435761u) >> 23;
            while (true) {
                const unsigned prev = atomicCAS(&s_key[hsh], CELL_EMPTY, c);
                if (prev == CELL_EMPTY || prev == c) break;
                hsh = (hsh + 1u) & (CELL_SLOTS - 1);
            }
            slot = hsh;
            atomicMin(&s_min[slot], p);
            atomicAdd(&s_cnt[slot], 1u);
        }
        __syncthreads();
        // exclusive prefix of the run lengths over the table (two slots per thread)
        {
            const unsigned a0 = s_cnt[2 * threadIdx.x], a1 = s_cnt[2 * threadIdx.x + 1];
            unsigned v = a0 + a1;
            for (int o = 1; o < 64; o <<= 1) { const unsigned u = (unsigned)__shfl_up((int)v, o, 64); if (lane >= o) v += u; }
            if (lane == 63) s_wsum[wv] = v;
            __syncthreads();
            unsigned wbase = 0;
            for (int w = 0; w < wv; w++) wbase += s_wsum[w];
            const unsigned excl = wbase + v - (a0 + a1);
            s_off[2 * threadIdx.x] = excl; s_off[2 * threadIdx.x + 1] = excl + a0;
            if (threadIdx.x == BLK - 1) s_total = excl + a0 + a1;
        }
        __syncthreads();
        if (active) {
            unsigned li = s_off[slot] + (p - s_min[slot]);
            li = li < BLK ? li : BLK - 1;   // cannot happen with the launch contract above
            s_x[li] = px; s_y[li] = py; s_z[li] = pz; s_i[li] = (unsigned)i; s_p[li] = p; s_c[li] = c;
        }
        __syncthreads();
        if (threadIdx.x < s_total) {
            const unsigned q = s_p[threadIdx.x];
            const unsigned io = s_i[threadIdx.x];
            x[q] = s_x[threadIdx.x]; y[q] = s_y[threadIdx.x]; z[q] = s_z[threadIdx.x];
            if constexpr (THIN) {
                // the thinned samples k_predict runs the radius loop on: appended to a list per level (with their cell), put in
                // cell order by k_thin_place
                const unsigned hsh = thin_hash(s_x[threadIdx.x], s_y[threadIdx.x], s_z[threadIdx.x]);
#pragma unroll
                for (int lv = 0; lv < 2; lv++) {
                    if ((hsh & (lv == 0 ? 63u : 1023u)) != 0u) continue;
                    const unsigned local = atomicAdd(&s_thin_n[lv], 1u);   // (LDS: the sub-list belongs to this block)
                    if (local < thin.cap[lv]) {
                        const unsigned slot = blockIdx.x * thin.cap[lv] + local;
                        const unsigned cc = s_c[threadIdx.x];
                        thin.lc[lv][slot] = cc;
                        thin.lx[lv][slot] = s_x[threadIdx.x]; thin.ly[lv][slot] = s_y[threadIdx.x]; thin.lz[lv][slot] = s_z[threadIdx.x];
                        atomicAdd(&thin.count[lv][cc], 1u);
                    }
                }
            }
            // the payload record of the particle (read from the block's own window of the caller's arrays)
            Pay r;
            r.x = s_x[threadIdx.x]; r.y = s_y[threadIdx.x]; r.z = s_z[threadIdx.x];
            if constexpr (narrow) r.m = (double)mass4[io]; else r.m = mass[io];
            if (nu) { if constexpr (narrow) r.vx = (double)vel4[io]; else r.vx = vel[io]; r.vy = 0.0; r.vz = 0.0; r.grnr = -1; r.fof = -1; }
            else {
                const long long v0 = 3 * (long long)io;
                if constexpr (narrow) { r.vx = (double)vel4[v0]; r.vy = (double)vel4[v0 + 1]; r.vz = (double)vel4[v0 + 2]; }
                else { r.vx = vel[v0]; r.vy = vel[v0 + 1]; r.vz = vel[v0 + 2]; }
                r.grnr = g64 ? g64[io] : (long long)g32[io];
                r.fof = fof[io];
            }
            r.orig = (long long)io;   // (instead of an index array of its own: one scattered 4-byte store less per particle)
            const double2 *w = (const double2 *)&r;
#pragma unroll
            for (int k = 0; k < 5; k++) s_pay[5 * threadIdx.x + k] = w[k];
        }
        __syncthreads();
        // records of one destination run are consecutive in LDS and in memory: word c of the block goes to word
        // (c % 5) of record s_p[c / 5]; consecutive threads store consecutive 16 bytes
        {
            double2 *out = (double2 *)pay;
            for (unsigned c = threadIdx.x; c < 5u * s_total; c += BLK) {
                const unsigned t = c / 5u, k = c - 5u * t;
                out[5ull * s_p[t] + k] = s_pay[c];
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < CELL_SLOTS; t += BLK) { s_key[t] = CELL_EMPTY; s_cnt[t] = 0u; s_min[t] = 0xffffffffu; }
        __syncthreads();
    }
    if (THIN && threadIdx.x < 2) thin.cursor[2 * blockIdx.x + threadIdx.x] = s_thin_n[threadIdx.x] < thin.cap[threadIdx.x] ? s_thin_n[threadIdx.x] : thin.cap[threadIdx.x];
}

// ---- the per-halo kernels live in halo_class.hip, one translation unit per thread-group class and variant
#define G3T 1024
extern "C" {
#define SOAP_DECL_LAUNCH(name) void name(int groups, hipStream_t stream, const RunParams *params, SlabDev slab, unsigned *queue, const int *todo, long long n_todo)
SOAP_DECL_LAUNCH(soap_launch_halo_64n); SOAP_DECL_LAUNCH(soap_launch_halo_64i);
SOAP_DECL_LAUNCH(soap_launch_halo_64n32); SOAP_DECL_LAUNCH(soap_launch_halo_64i32);   // LDS tables for 32 calculations: eight groups per CU
SOAP_DECL_LAUNCH(soap_launch_halo_256n); SOAP_DECL_LAUNCH(soap_launch_halo_256i);
SOAP_DECL_LAUNCH(soap_launch_halo_1024n); SOAP_DECL_LAUNCH(soap_launch_halo_1024i);
// pipeline stages (GMODE 1: build, GMODE 2: property passes) and the radius loop with lanes = halos (stage_a.hip)
SOAP_DECL_LAUNCH(soap_launch_halo_64b); SOAP_DECL_LAUNCH(soap_launch_halo_256b); SOAP_DECL_LAUNCH(soap_launch_halo_1024b);
SOAP_DECL_LAUNCH(soap_launch_halo_64s); SOAP_DECL_LAUNCH(soap_launch_halo_256s); SOAP_DECL_LAUNCH(soap_launch_halo_1024s);
// the small-halo class (small_halo.hip): one wavefront per halo, slab in LDS; variants by slab entries and chunk content
void soap_launch_small_128d(int groups, hipStream_t stream, const RunParams *params, unsigned *queue, const int *todo, long long n_todo);
int soap_small_global_128d();
void soap_launch_small_512d(int groups, hipStream_t stream, const RunParams *params, unsigned *queue, const int *todo, long long n_todo);
void soap_launch_small_128i(int groups, hipStream_t stream, const RunParams *params, unsigned *queue, const int *todo, long long n_todo);
void soap_launch_small_512i(int groups, hipStream_t stream, const RunParams *params, unsigned *queue, const int *todo, long long n_todo);
// the variants for chunks with gas / stars / black holes (per-type moments, five half-mass classes, rotation, projected apertures)
void soap_launch_small_128h(int groups, hipStream_t stream, const RunParams *params, unsigned *queue, const int *todo, long long n_todo);
void soap_launch_small_512h(int groups, hipStream_t stream, const RunParams *params, unsigned *queue, const int *todo, long long n_todo);
void soap_launch_stage_a(hipStream_t stream, const RunParams *params, SlabDev arena, const int *todo, long long n_todo);
void soap_launch_gbuild(hipStream_t stream, const RunParams *params, SlabDev arena, long long nlarge, long long ntiles, long long ngblocks,
                        long long nselparts, const int *rows, const int *first_tile, const int *ntile, const int *first_gblock,
                        const int *ngblock, const int *first_sel, const int *nsel, const int *bin_first, const int *nbin);
void soap_launch_bpartials(hipStream_t stream, const RunParams *params, SlabDev arena, long long ntiles, double *bpart);
#ifdef SOAP_EXPERIMENTS
void soap_launch_small_1024d(int groups, hipStream_t stream, const RunParams *params, unsigned *queue, const int *todo, long long n_todo);
int soap_team_max_groups();
void soap_launch_team(hipStream_t stream, const RunParams *params, SlabDev arena, TeamCtl *ctl, long long nlarge, int groups, int split,
                      long long ntiles, long long ngblocks, long long nselparts, const int *rows, const int *first_tile, const int *ntile, const int *first_gblock, const int *ngblock,
                      const int *first_sel, const int *nsel, const int *bin_first, const int *nbin, double *bpart, double *bscan);
#endif

}

// team run: one word per large halo for the host (the state its team left it in)
__global__ void k_team_states(const HaloHdr *hdr, const int *rows, long long n, int *state) {
    for (long long i = blockIdx.x * (long long)BLK + threadIdx.x; i < n; i += (long long)gridDim.x * BLK) state[i] = hdr[rows[i]].pipe_state;
}

// pipeline bookkeeping kernels: the records start out empty except for the arena slice; the host only ever reads
// back one word per halo
__global__ void k_init_hdr(HaloHdr *hdr, long long n, const long long *off, const int *cap, const double *rbuild) {
    for (long long i = blockIdx.x * (long long)BLK + threadIdx.x; i < n; i += (long long)gridDim.x * BLK) {
        if (cap[i] == 0) continue;   // not a row of this iteration (its property passes may still be reading the record)
        HaloHdr z;
        memset(&z, 0, sizeof(z));
        z.off = off[i]; z.cap = cap[i]; z.pipe_state = PIPE_PENDING; z.R_build = rbuild[i];
        hdr[i] = z;
    }
}
// one word per halo for the host, and for the halos whose radius loop ran past the superset the radius to build the next
// one out to: a round and a bit (x 1.25) beyond what the loop asked for -- three times as far once that was not enough --
// as far as the region that is resident allows
__global__ void k_collect_state(const RunParams *Pp, HaloHdr *hdr, long long n, int *state, double *rbuild) {
    const RunParams &P = *Pp;
    for (long long i = blockIdx.x * (long long)BLK + threadIdx.x; i < n; i += (long long)gridDim.x * BLK) {
        const int s = hdr[i].pipe_state;
        state[i] = s;
        if (s == PIPE_NEED_SUPERSET) {
            const double limit = fmax(P.halo.read_radius[i], P.halo.resident ? P.halo.resident[i] : 0.0);
            rbuild[i] = fmin(limit, (hdr[i].R_build > 0.0 ? 3.0 : P.rebuild_factor) * hdr[i].R_need);   // a second miss: a halo inside a larger one's envelope, reach far
        }
        hdr[i].pipe_state = (s == PIPE_BUILT) ? PIPE_PENDING : s;   // never looked at again unless it is rebuilt
    }
}

// Cheap work estimate per halo: particles in the mesh cells overlapped by the first sorted superset
// (an upper bound of what the halo kernel will have to sort).  Used to pick the thread-group class of
// each halo and to start the heaviest halos first: bound-particle counts alone miss small halos that sit
// next to a giant (their search sphere swallows the giant's cells).
// one wavefront per halo: the lanes share the (k, j) mesh rows of the sphere's bounding cells, so the
// dependent cell_offset lookups of one halo are in flight together
__global__ void k_estimate(const RunParams *Pp, long long *est, const double *rbuild) {
    const RunParams &P = *Pp;
    const int lane = threadIdx.x & 63;
    long long h = blockIdx.x * (long long)(BLK / 64) + (threadIdx.x >> 6);
    if (h >= P.halo.n) return;
    double R = fmin(P.halo.read_radius[h], P.halo.search_radius[h] * P.first_factor);
    if (rbuild && rbuild[h] > 0.0) R = rbuild[h];   // pipeline: the halo is built again out to this radius
    long long tot = 0;
    for (int ts = 0; ts < NT_ALL; ts++) {
        const TypeDev &T = P.t[ts];
        if (T.n == 0) continue;
        Mask256 mk[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            double c = P.halo.cofp[3 * h + d], L = P.boxsize;
            int min_copy = 0, max_copy = 0, guard = 0;
            while (c + (double)(min_copy - 1) * L + R >= T.pos_min[d] && guard++ < 64) min_copy--;
            guard = 0;
            while (c + (double)(max_copy + 1) * L - R <= T.pos_max[d] && guard++ < 64) max_copy++;
            for (int cp = min_copy; cp <= max_copy; cp++) {
                double lo = fmax(T.pos_min[d], c + (double)cp * L - R);
                double hi = fmin(T.pos_max[d], c + (double)cp * L + R);
                double qlo = floor((lo - T.pos_min[d]) / T.cell_size[d]);
                double qhi = floor((hi - T.pos_min[d]) / T.cell_size[d]);
                int ilo = qlo < 0.0 ? 0 : (qlo > 1e9 ? 1000000000 : (int)qlo);
                int ihi = qhi < -1.0 ? -1 : (qhi > (double)(T.res - 1) ? T.res - 1 : (int)qhi);
                if (ilo <= ihi) mk[d].set_range(ilo, ihi);
            }
        }
        const int res = T.res;
        // runs of set bits along x (a sphere gives at most a few: one per periodic copy)
        int run_lo[4], run_hi[4], nrun = 0;
        for (int i0 = 0; i0 < res && nrun < 4;) {
            if (!mk[0].get(i0)) { i0++; continue; }
            int i1 = i0;
            while (i1 + 1 < res && mk[0].get(i1 + 1)) i1++;
            run_lo[nrun] = i0; run_hi[nrun] = i1; nrun++;
            i0 = i1 + 1;
        }
        const int nk = mk[2].count(), nj = mk[1].count();
        for (int pair = lane; pair < nk * nj; pair += 64) {
            const int ak = pair / nj, aj = pair - ak * nj;
            const int k = mk[2].nth(ak), j = mk[1].nth(aj);   // ak-th set bit of the z mask, aj-th of the y mask
            const long long rowbase = (long long)res * j + (long long)res * res * k;
            for (int r = 0; r < nrun; r++)
                tot += (long long)(T.cell_offset[rowbase + run_hi[r] + 1] - T.cell_offset[rowbase + run_lo[r]]);
        }
    }
    // wave total
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
    if (lane == 0) est[h] = tot;
}

// ---- Where will the radius loop of a halo end?  (scheduling only: no result depends on it) -------------------------
// The loop of halo_tasks.py:74-187 grows the search sphere x1.2 per round until the enclosed density of particle types
// 0, 1, 4, 5 has dropped below the run's target density and every calculation has the radius it asks for (an SO beyond its
// own radius, a radius multiple beyond  multiple x parent, the flow-rate shells 1.05 R_SO, apertures their physical radius).
// For most halos that is a round or three beyond the catalogue's search radius and the default first superset
// (first_factor x search radius) holds it.  A small central inside the envelope of a massive neighbour is different: the
// density about it stays above the target until the sphere has grown to the neighbour's own overdensity radius -- tens of
// rounds, a sphere of 1e5 particles for a halo of 50 -- and the work estimate of its first sphere says nothing about
// that.  This kernel runs the loop on a THINNED copy of the particles: one particle in 64 (one in 1024 for spheres of
// millions of candidates), chosen by a hash of its coordinate bits -- a property of the particle, so the sample and with it
// every decision taken here is the same in every run, whatever order the placement pass's atomics left the particles of a
// cell in (the sample has its own cell-ordered copy of the coordinates, built with the mesh) -- weighted with the mean particle mass of its type,
// counted on the ladder R_k = search_radius x 1.2^k by one wave per halo.
// Out: the radius the first sorted superset should reach (0: the default) -- so that the sphere is selected and sorted once
// instead of shell by shell, the halo is sent to the thread-group class of its FINAL size, and the longest chains of a
// class start first.  A wrong guess costs time, never correctness: the halo kernels extend a superset that is too small
// and ignore what they do not need of one that is too large.
#define PRED_K 64          // ladder steps (1.2^63 ~ 1e5 search radii: never the limit)
#define PRED_MAXS 4096     // thinned particles per sphere and type at most (a sphere of 4e6 candidates is thinned further)
// the mesh rows (k, j, x-run) a sphere of radius R about halo h overlaps in the mesh of type T (shared_mesh.py:146-179, as k_estimate)
struct PredRows {
    Mask256 mk[3];
    int run_lo[4], run_hi[4], nrun, nk, nj, ntrip, res;
    __device__ __forceinline__ void build(const RunParams &P, const TypeDev &T, long long h, double R) {
        const double L = P.boxsize;
        mk[0] = Mask256(); mk[1] = Mask256(); mk[2] = Mask256();
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const double c = P.halo.cofp[3 * h + d];
            int min_copy = 0, max_copy = 0, guard = 0;
            while (c + (double)(min_copy - 1) * L + R >= T.pos_min[d] && guard++ < 64) min_copy--;
            guard = 0;
            while (c + (double)(max_copy + 1) * L - R <= T.pos_max[d] && guard++ < 64) max_copy++;
            for (int cp = min_copy; cp <= max_copy; cp++) {
                const double lo = fmax(T.pos_min[d], c + (double)cp * L - R), hi = fmin(T.pos_max[d], c + (double)cp * L + R);
                const double qlo = floor((lo - T.pos_min[d]) / T.cell_size[d]), qhi = floor((hi - T.pos_min[d]) / T.cell_size[d]);
                const int ilo = qlo < 0.0 ? 0 : (qlo > 1e9 ? 1000000000 : (int)qlo);
                const int ihi = qhi < -1.0 ? -1 : (qhi > (double)(T.res - 1) ? T.res - 1 : (int)qhi);
                if (ilo <= ihi) mk[d].set_range(ilo, ihi);
            }
        }
        res = T.res;
        nrun = 0;
        for (int i0 = 0; i0 < res && nrun < 4;) {
            if (!mk[0].get(i0)) { i0++; continue; }
            int i1 = i0;
            while (i1 + 1 < res && mk[0].get(i1 + 1)) i1++;
            run_lo[nrun] = i0; run_hi[nrun] = i1; nrun++;
            i0 = i1 + 1;
        }
        nk = mk[2].count(); nj = mk[1].count();
        ntrip = nk * nj * nrun;
    }
    // first and one-past-last cell-ordered particle of row number tl (`off`: the cells' offsets -- of all particles, or of a thinned copy)
    __device__ __forceinline__ void row(const TypeDev &T, int tl, unsigned &beg, unsigned &end) const { row_of(T.cell_offset, tl, beg, end); }
    __device__ __forceinline__ void row_of(const unsigned *off, int tl, unsigned &beg, unsigned &end) const {
        const int rr = tl % nrun, pj = (tl / nrun) % nj, pk = tl / (nrun * nj);
        const int k = mk[2].nth(pk), j = mk[1].nth(pj);
        const long long rowbase = (long long)res * j + (long long)res * res * k;
        int rl = run_lo[0], rh = run_hi[0];
        if (rr == 1) { rl = run_lo[1]; rh = run_hi[1]; } else if (rr == 2) { rl = run_lo[2]; rh = run_hi[2]; } else if (rr == 3) { rl = run_lo[3]; rh = run_hi[3]; }
        beg = off[rowbase + rl];
        end = off[rowbase + rh + 1];
    }
};
// particles in the mesh cells a sphere of radius R about halo h overlaps, all staged types (what k_estimate counts); wave total
__device__ __forceinline__ long long pred_candidates(const RunParams &P, long long h, double R, int lane) {
    long long tot = 0;
    for (int ts = 0; ts < NT_ALL; ts++) {
        const TypeDev &T = P.t[ts];
        if (T.n == 0) continue;
        PredRows rows;
        rows.build(P, T, h, R);
        for (int tl = lane; tl < rows.ntrip; tl += 64) { unsigned beg, end; rows.row(T, tl, beg, end); tot += (long long)(end - beg); }
    }
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
    return tot;
}
__global__ void __launch_bounds__(BLK) k_predict(const RunParams *Pp, double *r_first, long long *est, int min_samples, double margin, int force, long long always_above) {
    __shared__ int s_ibin[BLK / 64][PRED_K];
    __shared__ double s_dens[BLK / 64][PRED_K];
    const RunParams &P = *Pp;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long h = blockIdx.x * (long long)(BLK / 64) + wv;
    if (h >= P.halo.n) return;
    int *ibin = s_ibin[wv];
    double *dens = s_dens[wv];
    const double Rs = P.halo.search_radius[h], Rread = P.halo.read_radius[h];
    const double cap = fmin(0.5 * P.boxsize, fmax(Rread, P.halo.resident ? P.halo.resident[h] : 0.0));
    const double R0 = fmin(Rread, Rs * P.first_factor);
    double out = 0.0;
    // Only a halo whose first sphere holds many times its own bound particles can be inside another's envelope: everybody else
    // keeps the estimate k_estimate left (this kernel needs three times its registers: it runs for the suspects only) -- and every
    // halo of the thread-group classes (more than `always_above` candidates): there a first sphere that turns out too small is a chain of
    // rebuilds on ONE workgroup, the longest thing in the pass (uncapped C3 box: one central of 1.1e5 candidates whose loop ends beyond
    // 5e5 entries kept the 256-thread class busy for 23 ms and was then done again in a pass of its own)
    const long long cand0 = est[h];
    const bool suspect = force || cand0 > 8ll * (long long)P.halo.nr_bound[h] + 2048ll || cand0 > always_above;
    const bool central = P.halo.is_central[h] == 1 && P.target_density > 0.0;
    if (!suspect || !central) { if (lane == 0) r_first[h] = 0.0; return; }
    if (cap > R0 * 1.0001) {
        const double cx = P.halo.cofp[3 * h], cy = P.halo.cofp[3 * h + 1], cz = P.halo.cofp[3 * h + 2], L = P.boxsize;
        const double inv_log = 1.0 / log2(1.2);
        double binv = 0.0;    // lane k: mass of the thinned particles (times the thinning factor) on ladder step k, all types
        double bcnt = 0.0;    // ... and how many thinned particles that is
        // the halo's own bound particles lie inside the catalogue's search radius, i.e. inside every ladder step, whether the thinning
        // kept any of them or not (a central of 40 particles has none in a 1-in-64 sample: its ladder would start at zero density)
        double own_mass = 0.0;
        for (int ts = 0; ts < SOAP_NTYPES; ts++) if (P.t[ts].n > 0) own_mass = fmax(own_mass, P.t[ts].mean_mass);
        own_mass *= (double)P.halo.nr_bound[h];
        double Rlo = -1.0, Rhi = R0, need = 0.0;
        double wmax = 0.0;    // mass one thinned particle stands for (the largest over the types and thinning levels met)
        int nsamp = 0;
        for (int iter = 0; iter < 8; iter++) {
            // ---- thinned particles of the shell Rlo < r <= Rhi onto the ladder: the mesh rows of the sphere one after the other
            // (their cell_offset lookups 64 at a time, lanes = rows), lanes = 32-position words of the row's sample bitmap
            const double R2lo = Rlo < 0.0 ? -1.0 : Rlo * Rlo, R2hi = Rhi * Rhi;
            long long cand = 0;
            for (int ts = 0; ts < NT_ALL; ts++) {
                const TypeDev &T = P.t[ts];
                if (T.n == 0) continue;
                PredRows rows;
                rows.build(P, T, h, Rhi);
                // candidates of this type (the work estimate of the first sphere comes for free), and how far to thin them
                long long tot = 0;
                for (int tl = lane; tl < rows.ntrip; tl += 64) { unsigned beg, end; rows.row(T, tl, beg, end); tot += (long long)(end - beg); }
                for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
                cand += tot;
                if (ts >= SOAP_NTYPES) continue;   // (neutrinos: counted as candidates, no part of the density test)
                const bool coarse = tot > 64ll * PRED_MAXS;
                const int lv = coarse ? 1 : 0;
                const unsigned *toff = T.thin_off[lv];
                const double *tx = T.thin_x[lv], *ty = T.thin_y[lv], *tz = T.thin_z[lv];
                ibin[lane] = 0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                for (int t0 = 0; t0 < rows.ntrip; t0 += 64) {
                    unsigned mybeg = 0, myend = 0;
                    if (t0 + lane < rows.ntrip) rows.row_of(toff, t0 + lane, mybeg, myend);
                    const int cnt = rows.ntrip - t0 < 64 ? rows.ntrip - t0 : 64;
                    // four rows per trip: their thinned particles (lanes = particles) are requested together
                    for (int q0 = 0; q0 < cnt; q0 += 4) {
                        unsigned beg[4], end[4], longest = 0u;
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int q = q0 + j < cnt ? q0 + j : q0;
                            beg[j] = (unsigned)__builtin_amdgcn_readlane((int)mybeg, q);
                            end[j] = q0 + j < cnt ? (unsigned)__builtin_amdgcn_readlane((int)myend, q) : beg[j];
                            longest = end[j] - beg[j] > longest ? end[j] - beg[j] : longest;
                        }
                        for (unsigned o = 0; o < longest; o += 64u) {
                            double px[4], py[4], pz[4];
                            bool have[4];
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const unsigned p = beg[j] + o + (unsigned)lane;
                                have[j] = p < end[j];
                                px[j] = py[j] = pz[j] = 0.0;
                                if (have[j]) { px[j] = tx[p]; py[j] = ty[p]; pz[j] = tz[p]; }
                            }
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                if (!have[j]) continue;
                                const double ax = min_image(px[j], cx, L), ay = min_image(py[j], cy, L), az = min_image(pz[j], cz, L);
                                const double r2 = (ax * ax + ay * ay) + az * az;
                                if (r2 <= R2hi && r2 > R2lo) {
                                    // smallest k with r <= Rs 1.2^k
                                    int kb = r2 > Rs * Rs ? (int)ceil(0.5 * log2(r2 / (Rs * Rs)) * inv_log) : 0;
                                    kb = kb < 0 ? 0 : (kb > PRED_K - 1 ? PRED_K - 1 : kb);
                                    atomicAdd(&ibin[kb], 1);
                                    nsamp++;
                                }
                            }
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                // (integer counts per type, added in type order: the same sum whatever order the particles came in)
                binv += (double)ibin[lane] * (T.mean_mass * (coarse ? 1024.0 : 64.0));
                bcnt += (double)ibin[lane];
                wmax = fmax(wmax, T.mean_mass * (coarse ? 1024.0 : 64.0));
            }
            (void)cand;
            // ---- the loop on the ladder: lane k holds step k
            const double Rk = Rs * exp2((double)lane * log2(1.2));
            double M = binv, C = bcnt;
            for (int o = 1; o < 64; o <<= 1) { const double v = __shfl_up(M, o, 64), w = __shfl_up(C, o, 64); if (lane >= o) { M += v; C += w; } }
            const bool known = Rk <= Rhi * 1.000001;      // steps the thinned sphere covers
            // The enclosed mass of a step is known to about two standard deviations of its sample count; the ladder takes the UPPER end
            // and adds the halo's own particles (a density that is too high ends the guessed loop later: a larger first sphere than
            // needed costs a little, a smaller one a rebuild -- a handful of samples at the inner steps used to read as "every threshold
            // crossed at the first step", and 5 x R500 of such a crossing as the end of the loop)
            // (... of a Poisson count: C + 2 sqrt(C) + 2 samples -- NO sample inside a step says "fewer than about two", not "none": a central of
            // 56 particles in the envelope of a cluster, no thinned particle within 1.7 search radii, read as "below the target at the
            // fourth step", its loop ends at 27 search radii and the giants' pipeline built it three times)
            M = own_mass + M + (2.0 * sqrt(C) + 2.0) * wmax;
            dens[lane] = known ? M / (4.0 / 3.0 * M_PI * Rk * Rk * Rk) : 1e300;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (iter == 0) {
                // the enclosed density at the end of the default first superset: not above the target (any halo that is not
                // inside a larger one's envelope) means the loop ends inside it -- nothing to guess
                int klast = 0;
                for (int k = 0; k < PRED_K; k++) if (dens[k] < 1e299) klast = k;
                if (dens[klast] <= P.target_density && !force) break;   // (force: test hook, every halo gets a guess)
            }
            // lanes = calculations: the radius each one asks for (1e300: its density is not reached inside Rhi)
            double want = 0.0, Rso = 0.0;
            {
                const int c = lane;
                soap_calc_desc cd = P.calcs[c < P.ncalc ? c : 0];
                const bool live = c >= 1 && c < P.ncalc;
                double rho = 0.0;
                if (c == 0) rho = P.target_density;                   // (lane 0: the loop's own density test)
                else if (live && cd.kind == SOAP_CALC_SO_DENSITY) rho = cd.reference_density;
                if (rho > 0.0) {
                    Rso = 1e300;
                    for (int k = 0; k < PRED_K; k++) {
                        if (dens[k] <= rho) { Rso = Rs * exp2((double)k * log2(1.2)); break; }
                        if (dens[k] > 1e299) break;
                    }
                    want = Rso;
                    if (c >= 1 && (cd.flags & SOAP_CALC_FLAG_DM_FLOW) && Rso < 1e299) want = 1.05 * Rso;
                }
                const int par = (live && cd.kind == SOAP_CALC_SO_MULTIPLE) ? cd.parent : 0;
                const double Rpar = __shfl(Rso, par, 64);
                if (live && cd.kind == SOAP_CALC_SO_MULTIPLE) want = Rpar < 1e299 ? cd.radius * Rpar : 1e300;
                if (live && cd.kind != SOAP_CALC_SO_DENSITY && cd.kind != SOAP_CALC_SO_MULTIPLE) want = cd.physical_radius_mpc;
            }
#ifdef SOAP_DEBUG_PREDICT
            if (h == SOAP_DEBUG_PREDICT) {
                if (lane < 32) printf("iter %d lane %2d: Rk/Rs %8.3f dens/target %10.3g C %8.1f known %d | want/Rs %10.3g Rso/Rs %10.3g\n", iter, lane, Rk / Rs, dens[lane] / P.target_density, C, (int)known, want / Rs, Rso / Rs);
                if (lane == 0) printf("iter %d: Rlo/Rs %.3f Rhi/Rs %.3f cap/Rs %.3f own_mass %.4g nsamp(lane0) %d cand %lld nr_bound %lld\n", iter, Rlo / Rs, Rhi / Rs, cap / Rs, own_mass, nsamp, cand, (long long)P.halo.nr_bound[h]);
            }
#endif
            for (int o = 32; o > 0; o >>= 1) want = fmax(want, __shfl_xor(want, o, 64));
            need = want;
            if (need <= Rhi) break;                 // the thinned sphere holds the end of the loop
            if (Rhi >= cap) { need = cap; break; }  // as far as the resident region goes
            Rlo = Rhi;
            Rhi = fmin(cap, fmax(2.0 * Rhi, need < 1e299 ? 1.1 * need : 0.0));
        }
        int ns = nsamp;
        for (int o = 32; o > 0; o >>= 1) ns += __shfl_xor(ns, o, 64);
        // a guess from a handful of thinned particles is noise: such a halo keeps the default (it is small, whatever it does)
        if (ns >= min_samples && need * margin > R0) out = fmin(cap, need * margin);
    }
    // the work estimate: candidates of the first sorted superset
    const long long e = out > 0.0 ? pred_candidates(P, h, out, lane) : cand0;
    if (lane == 0) { r_first[h] = out; est[h] = e; }
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
            in = (T.pay[p].grnr == hindex);   // (the payload record: the caller's arrays are not kept after staging)
        }
        if (in) {
            unsigned s = atomicAdd(count, 1u);
            if ((long long)s < cap) out[s] = T.pay[p].orig;
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// Size-keyed cache of device allocations: hipMalloc/hipFree of multi-GB buffers cost milliseconds
// each and would dominate a staging call that is repeated for every chunk.
// Chunks differ in size, so blocks of a previous chunk rarely fit the next one exactly: a request takes the
// smallest cached block that is large enough but at most 25 % larger (best fit), the cache is trimmed to
// `max_cached` bytes whenever a block comes back (largest first: those are the ones a differently sized chunk
// cannot reuse), and a failed hipMalloc empties the cache and tries once more.
struct DevPool {
    std::multimap<size_t, void *> free_blocks;   // capacity -> block
    std::map<void *, size_t> capacity;           // blocks handed out: their real size
    size_t cached = 0, max_cached = (size_t)48 << 30;   // one generation of a 1e8-particle chunk is ~15 GB
    hipError_t alloc(void **p, size_t bytes) {
        if (bytes == 0) bytes = 8;
        auto it = free_blocks.lower_bound(bytes);
        if (it != free_blocks.end() && it->first <= bytes + bytes / 4) {
            *p = it->second; capacity[*p] = it->first; cached -= it->first; free_blocks.erase(it);
            return hipSuccess;
        }
        hipError_t e = hipMalloc(p, bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            purge();
            e = hipMalloc(p, bytes);
        }
        if (e == hipSuccess) capacity[*p] = bytes;
        return e;
    }
    void release(void *p, size_t bytes) {
        if (!p) return;
        auto it = capacity.find(p);
        const size_t cap = it != capacity.end() ? it->second : (bytes == 0 ? 8 : bytes);
        if (it != capacity.end()) capacity.erase(it);
        free_blocks.emplace(cap, p);
        cached += cap;
        while (cached > max_cached && !free_blocks.empty()) {
            auto big = std::prev(free_blocks.end());
            (void)hipFree(big->second);
            cached -= big->first;
            free_blocks.erase(big);
        }
    }
    void purge() { for (auto &kv : free_blocks) (void)hipFree(kv.second); free_blocks.clear(); cached = 0; }
};

struct TypeHost {
    long long n = 0;
    int res = 0;
    int grnr64 = 0;
    double *x = nullptr, *y = nullptr, *z = nullptr;
    // velocities / masses / ids: the caller's device arrays, or the engine's device copies of host arrays
    const double *vel = nullptr, *mass = nullptr;
    const int *g32 = nullptr;
    const long long *g64 = nullptr, *fof = nullptr;
    std::vector<std::pair<void *, size_t>> owned;   // device copies the engine made (host inputs)
    unsigned *orig = nullptr, *cell_offset = nullptr;
    Pay *pay = nullptr;
    double pos_min[3], pos_max[3], cell_size[3];
    double mean_mass = 0.0;
    int vbytes = 8;    // width of the Masses / Velocities this type was staged with (8 or 4)
    unsigned *thin_off[2] = {nullptr, nullptr};
    double *thin_xyz[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    size_t thin_cap[2] = {0, 0};
    void release(DevPool &pool) {
        for (int lv = 0; lv < 2; lv++) {
            pool.release(thin_off[lv], ((size_t)res * res * res + 1) * 4);
            for (int d = 0; d < 3; d++) pool.release(thin_xyz[lv][d], thin_cap[lv] * 8);
        }
        pool.release(pay, (size_t)n * sizeof(Pay));
        void *d8[] = {x, y, z};
        for (void *p : d8) pool.release(p, (size_t)n * 8);
        for (auto &o : owned) pool.release(o.first, o.second);
        pool.release(orig, (size_t)n * 4);
        pool.release(cell_offset, ((size_t)res * res * res + 1) * 4);
        *this = TypeHost();
    }
};

#define NSV 3   // variants of the small-halo kernels
#define SMALL_MAXG512 (256 * 11)    // most persistent waves of the 512- / 1024-entry variants (slices of the streamed-column buffer)
#define SMALL_MAXG1024 (256 * 8)
#define SMALL_MAXG128 (256 * 24)
#define SMALL_MAXPJ 8    // projected apertures the hydro variants of the small-halo kernels hold (SMAXPJ of small_halo.hip)
struct soap_engine {
    soap_engine_config cfg;
    hipStream_t stream = nullptr;
    bool own_streams = false;   // one priority stream per class
    DevPool pool;
    TypeHost t[NT_ALL];
    // halos
    long long n_halo = 0;
    double *d_cofp = nullptr, *d_search = nullptr, *d_read = nullptr;
    long long *d_index = nullptr, *d_nrbound = nullptr, *d_stats = nullptr;
    int *d_central = nullptr, *d_status = nullptr, *d_todo = nullptr;
    double *d_resident = nullptr;
    double *d_first = nullptr;    // radius the first superset of every halo should reach (k_predict), n_halo doubles
    size_t first_count = 0;
    std::vector<double> h_first;
    long long n_predicted = 0;    // halos whose guess lies beyond the default first superset
    // member probe
    long long probe_row = -1, probe_cap = 0;
    int probe_calc = 0;
    unsigned long long *d_probe = nullptr;
    unsigned *d_probe_count = nullptr;
    // run
    long long ncols = 0;
    double *d_table = nullptr;
    soap_calc_desc *d_calcs = nullptr;
    int *d_coloff = nullptr;
    size_t table_bytes = 0, calcs_bytes = 0, coloff_bytes = 0;
    double ms_host_pre = 0, ms_host_setup = 0, ms_host_classify = 0, ms_host_total = 0;   // wall clock inside run()
    unsigned *d_queue = nullptr;
    RunParams *d_params = nullptr;
    RunParams *d_params_small = nullptr;  // the small-halo kernels' own copy (hand-over list, streamed-column buffers): uploaded after the large classes, which read d_params, have been launched
    RunParams *d_params_pipe = nullptr;   // the pipeline's own copy: its iterations rewrite hdr / tile tables while class and small-halo kernels of the same run read d_params
    unsigned long long *d_phase = nullptr;
    unsigned long long phase[32] = {};
    // per-class slabs/queues: class 0 = one wavefront per halo, 1 = 256 threads, 2 = 1024 threads
    struct Class {
        long long cap = 0;
        int groups = 0;
        SlabDev slab{};
        unsigned *d_queue = nullptr;
        int *d_todo = nullptr;
        long long todo_cap = 0;
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr, start = nullptr;
        hipEvent_t pstart[2] = {nullptr, nullptr}, pdone[2] = {nullptr, nullptr};   // pipeline stages 1 and 3
        double pms[2] = {0.0, 0.0};
        double ms = 0.0;
        long long nh = 0;
    } cls[3];
    Class small[NSV];   // the small-halo class (no slab: its halos live in LDS), by slab entries: 128 (all of it in LDS), 512 and 1024 (streamed columns in global memory)
    double *d_small_cols = nullptr;   // ... the latter's columns: seven per persistent wave
    int *d_part_count = nullptr;      // large halos of chunks with gas / stars shared by two 1024-thread groups: groups that have finished, per row
    size_t part_count_n = 0;
    std::vector<int> split_rows;      // ... their task list (two entries per row)
    int *d_small_esc = nullptr;       // [0] count, then the rows the small-halo kernels hand on to the wavefront class
    size_t small_esc_count = 0;
    size_t small_cols_bytes = 0;
    std::vector<long long> h_nrbound;
    std::vector<unsigned long long> sort_ka, sort_kb;   // scratch of the classification sort (run)
    std::vector<unsigned> sort_rows, sort_e32;
    // pipeline path: arena of per-halo slab slices, one record per halo
    SlabDev arena{};
    long long arena_cap = 0;
    HaloHdr *d_hdr = nullptr;
    size_t hdr_bytes = 0;
    GBuild *d_gb = nullptr;        // scratch of the large halos' tile-parallel build
    size_t gb_count = 0;
    int *d_gints = nullptr;        // rows / tile tables of the large halos
    std::vector<int> gints_host;   // ... packed for one upload
    size_t gints_count = 0;
    hipEvent_t ev_b[8][2] = {};    // giants' mode: property passes of every pipeline iteration
    int *d_gbins = nullptr;        // radial bins of the giants' build
    double *d_bpart[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // tile sums for the property stage, per pipeline iteration
    size_t bpart_count[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t gbins_count = 0;
    double ms_gbuild = 0;
    long long n_gbuild = 0;
    hipEvent_t ev_p[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double ms_build = 0, ms_stage_a = 0, ms_stage_b = 0;
    int n_passes = 0;
    long long n_pipeline = 0, n_single = 0, n_small_escalated = 0;
    // timing
    double ms_halo = -1, ms_mesh = -1, ms_reorder = -1;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev_in = nullptr, ev_zero = nullptr;
    // staging that returned with its last kernels still in flight (on_device = 2): completed -- synchronised, checked, timed, its
    // temporaries released -- by the next call that needs the mesh (finish_stage)
    bool stage_pending = false;
    int stage_pending_slot = -1;
    int stage_error = 0;           // a deferred staging call failed and a call that cannot report it (soap_engine_kernel_ms) completed it: the next call that can does
    std::string stage_error_msg;
    std::vector<std::pair<void *, size_t>> stage_scratch;
    // team run (k_team): its own stream (highest priority: launched first, resident from the start), control block, clocks
    hipStream_t team_stream = nullptr;
    hipEvent_t team_start = nullptr, team_done = nullptr;
    TeamCtl *d_teamctl = nullptr;
    double ms_team = 0.0;
    long long n_team = 0, n_team_failed = 0;
};

static void free_slab(soap_engine::Class &c) {
    void *ptrs[] = {c.slab.r, c.slab.m, c.slab.dx, c.slab.dy, c.slab.dz, c.slab.vx,
                    c.slab.vy, c.slab.vz, c.slab.cum, c.slab.q, c.slab.dens, c.slab.pr, c.slab.pidx, c.slab.pi, c.slab.flags};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    c.slab = SlabDev{};
    c.cap = 0;
    c.groups = 0;
}

static int alloc_slab(soap_engine::Class &c, long long cap, int groups) {
    if (c.cap == cap && c.groups >= groups) return 0;   // keep a larger allocation
    free_slab(c);
    size_t ne = (size_t)cap * (size_t)groups;
    double **dp[] = {&c.slab.r, &c.slab.m, &c.slab.dx, &c.slab.dy, &c.slab.dz,
                     &c.slab.vx, &c.slab.vy, &c.slab.vz, &c.slab.cum, &c.slab.q, &c.slab.dens, &c.slab.pr};
    for (double **p : dp) HIPCHECK(hipMalloc((void **)p, ne * sizeof(double)));
    HIPCHECK(hipMalloc((void **)&c.slab.pidx, ne * sizeof(unsigned)));
    HIPCHECK(hipMalloc((void **)&c.slab.pi, ne * sizeof(unsigned)));
    HIPCHECK(hipMalloc((void **)&c.slab.flags, ne));
    c.slab.cap = cap;
    c.cap = cap;
    c.groups = groups;
    return 0;
}

static void free_arena(soap_engine *h) {
    void *ptrs[] = {h->arena.r, h->arena.m, h->arena.dx, h->arena.dy, h->arena.dz, h->arena.vx, h->arena.vy, h->arena.vz,
                    h->arena.cum, h->arena.q, h->arena.dens, h->arena.pr, h->arena.pidx, h->arena.pi, h->arena.flags};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    h->arena = SlabDev{};
    h->arena_cap = 0;
}

static int alloc_arena(soap_engine *h, long long entries) {
    if (h->arena_cap >= entries) return 0;
    free_arena(h);
    const size_t ne = (size_t)(entries + entries / 8);   // head room: the next chunk is rarely exactly as large
    double **dp[] = {&h->arena.r, &h->arena.m, &h->arena.dx, &h->arena.dy, &h->arena.dz, &h->arena.vx, &h->arena.vy,
                     &h->arena.vz, &h->arena.cum, &h->arena.q, &h->arena.dens, &h->arena.pr};
    h->pool.purge();   // cached staging blocks make room for the arena
    for (double **p : dp) HIPCHECK(hipMalloc((void **)p, ne * sizeof(double)));
    HIPCHECK(hipMalloc((void **)&h->arena.pidx, ne * sizeof(unsigned)));
    HIPCHECK(hipMalloc((void **)&h->arena.pi, ne * sizeof(unsigned)));
    HIPCHECK(hipMalloc((void **)&h->arena.flags, ne));
    h->arena.cap = (long long)ne;
    h->arena_cap = (long long)ne;
    return 0;
}

static void free_slabs(soap_engine *h) {
    for (int k = 0; k < 3; k++) free_slab(h->cls[k]);
    free_arena(h);
}

static int32_t create_resources(soap_engine *h);

extern "C" {

int32_t soap_engine_abi_version(void) { return SOAP_ENGINE_ABI_VERSION; }
const char *soap_engine_last_error(void) { return g_err.c_str(); }

int32_t soap_engine_desc_columns(const soap_calc_desc *desc) {
    if (!desc) return -1;
    int32_t nc = soap_engine_calc_columns(desc->kind);
    // the inertia tensors are the last columns of the subhalo and SO blocks: only there when wanted
    if (nc > 0 && desc->kind <= SOAP_CALC_SO_MULTIPLE && !(desc->flags & SOAP_CALC_FLAG_INERTIA)) nc -= INERTIA_NCOLS;
    // projected apertures: the 2D tensors follow the three projection blocks
    if (desc->kind == SOAP_CALC_PROJECTED && (desc->flags & SOAP_CALC_FLAG_INERTIA)) nc += PJ_INERTIA_NCOLS;
    return nc;
}

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
    const int32_t rc = create_resources(h);
    if (rc != 0) { const std::string msg = g_err; (void)soap_engine_destroy(h); g_err = msg; return rc; }
    *out = h;
    return 0;
}

static int32_t create_resources(soap_engine *h) {
    HIPCHECK(hipStreamCreate(&h->stream));
    HIPCHECK(hipEventCreate(&h->ev0));
    HIPCHECK(hipEventCreate(&h->ev1));
    HIPCHECK(hipEventCreate(&h->ev2));
    HIPCHECK(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&h->ev_zero, hipEventDisableTiming));
    HIPCHECK(hipMalloc((void **)&h->d_queue, sizeof(unsigned)));
    // one stream per thread-group class, on distinct priority levels where the device offers three:
    // streams of equal priority may be mapped onto the same hardware queue (and then serialise), streams of
    // different priority are not.  Giants (class 2) first: they are the tail.
    int prio_least = 0, prio_greatest = 0;
    HIPCHECK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    h->own_streams = (prio_least - prio_greatest >= 2) && !expt_env("SOAP_ENGINE_TWO_STREAMS");
    for (int k = 0; k < 3; k++) {
        if (h->own_streams) {
            int prio = k == 0 ? prio_least : (k == 1 ? prio_least - 1 : prio_greatest);
            if (const char *e = expt_env("SOAP_ENGINE_PRIO")) {   // diagnostic: three digits, rank of class 0 / 1 / 2 (0 = least)
                const int r = e[k] - '0';
                prio = r == 0 ? prio_least : (r == 1 ? prio_least - 1 : prio_greatest);
            }
            HIPCHECK(hipStreamCreateWithPriority(&h->cls[k].stream, hipStreamNonBlocking, prio));
        } else {
            HIPCHECK(hipStreamCreate(&h->cls[k].stream));
        }
        HIPCHECK(hipEventCreate(&h->cls[k].done));
        HIPCHECK(hipEventCreate(&h->cls[k].start));
        for (int q = 0; q < 2; q++) { HIPCHECK(hipEventCreate(&h->cls[k].pstart[q])); HIPCHECK(hipEventCreate(&h->cls[k].pdone[q])); }
        HIPCHECK(hipMalloc((void **)&h->cls[k].d_queue, sizeof(unsigned)));
    }
    {
        // the small-halo class shares the priority level of the wavefront class
        for (int v = 0; v < NSV; v++) {
            if (h->own_streams) HIPCHECK(hipStreamCreateWithPriority(&h->small[v].stream, hipStreamNonBlocking, prio_least));
            else HIPCHECK(hipStreamCreate(&h->small[v].stream));
            HIPCHECK(hipEventCreate(&h->small[v].done));
            HIPCHECK(hipEventCreate(&h->small[v].start));
            HIPCHECK(hipMalloc((void **)&h->small[v].d_queue, sizeof(unsigned)));
        }
    }
    if (h->own_streams) HIPCHECK(hipStreamCreateWithPriority(&h->team_stream, hipStreamNonBlocking, prio_greatest));
    else HIPCHECK(hipStreamCreate(&h->team_stream));
    HIPCHECK(hipEventCreate(&h->team_start));
    HIPCHECK(hipEventCreate(&h->team_done));
    HIPCHECK(hipMalloc((void **)&h->d_teamctl, sizeof(TeamCtl)));
    HIPCHECK(hipMalloc((void **)&h->d_params, sizeof(RunParams)));
    HIPCHECK(hipMalloc((void **)&h->d_params_pipe, sizeof(RunParams)));
    HIPCHECK(hipMalloc((void **)&h->d_params_small, sizeof(RunParams)));
    HIPCHECK(hipMalloc((void **)&h->d_phase, 32 * sizeof(unsigned long long)));
    for (int k = 0; k < 6; k++) HIPCHECK(hipEventCreate(&h->ev_p[k]));
    return 0;
}

static void free_halos(soap_engine *h) {
    // back into the handle's pool: the next chunk's catalogue usually has the same size (hipMalloc / hipFree of
    // nine arrays cost ~0.8 ms per staging)
    const size_t n = (size_t)h->n_halo;
    std::pair<void *, size_t> ptrs[] = {{h->d_cofp, n * 24}, {h->d_search, n * 8}, {h->d_read, n * 8}, {h->d_index, n * 8},
                                        {h->d_nrbound, n * 8}, {h->d_stats, n * 8 * SOAP_STATS_COLUMNS}, {h->d_central, n * 4},
                                        {h->d_status, n * 4}, {h->d_todo, n * 4}, {h->d_resident, n * 8}};
    for (auto &p : ptrs) if (p.first) h->pool.release(p.first, p.second);
    h->d_cofp = h->d_search = h->d_read = nullptr;
    h->d_resident = nullptr;
    h->d_index = h->d_nrbound = h->d_stats = nullptr;
    h->d_central = h->d_status = h->d_todo = nullptr;
    h->n_halo = 0;
}

// completes a staging call that returned early (on_device = 2): what stage_type does at its end otherwise
static int32_t finish_stage(soap_engine *h) {
    if (h->stage_error) {   // (sticky until reported: the type's mesh was released, a run would silently go on without it)
        const int e = h->stage_error;
        h->stage_error = 0;
        return fail(e, h->stage_error_msg);
    }
    if (!h->stage_pending) return 0;
    h->stage_pending = false;
    const hipError_t se = hipStreamSynchronize(h->stream);
    const hipError_t le = hipGetLastError();
    for (auto &p : h->stage_scratch) h->pool.release(p.first, p.second);
    h->stage_scratch.clear();
    if (se != hipSuccess || le != hipSuccess) {
        if (h->stage_pending_slot >= 0) h->t[h->stage_pending_slot].release(h->pool);   // never a half-built mesh
        h->stage_error_msg = std::string("mesh build: ") + hipGetErrorString(se != hipSuccess ? se : le);
        return fail(-101, h->stage_error_msg);
    }
    float ms1 = 0, ms2 = 0;
    (void)hipEventElapsedTime(&ms1, h->ev0, h->ev1);
    (void)hipEventElapsedTime(&ms2, h->ev1, h->ev2);
    h->ms_mesh = ms1;
    h->ms_reorder = ms2;
    return 0;
}

int32_t soap_engine_destroy(soap_engine *h) {
    if (!h) return 0;
    (void)hipSetDevice(h->cfg.device);
    (void)finish_stage(h);
    const bool trace = getenv("SOAP_ENGINE_TRACE") != nullptr;
    auto probe = [&](const char *tag) { const hipError_t e = hipGetLastError(); if (e != hipSuccess && trace) { fprintf(stderr, "[soap_engine] destroy: %s left %s\n", tag, hipGetErrorString(e)); fflush(stderr); } };
    probe("before");
    (void)hipStreamSynchronize(h->stream);
    probe("sync");
    for (int t = 0; t < NT_ALL; t++) h->t[t].release(h->pool);
    probe("types");
    free_halos(h);
    h->pool.purge();
    probe("purge");
    free_slabs(h);
    probe("slabs");
    for (int k = 0; k < 3; k++) {
        if (h->cls[k].d_queue) (void)hipFree(h->cls[k].d_queue);
        if (h->cls[k].d_todo) (void)hipFree(h->cls[k].d_todo);
        if (h->cls[k].done) (void)hipEventDestroy(h->cls[k].done);
        if (h->cls[k].start) (void)hipEventDestroy(h->cls[k].start);
        for (int q = 0; q < 2; q++) { if (h->cls[k].pstart[q]) (void)hipEventDestroy(h->cls[k].pstart[q]); if (h->cls[k].pdone[q]) (void)hipEventDestroy(h->cls[k].pdone[q]); }
        if (h->cls[k].stream) (void)hipStreamDestroy(h->cls[k].stream);
    }
    for (int v = 0; v < NSV; v++) {
        if (h->small[v].d_queue) (void)hipFree(h->small[v].d_queue);
        if (h->small[v].d_todo) (void)hipFree(h->small[v].d_todo);
        if (h->small[v].done) (void)hipEventDestroy(h->small[v].done);
        if (h->small[v].start) (void)hipEventDestroy(h->small[v].start);
        if (h->small[v].stream) (void)hipStreamDestroy(h->small[v].stream);
    }
    if (h->team_start) (void)hipEventDestroy(h->team_start);
    if (h->team_done) (void)hipEventDestroy(h->team_done);
    if (h->team_stream) (void)hipStreamDestroy(h->team_stream);
    if (h->d_teamctl) (void)hipFree(h->d_teamctl);
    probe("class resources");
    if (h->d_first) (void)hipFree(h->d_first);
    if (h->d_small_cols) (void)hipFree(h->d_small_cols);
    if (h->d_small_esc) (void)hipFree(h->d_small_esc);
    if (h->d_part_count) (void)hipFree(h->d_part_count);
    probe("small");
    if (h->d_table) (void)hipFree(h->d_table);
    if (h->d_calcs) (void)hipFree(h->d_calcs);
    if (h->d_coloff) (void)hipFree(h->d_coloff);
    if (h->d_queue) (void)hipFree(h->d_queue);
    if (h->d_params) (void)hipFree(h->d_params);
    if (h->d_params_pipe) (void)hipFree(h->d_params_pipe);
    if (h->d_params_small) (void)hipFree(h->d_params_small);
    if (h->d_phase) (void)hipFree(h->d_phase);
    if (h->d_hdr) (void)hipFree(h->d_hdr);
    if (h->d_gb) (void)hipFree(h->d_gb);
    for (int it = 0; it < 8; it++) for (int q = 0; q < 2; q++) if (h->ev_b[it][q]) (void)hipEventDestroy(h->ev_b[it][q]);
    if (h->d_gbins) (void)hipFree(h->d_gbins);
    for (int q = 0; q < 8; q++) if (h->d_bpart[q]) (void)hipFree(h->d_bpart[q]);
    if (h->d_gints) (void)hipFree(h->d_gints);
    for (int k = 0; k < 6; k++) if (h->ev_p[k]) (void)hipEventDestroy(h->ev_p[k]);
    if (h->d_probe) (void)hipFree(h->d_probe);
    if (h->d_probe_count) (void)hipFree(h->d_probe_count);
    probe("tables");
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev2) (void)hipEventDestroy(h->ev2);
    if (h->ev_in) (void)hipEventDestroy(h->ev_in);
    if (h->ev_zero) (void)hipEventDestroy(h->ev_zero);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    probe("events");
    delete h;
    return 0;
}

// sum of every 64th mass (the mean particle mass of the type, for k_predict)
__global__ void k_mass_sample(const void *mass, int value_bytes, long long n, double *partial /*[grid]*/) {
    double s = 0.0;
    for (long long i = (blockIdx.x * (long long)BLK + threadIdx.x) * 64; i < n; i += (long long)gridDim.x * BLK * 64)
        s += value_bytes == 4 ? (double)((const float *)mass)[i] : ((const double *)mass)[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ double w[BLK / 64];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int k = 0; k < BLK / 64; k++) t += w[k]; partial[blockIdx.x] = t; }
}
__global__ void k_widen(const float *in, long long n, double *out) {
    for (long long i = blockIdx.x * (long long)BLK + threadIdx.x; i < n; i += (long long)gridDim.x * BLK) out[i] = (double)in[i];
}
// weighted neutrino masses  m_eff = mass * weight  (SO_properties.py:381-383)
__global__ void k_weighted_mass(const void *mass, const void *weight, int value_bytes, long long n, double *out) {
    for (long long i = blockIdx.x * (long long)BLK + threadIdx.x; i < n; i += (long long)gridDim.x * BLK)
        out[i] = value_bytes == 4 ? (double)((const float *)mass)[i] * (double)((const float *)weight)[i]
                                  : ((const double *)mass)[i] * ((const double *)weight)[i];
}

static int32_t stage_type(soap_engine *h, int32_t ts, int64_t n, const double *pos, const void *vel,
                          const void *mass, int32_t value_bytes, const void *grnr, int32_t grnr_bytes, const int64_t *fof,
                          int32_t resolution, int32_t on_device);

int32_t soap_engine_stage_particles(soap_engine *h, int32_t ts, int64_t n, const double *pos,
                                    const void *vel, const void *mass, int32_t value_bytes, const void *grnr,
                                    int32_t grnr_bytes, const int64_t *fof, int32_t resolution,
                                    int32_t on_device) {
    if (!h) return fail(-1, "null handle");
    if (ts < 0 || ts >= SOAP_NTYPES) return fail(-2, "type_slot out of range");
    if (n > 0 && (!vel || !grnr || !fof)) return fail(-6, "null particle array");
    return stage_type(h, ts, n, pos, vel, mass, value_bytes, grnr, grnr_bytes, fof, resolution, on_device);
}

int32_t soap_engine_stage_neutrinos(soap_engine *h, int64_t n, const double *pos, const void *mass,
                                    const void *weight, int32_t value_bytes, int32_t resolution, int32_t on_device) {
    if (!h) return fail(-1, "null handle");
    if (n > 0 && !weight) return fail(-6, "null particle array");
    // the weights ride in the `vel` argument (3 values per particle are not needed: one is), ids are absent
    return stage_type(h, NU_SLOT, n, pos, weight, mass, value_bytes, nullptr, 4, nullptr, resolution, on_device);
}

int32_t soap_engine_wait_stream(soap_engine *h, void *producer_stream) {
    if (!h) return fail(-1, "null handle");
    HIPCHECK(hipSetDevice(h->cfg.device));
    // everything submitted to the caller's stream so far (the kernels that produced, widened or copied the device arrays the
    // next stage_* call reads) is ordered in front of the engine's own work: an event, no host synchronisation
    HIPCHECK(hipEventRecord(h->ev_in, (hipStream_t)producer_stream));
    HIPCHECK(hipStreamWaitEvent(h->stream, h->ev_in, 0));
    return 0;
}

static int32_t stage_type(soap_engine *h, int32_t ts, int64_t n, const double *pos, const void *vel,
                          const void *mass, int32_t value_bytes, const void *grnr, int32_t grnr_bytes, const int64_t *fof,
                          int32_t resolution, int32_t on_device) {
    const bool nu = ts == NU_SLOT;
    if (n < 0 || n >= (1ll << PIDX_BITS)) return fail(-3, "particle count per type must be < 2^29");
    if (grnr_bytes != 4 && grnr_bytes != 8) return fail(-4, "grnr_bytes must be 4 or 8");
    if (value_bytes != 4 && value_bytes != 8) return fail(-4, "value_bytes (Velocities / Masses / Weights) must be 4 or 8");
    if (resolution < 1 || resolution > 256) return fail(-5, "mesh resolution must be in [1,256]");
    HIPCHECK(hipSetDevice(h->cfg.device));
    { const int32_t frc = finish_stage(h); if (frc) return frc; }
    (void)hipGetLastError();   // (a stale error of an earlier, unrelated call is not this call's failure)
    DevPool &pool = h->pool;
    h->t[ts].release(pool);
    if (n == 0) return 0;   // SharedMesh.empty (shared_mesh.py:25-27)
    if (!pos || !vel || !mass) return fail(-6, "null particle array");
    hipStream_t st = h->stream;
    // The new mesh is built in a local object and installed only when every step succeeded: a failed allocation
    // or kernel leaves the slot empty (n == 0), never half-built, and everything allocated on the way goes back
    // to the cache.
    TypeHost T;
    std::vector<std::pair<void *, size_t>> inputs;    // device copies of host inputs
    std::vector<std::pair<void *, size_t>> scratch;   // temporaries of the mesh build
    const hipEvent_t ev2 = h->ev2;
    bool deferred = false;
    auto grab = [&](void **p, size_t bytes, std::vector<std::pair<void *, size_t>> &owner) -> hipError_t {
        hipError_t e = pool.alloc(p, bytes);
        if (e == hipSuccess) owner.push_back({*p, bytes});
        return e;
    };
    auto body = [&]() -> int32_t {
        const double *dpos = pos;
        const void *dvel = vel, *dmass = mass;
        const void *dgr = grnr;
        const long long *dfof = (const long long *)fof;
        if (!on_device) {
            const size_t vbytes = (size_t)n * (nu ? 1 : 3) * (size_t)value_bytes;   // neutrinos: one weight per particle in place of a velocity
            void *a, *b, *c, *d = nullptr, *e = nullptr;
            HIPCHECK(grab(&a, (size_t)n * 24, inputs));
            HIPCHECK(grab(&b, vbytes, inputs));
            HIPCHECK(grab(&c, (size_t)n * value_bytes, inputs));
            HIPCHECK(hipMemcpyAsync(a, pos, (size_t)n * 24, hipMemcpyHostToDevice, st));
            HIPCHECK(hipMemcpyAsync(b, vel, vbytes, hipMemcpyHostToDevice, st));
            HIPCHECK(hipMemcpyAsync(c, mass, (size_t)n * value_bytes, hipMemcpyHostToDevice, st));
            if (!nu) {
                HIPCHECK(grab(&d, (size_t)n * grnr_bytes, inputs));
                HIPCHECK(grab(&e, (size_t)n * 8, inputs));
                HIPCHECK(hipMemcpyAsync(d, grnr, (size_t)n * grnr_bytes, hipMemcpyHostToDevice, st));
                HIPCHECK(hipMemcpyAsync(e, fof, (size_t)n * 8, hipMemcpyHostToDevice, st));
            }
            dpos = (const double *)a; dvel = b; dmass = c;
            dgr = d; dfof = (const long long *)e;
        }
        HIPCHECK(hipEventRecord(h->ev0, st));
        // bounding box (shared_mesh.py:33-61)
        const int gmm = 1024;
        double *d_partial;
        HIPCHECK(grab((void **)&d_partial, gmm * 6 * sizeof(double), scratch));
        hipLaunchKernelGGL(k_minmax, dim3(gmm), dim3(BLK), 0, st, dpos, (long long)n, d_partial);
        std::vector<double> part(gmm * 6);
        HIPCHECK(hipMemcpyAsync(part.data(), d_partial, gmm * 6 * sizeof(double), hipMemcpyDeviceToHost, st));
        const int gms = 64;
        double *d_msum;
        std::vector<double> msum(gms);
        HIPCHECK(grab((void **)&d_msum, gms * sizeof(double), scratch));
        hipLaunchKernelGGL(k_mass_sample, dim3(gms), dim3(BLK), 0, st, dmass, (int)value_bytes, (long long)n, d_msum);
        HIPCHECK(hipMemcpyAsync(msum.data(), d_msum, gms * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        { double t = 0.0; for (double v : msum) t += v; T.mean_mass = t / (double)((n + 63) / 64); }
        for (int d = 0; d < 3; d++) { T.pos_min[d] = INFINITY; T.pos_max[d] = -INFINITY; }
        for (int b = 0; b < gmm; b++)
            for (int d = 0; d < 3; d++) {
                T.pos_min[d] = std::min(T.pos_min[d], part[b * 6 + d]);
                T.pos_max[d] = std::max(T.pos_max[d], part[b * 6 + 3 + d]);
            }
        for (int d = 0; d < 3; d++) {
            if (!(T.pos_min[d] <= T.pos_max[d])) return fail(-7, "non-finite coordinates");
            if (T.pos_min[d] == T.pos_max[d]) T.pos_max[d] = T.pos_min[d] + 1.0;
            T.cell_size[d] = (T.pos_max[d] - T.pos_min[d]) / (double)resolution;
        }
        T.res = resolution;
        T.n = n;
        T.grnr64 = grnr_bytes == 8;
        T.vbytes = value_bytes;
        const long long ncell = (long long)resolution * resolution * resolution;

        unsigned *d_cell, *d_count, *d_rank, *d_tilesum;
        HIPCHECK(grab((void **)&d_cell, (size_t)n * 4, scratch));
        HIPCHECK(grab((void **)&d_count, (size_t)ncell * 4, scratch));
        HIPCHECK(grab((void **)&d_rank, (size_t)n * 4, scratch));
        HIPCHECK(pool.alloc((void **)&T.cell_offset, (size_t)(ncell + 1) * 4));
        HIPCHECK(hipMemsetAsync(d_count, 0, (size_t)ncell * 4, st));
        MeshGeom g;
        for (int d = 0; d < 3; d++) { g.pos_min[d] = T.pos_min[d]; g.cell_size[d] = T.cell_size[d]; }
        g.res = resolution;
        int grid = (int)std::min<long long>((n + BLK - 1) / BLK, 256 * 8);
        hipLaunchKernelGGL(k_cellid, dim3(grid), dim3(BLK), 0, st, dpos, (long long)n, g, d_cell, d_count, d_rank);
        long long ntiles = (ncell + SCAN_TILE - 1) / SCAN_TILE;
        HIPCHECK(grab((void **)&d_tilesum, (size_t)ntiles * 4, scratch));
        hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)ntiles), dim3(BLK), 0, st, d_count, ncell, T.cell_offset, d_tilesum);
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(BLK), 0, st, d_tilesum, ntiles);
        hipLaunchKernelGGL(k_scan_add, dim3((unsigned)ntiles), dim3(BLK), 0, st, T.cell_offset, ncell, d_tilesum, (unsigned)n);
        HIPCHECK(hipEventRecord(h->ev1, st));

        // placement + coordinates into cell order
        double **dp[] = {&T.x, &T.y, &T.z};
        for (double **p : dp) HIPCHECK(pool.alloc((void **)p, (size_t)n * 8));
        // (velocities, masses and ids are only read by the placement pass below, which copies them into the payload records:
        // the pointers live in locals, nothing of the caller's -- or of the device copies of host inputs -- is kept)
        const void *t_vel = dvel, *t_mass = dmass;
        int t_vbytes = value_bytes;
        const long long *t_fof = dfof, *t_g64 = T.grnr64 ? (const long long *)dgr : nullptr;
        const int *t_g32 = T.grnr64 ? nullptr : (const int *)dgr;
        if (nu) {
            // profile mass = mass * weight (owned array); the raw masses take the place of the velocities
            double *meff;
            HIPCHECK(pool.alloc((void **)&meff, (size_t)n * 8));
            scratch.push_back({meff, (size_t)n * 8});
            hipLaunchKernelGGL(k_weighted_mass, dim3(grid), dim3(BLK), 0, st, dmass, dvel, (int)value_bytes, (long long)n, meff);
            if (value_bytes == 4) {
                // the raw masses ride in the velocity slot of the record next to the fp64 weighted ones: widen them too
                double *raw;
                HIPCHECK(pool.alloc((void **)&raw, (size_t)n * 8));
                scratch.push_back({raw, (size_t)n * 8});
                hipLaunchKernelGGL(k_widen, dim3(grid), dim3(BLK), 0, st, (const float *)dmass, (long long)n, raw);
                t_vel = raw;
            } else t_vel = dmass;
            t_mass = meff;
            t_vbytes = 8;
            t_g32 = nullptr; t_g64 = nullptr; t_fof = nullptr;
        }
        HIPCHECK(pool.alloc((void **)&T.pay, (size_t)n * sizeof(Pay)));
        ThinBuild thin;
        memset(&thin, 0, sizeof(thin));
        unsigned *d_tfill[2] = {nullptr, nullptr};
        const bool want_thin = !nu && !(getenv("SOAP_ENGINE_THIN") && atoi(getenv("SOAP_ENGINE_THIN")) == 0);
        if (want_thin) {
            // thinned samples for k_predict: one particle in 64 / in 1024 by a hash of its coordinates, appended by the placement pass
            HIPCHECK(grab((void **)&thin.cursor, (size_t)grid * 2 * sizeof(unsigned), scratch));
            for (int lv = 0; lv < 2; lv++) {
                // every block of the placement pass appends to a sub-list of its own: twice what it expects (the hash is uniform) and some
                const size_t sub = (size_t)n / (size_t)grid / (lv == 0 ? 32 : 512) + 256;
                const size_t cap = sub * (size_t)grid;
                thin.cap[lv] = (unsigned)sub;
                T.thin_cap[lv] = cap;
                HIPCHECK(grab((void **)&thin.count[lv], (size_t)ncell * 4, scratch));
                HIPCHECK(grab((void **)&d_tfill[lv], (size_t)ncell * 4, scratch));
                HIPCHECK(grab((void **)&thin.lc[lv], cap * 4, scratch));
                HIPCHECK(grab((void **)&thin.lx[lv], cap * 8, scratch));
                HIPCHECK(grab((void **)&thin.ly[lv], cap * 8, scratch));
                HIPCHECK(grab((void **)&thin.lz[lv], cap * 8, scratch));
                HIPCHECK(hipMemsetAsync(thin.count[lv], 0, (size_t)ncell * 4, st));
                HIPCHECK(hipMemsetAsync(d_tfill[lv], 0, (size_t)ncell * 4, st));
                HIPCHECK(pool.alloc((void **)&T.thin_off[lv], (size_t)(ncell + 1) * 4));
                for (int d = 0; d < 3; d++) HIPCHECK(pool.alloc((void **)&T.thin_xyz[lv][d], cap * 8));
            }
        }
#define SCATTER_ARGS d_cell, d_rank, (long long)n, T.cell_offset, dpos, T.x, T.y, T.z, T.pay, t_vel, t_mass, t_vbytes, t_g32, t_g64, t_fof, nu ? 1 : 0, thin
        if (t_vbytes == 4) {
            if (want_thin) hipLaunchKernelGGL((k_scatter_pos<true, true>), dim3(grid), dim3(BLK), 0, st, SCATTER_ARGS);
            else hipLaunchKernelGGL((k_scatter_pos<true, false>), dim3(grid), dim3(BLK), 0, st, SCATTER_ARGS);
        } else {
            if (want_thin) hipLaunchKernelGGL((k_scatter_pos<false, true>), dim3(grid), dim3(BLK), 0, st, SCATTER_ARGS);
            else hipLaunchKernelGGL((k_scatter_pos<false, false>), dim3(grid), dim3(BLK), 0, st, SCATTER_ARGS);
        }
#undef SCATTER_ARGS
        if (want_thin) {
            for (int lv = 0; lv < 2; lv++) {
                hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)ntiles), dim3(BLK), 0, st, (const unsigned *)thin.count[lv], ncell, T.thin_off[lv], d_tilesum);
                hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(BLK), 0, st, d_tilesum, ntiles);
                hipLaunchKernelGGL(k_scan_add, dim3((unsigned)ntiles), dim3(BLK), 0, st, T.thin_off[lv], ncell, d_tilesum, 0u);
                hipLaunchKernelGGL(k_thin_place, dim3((unsigned)grid), dim3(BLK), 0, st, (const unsigned *)thin.cursor, lv, thin.cap[lv], grid,
                                   (const unsigned *)thin.lc[lv], (const double *)thin.lx[lv], (const double *)thin.ly[lv], (const double *)thin.lz[lv],
                                   T.thin_off[lv], ncell, (const unsigned *)thin.count[lv], d_tfill[lv], T.thin_xyz[lv][0], T.thin_xyz[lv][1], T.thin_xyz[lv][2]);
            }
        }
        HIPCHECK(hipEventRecord(ev2, st));
        if (on_device == 2) {
            // the caller keeps the inputs alive and unmodified until the next call on this handle that needs the mesh: the placement
            // pass stays in flight, finish_stage() completes the call
            hipError_t le = hipGetLastError();
            if (le != hipSuccess) return fail(-101, std::string("mesh build: ") + hipGetErrorString(le));
            deferred = true;
            return 0;
        }
        HIPCHECK(hipStreamSynchronize(st));
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) return fail(-101, std::string("mesh build: ") + hipGetErrorString(le));
        float ms1 = 0, ms2 = 0;
        (void)hipEventElapsedTime(&ms1, h->ev0, h->ev1);
        (void)hipEventElapsedTime(&ms2, h->ev1, ev2);
        h->ms_mesh = ms1;
        h->ms_reorder = ms2;
        return 0;
    };
    const int32_t rc = body();
    if (deferred && rc == 0) {
        h->stage_scratch = scratch;
        h->stage_pending = true;
        h->stage_pending_slot = ts;
        h->t[ts] = T;
        return 0;
    }
    for (auto &p : scratch) pool.release(p.first, p.second);
    if (rc != 0) {
        (void)hipStreamSynchronize(st);   // nothing may still be writing into what goes back to the cache
        for (auto &p : inputs) pool.release(p.first, p.second);
        T.release(pool);
        return rc;
    }
    // the device copies of host inputs go back to the cache: everything the halo kernels (and the membership probe) read lives
    // in the cell-ordered coordinates and the payload records
    for (auto &p : inputs) pool.release(p.first, p.second);
    h->t[ts] = T;
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
    HIPCHECK(h->pool.alloc((void **)&h->d_cofp, (size_t)n * 24));
    HIPCHECK(h->pool.alloc((void **)&h->d_search, (size_t)n * 8));
    HIPCHECK(h->pool.alloc((void **)&h->d_read, (size_t)n * 8));
    HIPCHECK(h->pool.alloc((void **)&h->d_index, (size_t)n * 8));
    HIPCHECK(h->pool.alloc((void **)&h->d_nrbound, (size_t)n * 8));
    HIPCHECK(h->pool.alloc((void **)&h->d_stats, (size_t)n * 8 * SOAP_STATS_COLUMNS));
    HIPCHECK(h->pool.alloc((void **)&h->d_central, (size_t)n * 4));
    HIPCHECK(h->pool.alloc((void **)&h->d_status, (size_t)n * 4));
    HIPCHECK(h->pool.alloc((void **)&h->d_todo, (size_t)n * 4));
    // (not behind a placement pass that may still be in flight on the engine's main stream: the teams' stream is idle between runs.
    // No stream of its own -- one more stream changes which hardware queues the class streams share, and with that the halo phase)
    hipStream_t st = h->team_stream;
    HIPCHECK(hipMemcpyAsync(h->d_cofp, cofp, (size_t)n * 24, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(h->d_search, search_radius, (size_t)n * 8, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(h->d_read, read_radius, (size_t)n * 8, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(h->d_index, index, (size_t)n * 8, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(h->d_nrbound, nr_bound_part, (size_t)n * 8, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(h->d_central, is_central, (size_t)n * 4, hipMemcpyHostToDevice, st));
    h->h_nrbound.assign(nr_bound_part, nr_bound_part + n);
    HIPCHECK(hipMemsetAsync(h->d_status, 0, (size_t)n * 4, st));
    HIPCHECK(hipMemsetAsync(h->d_stats, 0, (size_t)n * 8 * SOAP_STATS_COLUMNS, st));
    HIPCHECK(hipStreamSynchronize(st));
    return 0;
}

int32_t soap_engine_set_resident_radius(soap_engine *h, const double *resident_radius) {
    if (!h) return fail(-1, "null handle");
    HIPCHECK(hipSetDevice(h->cfg.device));
    const size_t n = (size_t)h->n_halo;
    if (h->d_resident) { h->pool.release(h->d_resident, n * 8); h->d_resident = nullptr; }
    if (!resident_radius || n == 0) return 0;
    HIPCHECK(h->pool.alloc((void **)&h->d_resident, n * 8));
    HIPCHECK(hipMemcpyAsync(h->d_resident, resident_radius, n * 8, hipMemcpyHostToDevice, h->team_stream));
    HIPCHECK(hipStreamSynchronize(h->team_stream));
    return 0;
}

int32_t soap_engine_set_member_probe(soap_engine *h, int64_t row, int32_t calc, int64_t cap) {
    if (!h) return fail(-1, "null handle");
    HIPCHECK(hipSetDevice(h->cfg.device));
    if (row < 0) { h->probe_row = -1; return 0; }
    if (calc < 0 || cap < 1) return fail(-2, "member probe: bad calculation or capacity");
    if (cap > h->probe_cap || !h->d_probe) {
        if (h->d_probe) (void)hipFree(h->d_probe);
        h->d_probe = nullptr;
        HIPCHECK(hipMalloc((void **)&h->d_probe, (size_t)cap * 8));
    }
    if (!h->d_probe_count) HIPCHECK(hipMalloc((void **)&h->d_probe_count, sizeof(unsigned)));
    HIPCHECK(hipMemset(h->d_probe_count, 0, sizeof(unsigned)));
    h->probe_row = row; h->probe_calc = calc; h->probe_cap = cap;
    return 0;
}

int64_t soap_engine_fetch_members(soap_engine *h, int32_t *type_slot, int64_t *index, int64_t cap) {
    if (!h) return fail(-1, "null handle");
    if (h->probe_row < 0 || !h->d_probe) return fail(-2, "no member probe set");
    HIPCHECK(hipSetDevice(h->cfg.device));
    unsigned cnt = 0;
    HIPCHECK(hipMemcpy(&cnt, h->d_probe_count, sizeof(unsigned), hipMemcpyDeviceToHost));
    const long long have = std::min<long long>(cnt, h->probe_cap);
    std::vector<unsigned long long> v((size_t)have);
    if (have > 0) HIPCHECK(hipMemcpy(v.data(), h->d_probe, (size_t)have * 8, hipMemcpyDeviceToHost));
    std::sort(v.begin(), v.end());
    const long long nw = std::min<long long>(have, cap);
    for (long long i = 0; i < nw; i++) {
        if (type_slot) type_slot[i] = (int32_t)(v[i] >> 56);
        if (index) index[i] = (int64_t)(v[i] & ((1ull << 56) - 1ull));
    }
    return (int64_t)cnt;
}

static long long next_pow2(long long v) { long long p = 1; while (p < v) p <<= 1; return p; }

int32_t soap_engine_run(soap_engine *h, const soap_calc_desc *calcs, int32_t n_calcs,
                        double target_density) {
    static const bool trace = getenv("SOAP_ENGINE_TRACE") != nullptr;
    const auto t_trace0 = std::chrono::steady_clock::now();
#define TRACE(msg) do { if (trace) { fprintf(stderr, "[soap_engine] %8.3f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_trace0).count(), msg); fflush(stderr); } } while (0)
    if (!h) return fail(-1, "null handle");
    TRACE("run: begin");
    // a staging call that returned with its placement pass in flight is completed FIRST, whatever this call goes on to return: the
    // caller drops its references to the staged arrays when `run` comes back
    (void)hipSetDevice(h->cfg.device);
    { const int32_t frc = finish_stage(h); if (frc) return frc; }
    if (n_calcs < 1 || n_calcs > MAXCALC)
        return fail(-2, "number of calculations must be in [1,64]: one lane of a wavefront and one bit of the done mask each; split a "
                        "longer list into several runs, each starting with the bound subhalo");
    if (!calcs) return fail(-3, "null calcs");
    if (calcs[0].kind != SOAP_CALC_SUBHALO)
        return fail(-4, "the first calculation must be the bound subhalo (compute_halo_properties.py:232-247)");
    HIPCHECK(hipSetDevice(h->cfg.device));
    std::vector<int> coloff(n_calcs);
    long long ncols = 0;
    bool want_inertia = false, want_property_radius = false;
    for (int c = 0; c < n_calcs; c++) want_inertia = want_inertia || (calcs[c].flags & SOAP_CALC_FLAG_INERTIA);
    for (int c = 0; c < n_calcs; c++) {
        int nc = soap_engine_desc_columns(&calcs[c]);
        if (nc < 0) return fail(-5, "unknown calculation kind");
        if (c > 0 && calcs[c].kind == SOAP_CALC_SUBHALO) return fail(-6, "only one subhalo calculation allowed");
        if (calcs[c].kind == SOAP_CALC_SO_MULTIPLE) {
            int p = calcs[c].parent;
            if (p < 1 || p >= c || calcs[p].kind < SOAP_CALC_SO_DENSITY || calcs[p].kind > SOAP_CALC_SO_MULTIPLE)
                return fail(-7, "SO radius multiple needs an earlier SO as parent (SO_properties.py:3912-3916)");
        }
        if (calcs[c].flags & SOAP_CALC_FLAG_PROPERTY_RADIUS) {
            if (calcs[c].kind != SOAP_CALC_APERTURE_EXCL && calcs[c].kind != SOAP_CALC_APERTURE_INCL)
                return fail(-9, "only spherical apertures can be defined by a bound-subhalo property (aperture_properties.py:3973-3981)");
            if (calcs[c].parent < 0 || calcs[c].parent >= SOAP_RADIUS_NPROPS)
                return fail(-9, "unknown bound-subhalo property for an aperture radius (SOAP_RADIUS_*)");
            want_property_radius = true;
        } else if ((calcs[c].kind == SOAP_CALC_APERTURE_EXCL || calcs[c].kind == SOAP_CALC_APERTURE_INCL || calcs[c].kind == SOAP_CALC_PROJECTED) && calcs[c].parent >= 0) {
            int p = calcs[c].parent;
            if (p < 1 || p >= c || calcs[p].kind != calcs[c].kind) return fail(-8, "aperture parent must be an earlier aperture of the same kind");
        }
        coloff[c] = (int)ncols;
        ncols += nc;
    }
    hipStream_t st = h->stream;
    const auto t_host0 = std::chrono::steady_clock::now();
    // result table and descriptors come from the allocation cache: a chunk loop re-runs with the same sizes
    if (h->d_table) { h->pool.release(h->d_table, h->table_bytes); h->d_table = nullptr; }
    if (h->d_calcs) { h->pool.release(h->d_calcs, h->calcs_bytes); h->d_calcs = nullptr; }
    if (h->d_coloff) { h->pool.release(h->d_coloff, h->coloff_bytes); h->d_coloff = nullptr; }
    h->ncols = ncols;
    if (h->n_halo == 0) return 0;
    h->table_bytes = (size_t)h->n_halo * ncols * 8;
    h->calcs_bytes = (size_t)n_calcs * sizeof(soap_calc_desc);
    h->coloff_bytes = (size_t)n_calcs * 4;
    HIPCHECK(h->pool.alloc((void **)&h->d_table, h->table_bytes));
    HIPCHECK(h->pool.alloc((void **)&h->d_calcs, h->calcs_bytes));
    HIPCHECK(h->pool.alloc((void **)&h->d_coloff, h->coloff_bytes));
    // the table is cleared ONCE, here, on a stream that is idle between runs, while the estimate kernels work (0.15 ms for the 766 MB of the C3
    // list at HBM speed): no group zeroes a row from the inside (957 columns per halo: 15 stores per lane of a small halo's wave, before its first load)
    HIPCHECK(hipMemsetAsync(h->d_table, 0, h->table_bytes, h->team_stream));
    HIPCHECK(hipEventRecord(h->ev_zero, h->team_stream));
    HIPCHECK(hipMemcpyAsync(h->d_calcs, calcs, (size_t)n_calcs * sizeof(soap_calc_desc), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(h->d_coloff, coloff.data(), (size_t)n_calcs * 4, hipMemcpyHostToDevice, st));

    long long ntot = 0;
    for (int t = 0; t < NT_ALL; t++) ntot += h->t[t].n;
    // room for every staged particle plus the power-of-two padding of the fallback sort of an appended shell
    const long long cap_max = 2 * next_pow2(std::max<long long>(ntot, 64));

    RunParams P;
    memset(&P, 0, sizeof(P));
    for (int t = 0; t < NT_ALL; t++) {
        const TypeHost &T = h->t[t];
        TypeDev &D = P.t[t];
        D.n = T.n; D.res = T.res; D.grnr64 = T.grnr64;
        D.x = T.x; D.y = T.y; D.z = T.z; D.vel = T.vel; D.mass = T.mass;
        D.grnr32 = T.g32; D.grnr64p = T.g64; D.fof = T.fof; D.orig = T.orig; D.cell_offset = T.cell_offset; D.pay = T.pay;
        for (int d = 0; d < 3; d++) { D.pos_min[d] = T.pos_min[d]; D.pos_max[d] = T.pos_max[d]; D.cell_size[d] = T.cell_size[d]; }
        D.mean_mass = T.mean_mass;
        for (int lv = 0; lv < 2; lv++) { D.thin_off[lv] = T.thin_off[lv]; D.thin_x[lv] = T.thin_xyz[lv][0]; D.thin_y[lv] = T.thin_xyz[lv][1]; D.thin_z[lv] = T.thin_xyz[lv][2]; }
        if (t < SOAP_NTYPES) P.soft[t] = h->cfg.softening[t];
    }
    P.halo.n = h->n_halo; P.halo.cofp = h->d_cofp; P.halo.search_radius = h->d_search; P.halo.read_radius = h->d_read;
    P.halo.index = h->d_index; P.halo.is_central = h->d_central; P.halo.nr_bound = h->d_nrbound;
    P.halo.status = h->d_status; P.halo.stats = h->d_stats; P.halo.todo = nullptr; P.halo.n_todo = 0;
    P.halo.resident = h->d_resident;
    P.probe_row = h->probe_row; P.probe_calc = h->probe_calc; P.probe_cap = h->probe_cap;
    P.probe_out = h->d_probe; P.probe_count = h->d_probe_count;
    if (h->probe_row >= 0) {
        if (h->probe_calc >= n_calcs || calcs[h->probe_calc].kind == SOAP_CALC_PROJECTED)
            return fail(-9, "member probe: calculation out of range or a projected aperture");
        HIPCHECK(hipMemsetAsync(h->d_probe_count, 0, sizeof(unsigned), st));
    }
    P.boxsize = h->cfg.boxsize; P.G_eff = h->cfg.G_eff; P.nu_density = h->cfg.nu_density; P.mpc20 = h->cfg.mpc20;
    P.a = h->cfg.a; P.mpc = h->cfg.mpc;
    P.G_phys = h->cfg.G_phys; P.hubble = h->cfg.hubble; P.rdot_omega = h->cfg.rdot_omega;
    { const char *e = getenv("SOAP_ENGINE_LITERAL"); P.literal = (e && e[0] == '1') ? 1 : 0; }
    // radius the loop is expected to end at, in units of the catalogue search radius: a 5xR500 multiple
    // needs ~3.3-4 R200, density thresholds down to 50 rho_crit ~2 R200 (halo_tasks.py:184-187 steps of 1.2)
    P.first_factor = 1.2 * 1.2;
    for (int c = 0; c < n_calcs; c++) {
        if (calcs[c].kind == SOAP_CALC_SO_DENSITY) P.first_factor = std::max(P.first_factor, 2.0736);
        if (calcs[c].kind == SOAP_CALC_SO_MULTIPLE) P.first_factor = std::max(P.first_factor, 0.86 * calcs[c].radius);
    }
    { const char *e = getenv("SOAP_ENGINE_PROFILE"); P.profile = e ? atoi(e) : 0; }
    P.hydro = (h->t[0].n > 0 || h->t[2].n > 0) ? 1 : 0;
    P.has_nu = h->t[NU_SLOT].n > 0 ? 1 : 0;
    // float32 Masses in every staged type: the SO mass profile goes through the reference's float32 casts (SO_properties.py:400-402);
    // a chunk that mixes widths concatenates to float64 in the reference, as here
    P.mass_f32 = 1;
    for (int t = 0; t < NT_ALL; t++) if (h->t[t].n > 0 && h->t[t].vbytes != 4) P.mass_f32 = 0;
    if (ntot == 0) P.mass_f32 = 0;
    P.dm_only = (h->t[0].n == 0 && h->t[2].n == 0 && h->t[3].n == 0 && !P.has_nu) ? 1 : 0;
    P.target_density = target_density;
    P.ncalc = n_calcs; P.ncols = ncols; P.table = h->d_table; P.queue = h->d_queue;
    P.calcs = h->d_calcs; P.coloff = h->d_coloff;
    P.phase = h->d_phase;
    // Experiment (builds with -DSOAP_EXPERIMENTS, SOAP_ENGINE_SPLIT_LARGE=1): chunks with gas and stars, a halo of the 1024-thread class shared
    // by TWO groups -- both build the sorted slab and run the radius loop (the same decisions to the bit), then one does the calculations
    // over all particles, the other those over the bound particles and the projected apertures (stage_b's shares).  Same values; on the C5
    // share the class ends at 11.3 instead of 15.2 ms and the phase stays at 15.4: the second build of 65 halos (7.4 of a halo's 15 ms) costs
    // the 500 CU ms the other CUs were idle for.
    bool split_large = false;
    { const char *e = expt_env("SOAP_ENGINE_SPLIT_LARGE"); if (e) split_large = atoi(e) != 0 && P.hydro && !want_inertia && h->n_halo < (1ll << 29) && h->probe_row < 0 && !P.literal; }
    if (split_large) {
        if (h->part_count_n < (size_t)h->n_halo) {
            if (h->d_part_count) (void)hipFree(h->d_part_count);
            h->d_part_count = nullptr;
            h->part_count_n = (size_t)h->n_halo + (size_t)h->n_halo / 4;
            HIPCHECK(hipMalloc((void **)&h->d_part_count, h->part_count_n * sizeof(int)));
        }
        P.halo.part_count = h->d_part_count;
    }
    HIPCHECK(hipMemsetAsync(h->d_phase, 0, 32 * sizeof(unsigned long long), st));
    HIPCHECK(hipMemcpyAsync(h->d_params, &P, sizeof(RunParams), hipMemcpyHostToDevice, st));

    // three classes of halos by estimated sphere content: one wavefront, 256 threads, 1024 threads per
    // halo.  A halo whose search sphere outgrows its class' slab is retried in the next class.
    // (thresholds from scripts/sweep.sh on C3: 8192 / 131072 gave 29.4 ms of halo kernels, 16384 / 262144 27.3 ms;
    // the largest halo alone keeps one 1024-thread group busy for ~22 ms, so the giants class should hold little else)
    // Chunks with gas and stars do about twice the work per particle (typed passes, projected apertures): there the
    // 256-thread class keeps the lower limit (C5: 131072 gives 29.7 ms/step, 262144 37.9 ms/step).
    long long nb_small = 16384, nb_medium = 262144;
    if (h->t[0].n > 0 || h->t[2].n > 0) nb_medium = 131072;
    { const char *e = getenv("SOAP_ENGINE_NB_SMALL"); if (e) nb_small = atoll(e); }
    { const char *e = expt_env("SOAP_ENGINE_NB_MEDIUM"); if (e) nb_medium = atoll(e); }
    // (twice the class limit: the radius loop may take a sphere a round or two further than the estimate saw, and a halo
    // that outgrows its slab is done again in the next class, in a pass of its own -- 10 ms on C3 for three seeds in four
    // when the 256-thread class had no headroom)
    long long caps[3] = {1ll << 16, 1ll << 19, 1ll << 20};
    int groups[3] = {2048, 512, 128};
    if (h->cfg.slab_capacity > 0) {
        long long c0 = next_pow2(h->cfg.slab_capacity);
        caps[0] = c0; caps[1] = c0 * 8; caps[2] = c0 * 64;
        nb_small = c0 / 4; nb_medium = c0 * 2;
    }
    { const char *e = expt_env("SOAP_ENGINE_G0"); if (e) groups[0] = atoi(e); }
    { const char *e = expt_env("SOAP_ENGINE_G1"); if (e) groups[1] = atoi(e); }
    { const char *e = expt_env("SOAP_ENGINE_G2"); if (e) groups[2] = atoi(e); }
    if (h->cfg.n_workgroups > 0) { groups[0] = h->cfg.n_workgroups * 4; groups[1] = h->cfg.n_workgroups; groups[2] = std::max(1, h->cfg.n_workgroups / 4); }
    // never below 256 entries: stage B parks 3 values per calculation in two slab columns (finalize_moments)
    for (int k = 0; k < 3; k++) caps[k] = std::max(std::min(caps[k], cap_max), 256ll);
    std::vector<long long> est(h->n_halo);
    const auto t_host0b = std::chrono::steady_clock::now();
    {
        // where each halo's radius loop is expected to end (k_predict), then the candidates of the first superset out to there:
        // the work estimate the classes, the launch order and the slab sizes go by
        int pred = (P.literal || h->cfg.slab_capacity > 0) ? 0 : 1;
        { const char *e = getenv("SOAP_ENGINE_PREDICT"); if (e) pred = atoi(e); }
        for (int t = 0; t < SOAP_NTYPES; t++) if (h->t[t].n > 0 && !h->t[t].thin_off[0]) pred = 0;   // (staged without the thinned copies)
        if (h->first_count < (size_t)h->n_halo) {
            if (h->d_first) (void)hipFree(h->d_first);
            h->d_first = nullptr;
            h->first_count = (size_t)h->n_halo + (size_t)h->n_halo / 4;
            HIPCHECK(hipMalloc((void **)&h->d_first, h->first_count * sizeof(double)));
        }
        h->h_first.assign((size_t)h->n_halo, 0.0);
        h->n_predicted = 0;
        int pred_min = 32;
        double pred_margin = 1.0;
        if (pred) {
            { const char *e = getenv("SOAP_ENGINE_PRED_MIN"); if (e) pred_min = atoi(e); }
            { const char *e = getenv("SOAP_ENGINE_PRED_MARGIN"); if (e) pred_margin = atof(e); }
            P.halo.first_radius = h->d_first;
        }
        long long *d_est;
        HIPCHECK(h->pool.alloc((void **)&d_est, (size_t)h->n_halo * 8));
        const unsigned gw = (unsigned)((h->n_halo + BLK / 64 - 1) / (BLK / 64));
        hipLaunchKernelGGL(k_estimate, dim3(gw), dim3(BLK), 0, st, (const RunParams *)h->d_params, d_est, (const double *)nullptr);
        if (pred) {
            const int pred_force = getenv("SOAP_ENGINE_PRED_FORCE") && atoi(getenv("SOAP_ENGINE_PRED_FORCE")) != 0;
            long long pred_always = nb_small;
            { const char *e = expt_env("SOAP_ENGINE_PRED_ALWAYS"); if (e) pred_always = atoll(e); }
            hipLaunchKernelGGL(k_predict, dim3(gw), dim3(BLK), 0, st, (const RunParams *)h->d_params, h->d_first, d_est, pred_min, pred_margin, pred_force, pred_always);
            HIPCHECK(hipGetLastError());
            HIPCHECK(hipMemcpyAsync(h->h_first.data(), h->d_first, (size_t)h->n_halo * 8, hipMemcpyDeviceToHost, st));
            HIPCHECK(hipMemcpyAsync(h->d_params, &P, sizeof(RunParams), hipMemcpyHostToDevice, st));   // (first_radius set)
        }
        HIPCHECK(hipMemcpyAsync(est.data(), d_est, (size_t)h->n_halo * 8, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        h->pool.release(d_est, (size_t)h->n_halo * 8);
        for (double v : h->h_first) h->n_predicted += v > 0.0;
        if (const char *dump = getenv("SOAP_ENGINE_DUMP_EST")) {   // diagnostic: the estimates and the guessed radii of this run, raw (int64[n], float64[n])
            if (FILE *f = fopen(dump, "wb")) { fwrite(est.data(), 8, (size_t)h->n_halo, f); fwrite(h->h_first.data(), 8, (size_t)h->n_halo, f); fclose(f); }
        }
    }
    TRACE("run: estimate done");
    // ---- class limits follow the load.  A halo is a serial chain on ONE thread group: ~4e-7 s per candidate on a
    // wavefront (6.7 ms for 16 k candidates, measured alone), ~6e-8 s on 256 threads, ~3.8e-8 s on 1024 threads, however
    // empty the GPU is; the whole run needs ~W / 1.2e10 s (W: candidates of all halos; half that rate with gas and
    // stars).  On the full C3 chunk the chains hide behind 1e5 other halos and the limits are the tuned 16384 / 262144;
    // on an eighth of it (one rank of 8) or on c2 they would BE the run, so no halo may take longer than tau = 3/4 of
    // the run's own time on its class.  What is too long even for 1024 threads goes through the multi-workgroup
    // pipeline (giants' mode below).
    double run_W = 0.0, run_est_max = 0.0;
    for (long long i = 0; i < h->n_halo; i++) { const double e = (double)std::max<long long>(est[i], 0); run_W += e; run_est_max = std::max(run_est_max, e); }
    const double run_t_all = run_W / (P.hydro ? 0.6e10 : 1.2e10) + 1.0e-3;
    const double run_tau = std::max(0.75 * run_t_all, 2.0e-3);
    const bool load_classes = h->cfg.slab_capacity <= 0 && !getenv("SOAP_ENGINE_NB_SMALL") && !expt_env("SOAP_ENGINE_NB_MEDIUM") &&
                              !(expt_env("SOAP_ENGINE_LOAD_CLASSES") && atoi(expt_env("SOAP_ENGINE_LOAD_CLASSES")) == 0);
    if (load_classes) {
        nb_small = std::min<long long>(nb_small, std::max<long long>(2048, (long long)(run_tau / 4.0e-7)));
        const double medium_cost = expt_env("SOAP_ENGINE_MEDIUM_COST") ? atof(expt_env("SOAP_ENGINE_MEDIUM_COST")) : 6.0e-8;   // s per candidate, 256 threads
        nb_medium = std::min<long long>(nb_medium, std::max<long long>(2 * nb_small, (long long)(run_tau / medium_cost)));
    }
    // ---- three-stage pipeline: build the sorted superset of every halo in its arena slice
    // (the class kernels, stage 1), run the radius loop of ALL halos with one lane per halo (stage_a.hip, stage 2), then
    // the property passes (the class kernels again, stage 3).  Halos the pipeline cannot finish -- radius ties, bound
    // counts that disagree with the catalogue, a loop that runs past the first superset -- fall through to the
    // single-kernel path below, which also serves the runs with inertia tensors / neutrinos / the literal mode.
    // one launch of the single-kernel path per class on the class streams (no host synchronisation here)
    // the wavefront class' unit with LDS tables for 32 calculations (eight groups per CU instead of seven) for the lists that fit
    const bool narrow64 = n_calcs <= 32 && !(expt_env("SOAP_ENGINE_NARROW64") && atoi(expt_env("SOAP_ENGINE_NARROW64")) == 0);
    // (shared halos: a group per task where the slabs fit in memory -- the tasks of a halo should start together -- with the same headroom
    // of three radius rounds as below, here without giving up groups for it: 105 B per slab entry, at most 64 GB in all)
    auto size_top_split = [&](long long n_tasks, long long need) {
        if (!split_large || n_tasks == 0) return;
        groups[2] = (int)std::max<long long>(groups[2], std::min<long long>(n_tasks, 256));
        while (caps[2] < cap_max && caps[2] < 4 * need && 2.0 * (double)caps[2] * (double)groups[2] * 105.0 <= 64.0e9) caps[2] = std::min(caps[2] * 2, cap_max);
    };
    auto launch_single = [&](std::vector<int> (&lists)[3], hipEvent_t after, bool join) -> int32_t {
        // submission order: the 1024-thread class (few, long halos) first.  Without three priority levels
        // the 1024-thread and wavefront classes share one stream and the 256-thread class takes the other
        // (two equal-priority user streams get distinct hardware queues, three do not).
        const int order[3] = {2, 1, 0};
        for (int oi = 0; oi < 3; oi++) {
            const int k = order[oi];
            soap_engine::Class &C = h->cls[k];
            hipStream_t cs = h->own_streams ? C.stream : ((k == 1) ? h->cls[0].stream : h->cls[2].stream);
            if (lists[k].empty()) continue;
            const int *task_list = lists[k].data();
            long long nt = (long long)lists[k].size();
            if (k == 2 && split_large) {
                // two tasks per row, side by side in the queue: bits 29-31 = the shares (PART_* of stage_b: 1 all particles, 2 | 4 bound particles + projected apertures)
                h->split_rows.clear();
                for (int i : lists[k]) { h->split_rows.push_back((int)((unsigned)i | (1u << 29))); h->split_rows.push_back((int)((unsigned)i | (6u << 29))); }
                task_list = h->split_rows.data();
                nt = (long long)h->split_rows.size();
            }
            const int nwg = (int)std::min<long long>(groups[k], nt);
            int rc = alloc_slab(C, caps[k], groups[k]);
            if (rc) return rc;
            if (C.todo_cap < nt) {
                if (C.d_todo) (void)hipFree(C.d_todo);
                C.todo_cap = std::max<long long>(nt, 2 * h->n_halo);
                HIPCHECK(hipMalloc((void **)&C.d_todo, (size_t)C.todo_cap * 4));
            }
            HIPCHECK(hipStreamWaitEvent(cs, after, 0));
            HIPCHECK(hipMemcpyAsync(C.d_todo, task_list, (size_t)nt * 4, hipMemcpyHostToDevice, cs));
            if (k == 2 && split_large) HIPCHECK(hipMemsetAsync(h->d_part_count, 0, (size_t)h->n_halo * sizeof(int), cs));
            HIPCHECK(hipMemsetAsync(C.d_queue, 0, sizeof(unsigned), cs));
            const RunParams *dp = h->d_params;
            HIPCHECK(hipEventRecord(C.start, cs));
            C.nh += (long long)lists[k].size();
            if (!want_inertia) {
                if (k == 0 && narrow64) soap_launch_halo_64n32(nwg, cs, dp, C.slab, C.d_queue, (const int *)C.d_todo, nt);
                else if (k == 0) soap_launch_halo_64n(nwg, cs, dp, C.slab, C.d_queue, (const int *)C.d_todo, nt);
                else if (k == 1) soap_launch_halo_256n(nwg, cs, dp, C.slab, C.d_queue, (const int *)C.d_todo, nt);
                else soap_launch_halo_1024n(nwg, cs, dp, C.slab, C.d_queue, (const int *)C.d_todo, nt);
            } else {
                if (k == 0 && narrow64) soap_launch_halo_64i32(nwg, cs, dp, C.slab, C.d_queue, (const int *)C.d_todo, nt);
                else if (k == 0) soap_launch_halo_64i(nwg, cs, dp, C.slab, C.d_queue, (const int *)C.d_todo, nt);
                else if (k == 1) soap_launch_halo_256i(nwg, cs, dp, C.slab, C.d_queue, (const int *)C.d_todo, nt);
                else soap_launch_halo_1024i(nwg, cs, dp, C.slab, C.d_queue, (const int *)C.d_todo, nt);
            }
            HIPCHECK(hipEventRecord(C.done, cs));
            if (join) HIPCHECK(hipStreamWaitEvent(st, C.done, 0));
        }
        return 0;
    };
    bool pipeline = !want_inertia && !P.has_nu && !P.literal && !want_property_radius;
    // SOAP_ENGINE_PIPELINE: 0 the single-kernel path for every halo; 1 the mix described below; 2 every halo through the
    // pipeline; 3 (the default) the GIANTS through the pipeline -- halos whose sphere holds more than nb_giant candidates
    // are built by several workgroups each (k_g* kernels) instead of keeping one 1024-thread group busy for tens of
    // milliseconds -- and everything else on the single-kernel path, which is the fastest for the small halos
    int pmode = 3;
    { const char *e = getenv("SOAP_ENGINE_PIPELINE"); if (e) pmode = atoi(e); }
    pipeline = pipeline && pmode != 0;
    // Giants: the largest halo would keep a 1024-thread group busy for longer than giant_factor * tau (the
    // multi-workgroup kernels have to find room beside the persistent groups of the other classes, which costs more
    // than it saves unless the giants are the long pole by a clear margin: C3 and C5 on one GPU never qualify) and is
    // large enough for the pipeline's fixed latency (~8 ms of dependent launches) to pay.  Then everything beyond the
    // 256-thread class goes through the pipeline.  SOAP_ENGINE_NB_GIANT fixes the limit instead.
    long long nb_giant = 1ll << 62;
    {
        // (the busier the GPU, the clearer the margin has to be: beside 1e5 halos the multi-workgroup kernels wait for the
        // persistent groups to drain -- seed 5 of the C3 box, largest halo 5.7e5 candidates: 32.7 ms with it through the
        // pipeline, 27.8 ms without)
        double giant_factor = 1.0 + 0.5 * std::min(1.0, std::max(0.0, (run_t_all - 10.0e-3) / 10.0e-3));
        { const char *e = expt_env("SOAP_ENGINE_GIANT_FACTOR"); if (e) giant_factor = atof(e); }
        if (run_est_max * 3.8e-8 > giant_factor * run_tau && run_est_max > 2.1e5) nb_giant = nb_medium;
        const char *e = getenv("SOAP_ENGINE_NB_GIANT");
        if (e) nb_giant = atoll(e);
        if (trace) { fprintf(stderr, "[soap_engine] candidates %.3g largest %.3g: t_all %.2f ms tau %.2f ms limits %lld / %lld giants above %lld\n", run_W, run_est_max, 1e3 * run_t_all, 1e3 * run_tau, nb_small, nb_medium, nb_giant); fflush(stderr); }
    }
    // ---- the large halos start at once.  What comes next on the host -- the sort of 1e5 rows by estimate, the routing of the
    // small halos, the uploads of their lists -- takes over a millisecond of a C3 pass, and the thread-group classes are the
    // long chains of the halo phase: their rows (a few thousand: estimate above nb_small) are picked, sorted and launched
    // first, the rest of the classification runs while they work.  Not in the modes that hand large halos to the pipeline or
    // to teams (those decide what the classes get).
    std::vector<int> early[3];
    int only_class = -1;   // SOAP_ENGINE_ONLY_CLASS=k: profile one class alone (other halos stay unprocessed)
    { const char *e = getenv("SOAP_ENGINE_ONLY_CLASS"); if (e) only_class = atoi(e); }
    bool early_big = !(pipeline && (pmode == 1 || pmode == 2 || pmode == 4)) && !(pipeline && pmode == 3 && run_est_max > (double)nb_giant) && only_class < 0 &&
                     !(expt_env("SOAP_ENGINE_EARLY_BIG") && atoi(expt_env("SOAP_ENGINE_EARLY_BIG")) == 0);
    if (early_big) {
        std::vector<std::pair<long long, int>> big;
        for (long long i = 0; i < h->n_halo; i++) if (est[i] > nb_small) big.push_back({std::min<long long>(est[i], 0xffffffffLL), (int)i});
        std::sort(big.begin(), big.end(), [](const std::pair<long long, int> &x, const std::pair<long long, int> &y) { return x.first != y.first ? x.first > y.first : x.second < y.second; });
        for (auto &b : big) early[b.first <= nb_medium ? 1 : 2].push_back(b.second);
        {
            // (room in the largest class' slab for the largest sphere and three rounds more: see the single-kernel passes below)
            long long need = 0, n_top = (long long)early[2].size() * (split_large ? 2 : 1);
            for (int i : early[2]) need = std::max(need, est[i]);
            size_top_split(n_top, need);
            while (caps[2] < cap_max && caps[2] < 4 * need && groups[2] / 2 >= std::max<long long>(16, std::min<long long>(n_top, split_large ? 256 : 64))) { caps[2] = std::min(caps[2] * 2, cap_max); groups[2] = std::max(1, groups[2] / 2); }
            while (caps[2] < cap_max && caps[2] < need + need / 4) { caps[2] = std::min(caps[2] * 2, cap_max); groups[2] = std::max(1, groups[2] / 2); }
        }
        HIPCHECK(hipStreamWaitEvent(st, h->ev_zero, 0));   // (the cleared table)
        HIPCHECK(hipEventRecord(h->ev_p[0], st));
        for (int k = 0; k < 3; k++) { h->cls[k].ms = 0.0; h->cls[k].nh = 0; }
        const int rc = launch_single(early, h->ev_p[0], false);
        if (rc) return rc;
    }
    const auto t_host1 = std::chrono::steady_clock::now();
    // heaviest first inside every class (the reference orders by bound-particle count for the same
    // reason, chunk_tasks.py:118-120); ties keep the staged order.  One sort of packed (estimate, row) keys.
    std::vector<int> todo[3];
    {
        // LSD radix sort (3 x 11 bits) of the row numbers by estimate: rows enter in descending order and
        // the ascending result is read backwards, which gives descending estimates with ties in staged order
        const size_t n = (size_t)h->n_halo;
        // keys and row numbers travel together (one sequential read per pass instead of a gather through the permutation),
        // and a pass whose digit is zero in every key is skipped
        // (scratch kept in the handle: fresh vectors of this size are fresh pages every call)
        std::vector<unsigned long long> &ka = h->sort_ka, &kb = h->sort_kb;
        std::vector<unsigned> &a = h->sort_rows, &e32 = h->sort_e32;
        ka.resize(n); kb.resize(n); a.resize(n); e32.resize(n);
        unsigned emax = 0u;
        for (size_t i = 0; i < n; i++) {
            e32[i] = (unsigned)std::min<long long>(std::max<long long>(est[i], 0), 0xffffffffLL);
            emax = std::max(emax, e32[i]);
        }
        for (size_t i = 0; i < n; i++) ka[i] = ((unsigned long long)e32[n - 1 - i] << 32) | (unsigned long long)(n - 1 - i);
        for (int pass = 0; pass < 3; pass++) {
            const int shift = 32 + 11 * pass;
            if (pass > 0 && (emax >> (11 * pass)) == 0u) break;
            size_t cnt[2049] = {0};
            for (size_t i = 0; i < n; i++) cnt[((ka[i] >> shift) & 2047ull) + 1]++;
            for (int d = 0; d < 2048; d++) cnt[d + 1] += cnt[d];
            for (size_t i = 0; i < n; i++) kb[cnt[(ka[i] >> shift) & 2047ull]++] = ka[i];
            ka.swap(kb);
        }
        for (size_t r = 0; r < n; r++) a[r] = (unsigned)(ka[r] & 0xffffffffull);
        for (size_t r = n; r-- > 0;) {
            const unsigned i = a[r];
            const long long nb = e32[i];
            if (early_big && nb > nb_small) continue;   // (launched already)
            todo[nb <= nb_small ? 0 : (nb <= nb_medium ? 1 : 2)].push_back((int)i);
        }
    }
    // ---- the small-halo class (small_halo.hip): halos of the wavefront class whose sorted superset is expected to fit SMALL_NS
    // slab entries in LDS.  The superset of a halo that is not inside a larger one's envelope holds ~1.5 x its bound
    // particles (C3: 1.39 / 1.52 / 1.69 at the 10th / 50th / 90th percentile); one that does not fit after all comes back
    // with SOAP_HALO_FATAL_SLAB_OVERFLOW and is done by the wavefront class in the next pass.  Runs that need what the class
    // leaves out stay on the general path altogether.
    std::vector<int> todo_small[NSV];
    const long long small_ns[NSV] = {128, 512, 1024};
    bool small_hydro = false;
    {
        bool small_ok = !P.has_nu && !P.literal && !want_property_radius && h->probe_row < 0 && n_calcs <= MAXCALC &&
                        P.profile != 1 && h->cfg.slab_capacity <= 0;
        // chunks with gas / stars / black holes use the variants with the typed passes (k_small128h / k_small512h: per-type moments, five
        // half-mass classes, rotation, projected apertures of up to SMALL_MAXPJ radii); they have no inertia tensor pass
        small_hydro = !P.dm_only;
        if (small_hydro && want_inertia) small_ok = false;
        int n_tensor_calcs = 0, n_projected = 0;   // (the inertia pass of the small-halo kernels holds the states of up to 16 calculations in its 64 lanes)
        for (int c = 0; c < n_calcs; c++) {
            if (calcs[c].kind == SOAP_CALC_PROJECTED) n_projected++;
            if (!small_hydro && (calcs[c].kind == SOAP_CALC_PROJECTED || (calcs[c].flags & SOAP_CALC_FLAG_ROTATION))) small_ok = false;
            if ((calcs[c].flags & SOAP_CALC_FLAG_INERTIA) && calcs[c].kind <= SOAP_CALC_SO_MULTIPLE) n_tensor_calcs++;
        }
        if (n_tensor_calcs > 16 || n_projected > SMALL_MAXPJ) small_ok = false;
        if (small_hydro && getenv("SOAP_ENGINE_SMALL_HYDRO") && atoi(getenv("SOAP_ENGINE_SMALL_HYDRO")) == 0) small_ok = false;
        if (want_inertia && expt_env("SOAP_ENGINE_SMALL_INERTIA") && atoi(expt_env("SOAP_ENGINE_SMALL_INERTIA")) == 0) small_ok = false;
        // SOAP_ENGINE_SMALL: 0 off; 4 (default) EVERY halo of the wavefront class goes through the small-halo kernels -- the
        // 128-entry one where the bound-particle count says so, the 512-entry one otherwise -- and what does not fit is handed on
        // (C3: 17.5 ms of halo kernels against 20.3 with the larger halos on the wavefront class from the start: its 2048
        // persistent waves hold the LDS the small-halo kernels wait for); 1 only the halos that are expected to fit; 2 / 3 test hooks
        int small_mode = 4;
        { const char *e = getenv("SOAP_ENGINE_SMALL"); if (e) small_mode = atoi(e); }
        if (small_ok && small_mode != 0) {
            double small_factor = 1.6;
            { const char *e = expt_env("SOAP_ENGINE_SMALL_FACTOR"); if (e) small_factor = atof(e); }
            const bool use512 = !(expt_env("SOAP_ENGINE_SMALL512") && atoi(expt_env("SOAP_ENGINE_SMALL512")) == 0);
            // The 1024-entry variant (25 KB of LDS and 256 registers per wave) is built and tested but NOT used by default: on C3 it
            // takes 11 560 halos and halves the hand-overs (11 865 -> 6 404, clean-up launch 5.1 -> 3.7 ms), and the halo phase stays at
            // 18.3 - 18.7 ms (rank of 8: 7.3 / 6.0 / 8.0 ms either way) -- the phase ends with the 256-thread class, not with the
            // clean-up.  SOAP_ENGINE_SMALL1024=1 switches it on; it has no tensor pass.
            const bool use1024 = !want_inertia && !small_hydro && expt_env("SOAP_ENGINE_SMALL1024") && atoi(expt_env("SOAP_ENGINE_SMALL1024")) != 0;
            std::vector<int> rest;
            rest.reserve(todo[0].size());
            // A halo that is certain to outgrow the largest LDS slab in use is done by the wavefront class after the small-halo
            // kernels -- one wave per halo, 4e-7 s per candidate: the longest of them are the chains that clean-up launch ends with,
            // alone on the GPU.  Above nb_overflow candidates such a halo goes to the 256-thread class from the start instead.
            long long nb_overflow = 1ll << 62;
            { const char *e = expt_env("SOAP_ENGINE_NB_OVERFLOW"); if (e) nb_overflow = atoll(e); }
            const double small_fit = (double)(use1024 ? small_ns[2] : (use512 ? small_ns[1] : small_ns[0]));
            std::vector<int> to_medium;
            for (int i : todo[0]) {
                const double pred = (double)h->h_nrbound[i] * small_factor + 4.0;
                if (small_mode == 4 && est[i] > nb_overflow && pred > small_fit) { to_medium.push_back(i); continue; }
                int v = pred <= (double)small_ns[0] ? 0 : (pred <= (double)small_ns[1] && use512 ? 1 : (pred <= (double)small_ns[2] && use1024 ? 2 : -1));
                if (small_mode == 2) v = 0;   // (test hooks: every halo of the class tries the smallest slab first / the 512-entry / the 1024-entry variant)
                if (small_mode == 3) v = 1;
#ifdef SOAP_EXPERIMENTS
                if (small_mode == 5 && !want_inertia && !small_hydro) v = 2;
#endif
                if (small_mode == 4 && v < 0) v = use1024 ? 2 : 1;   // (every halo of the class through the small-halo kernels: what does not fit is handed on)
                if (v >= 0) todo_small[v].push_back(i); else rest.push_back(i);
            }
            todo[0].swap(rest);
            if (!to_medium.empty()) {
                // (todo[1] is in descending order of the estimates, all above those of the wavefront class: these follow)
                todo[1].insert(todo[1].end(), to_medium.begin(), to_medium.end());
            }
        }
    }
    // (diagnostic switches are read on every run, like SOAP_ENGINE_SMALL above: a sweep that toggles them inside one process measures what it says)
    const bool small_handover = !(getenv("SOAP_ENGINE_SMALL_HANDOVER") && atoi(getenv("SOAP_ENGINE_SMALL_HANDOVER")) == 0);
    if (small_handover && (!todo_small[0].empty() || !todo_small[1].empty() || !todo_small[2].empty())) {
        // halos that outgrow the small-halo class are handed on inside the pass: list on the device, filled by its kernels
        if (h->small_esc_count < (size_t)h->n_halo + 1) {
            if (h->d_small_esc) (void)hipFree(h->d_small_esc);
            h->d_small_esc = nullptr;
            h->small_esc_count = (size_t)h->n_halo + 1 + (size_t)h->n_halo / 4;
            HIPCHECK(hipMalloc((void **)&h->d_small_esc, h->small_esc_count * sizeof(int)));
        }
        P.small_esc = h->d_small_esc;
        HIPCHECK(hipMemsetAsync(h->d_small_esc, 0, sizeof(int), st));
    }
    if (!todo_small[1].empty() || !todo_small[2].empty() || (!todo_small[0].empty() && soap_small_global_128d())) {
        // the streamed columns of the 512- and 1024-entry variants: a slice per persistent wave, one region per variant
        // (sized for the largest grids launch_small may use, SMALL_MAXG512 / SMALL_MAXG1024 waves; the region of the 1024-entry
        // variant -- 117 MB -- only when that variant has work)
        const size_t need1 = (size_t)SMALL_MAXG512 * 7 * (size_t)small_ns[1] * sizeof(double);
        const size_t need2 = todo_small[2].empty() ? 0 : (size_t)SMALL_MAXG1024 * 7 * (size_t)small_ns[2] * sizeof(double);
        const size_t need0 = (todo_small[0].empty() || !soap_small_global_128d()) ? 0 : (size_t)SMALL_MAXG128 * 7 * (size_t)small_ns[0] * sizeof(double);
        if (h->small_cols_bytes < need1 + need2 + need0) {
            if (h->d_small_cols) (void)hipFree(h->d_small_cols);
            h->d_small_cols = nullptr;
            HIPCHECK(hipMalloc((void **)&h->d_small_cols, need1 + need2 + need0));
            h->small_cols_bytes = need1 + need2 + need0;
        }
        P.small_cols = h->d_small_cols;
        P.small_cols2 = need2 ? (double *)((char *)h->d_small_cols + need1) : nullptr;
        P.small_cols0 = need0 ? (double *)((char *)h->d_small_cols + need1 + need2) : nullptr;
    }
    // the small-halo kernels read their own parameter block: the thread-group classes launched above are reading d_params by now, which
    // is not written again in this run
    if (!todo_small[0].empty() || !todo_small[1].empty() || !todo_small[2].empty())
        HIPCHECK(hipMemcpyAsync(h->d_params_small, &P, sizeof(RunParams), hipMemcpyHostToDevice, st));
    TRACE("run: classified");
    const auto t_host2 = std::chrono::steady_clock::now();
    h->ms_host_setup = std::chrono::duration<double, std::milli>(t_host1 - t_host0).count();
    h->ms_host_pre = std::chrono::duration<double, std::milli>(t_host0b - t_host0).count();
    h->ms_host_classify = std::chrono::duration<double, std::milli>(t_host2 - t_host1).count();
    int groups_small[NSV] = {256 * 11, 256 * 8, 256 * 6};   // (14 KB of LDS per wave: eleven waves per CU; the 512-entry variant holds 236 registers: eight; the 1024-entry one 25 KB of LDS: six)
    if (want_inertia) groups_small[0] = 256 * 8;   // (20 KB of LDS per wave with the tensor pass)
    if (small_hydro) groups_small[0] = groups_small[1] = 256 * 8;   // (256 registers per wave: two waves per SIMD)
    { const char *e = getenv("SOAP_ENGINE_GS"); if (e) { groups_small[0] = groups_small[1] = atoi(e); groups_small[2] = std::min(groups_small[2], atoi(e)); } }
    { const char *e = expt_env("SOAP_ENGINE_GS512"); if (e) groups_small[1] = atoi(e); }
    // every persistent wave of the 512- / 1024-entry variants owns a slice of the streamed-column buffer: never more waves than slices
    { const char *e = expt_env("SOAP_ENGINE_GS128"); if (e) groups_small[0] = atoi(e); }
    groups_small[0] = std::min(std::max(1, groups_small[0]), SMALL_MAXG128);
    groups_small[1] = std::min(std::max(1, groups_small[1]), SMALL_MAXG512);
    groups_small[2] = std::min(std::max(1, groups_small[2]), SMALL_MAXG1024);
    auto launch_small = [&](int v, std::vector<int> &list, hipEvent_t after, bool join) -> int32_t {
        if (list.empty()) return 0;
        soap_engine::Class &C = h->small[v];
        const long long nt = (long long)list.size();
        if (C.todo_cap < nt) {
            if (C.d_todo) (void)hipFree(C.d_todo);
            C.todo_cap = std::max<long long>(nt, h->n_halo);
            HIPCHECK(hipMalloc((void **)&C.d_todo, (size_t)C.todo_cap * 4));
        }
        hipStream_t cs = C.stream;
        HIPCHECK(hipStreamWaitEvent(cs, after, 0));
        HIPCHECK(hipMemcpyAsync(C.d_todo, list.data(), (size_t)nt * 4, hipMemcpyHostToDevice, cs));
        HIPCHECK(hipMemsetAsync(C.d_queue, 0, sizeof(unsigned), cs));
        HIPCHECK(hipEventRecord(C.start, cs));
        C.nh += nt;
        const int ng = (int)std::min<long long>(groups_small[v], nt);
        if (small_hydro) {    // chunks with gas / stars / black holes
            if (v == 0) soap_launch_small_128h(ng, cs, h->d_params_small, C.d_queue, (const int *)C.d_todo, nt);
            else soap_launch_small_512h(ng, cs, h->d_params_small, C.d_queue, (const int *)C.d_todo, nt);
        } else if (want_inertia) {   // the variants with the inertia tensor pass (more LDS per wave: fewer waves per CU)
            if (v == 0) soap_launch_small_128i(ng, cs, h->d_params_small, C.d_queue, (const int *)C.d_todo, nt);
            else soap_launch_small_512i(ng, cs, h->d_params_small, C.d_queue, (const int *)C.d_todo, nt);
        } else if (v == 0) soap_launch_small_128d(ng, cs, h->d_params_small, C.d_queue, (const int *)C.d_todo, nt);
        else if (v == 1) soap_launch_small_512d(ng, cs, h->d_params_small, C.d_queue, (const int *)C.d_todo, nt);
#ifdef SOAP_EXPERIMENTS
        else soap_launch_small_1024d(ng, cs, h->d_params_small, C.d_queue, (const int *)C.d_todo, nt);   // (measured, no gain: DESIGN.md 5.5, 5.13)
#else
        else return fail(-103, "the 1024-entry small-halo kernel is only in builds with -DSOAP_EXPERIMENTS");
#endif
        HIPCHECK(hipEventRecord(C.done, cs));
        if (join) HIPCHECK(hipStreamWaitEvent(st, C.done, 0));
        return 0;
    };
    h->ms_build = h->ms_stage_a = h->ms_stage_b = 0.0;
    h->n_pipeline = 0;
    h->n_gbuild = 0;
    if (trace) { fprintf(stderr, "[soap_engine] ev_p0=%p pipeline=%d st=%p\n", (void *)h->ev_p[0], (int)pipeline, (void *)st); fflush(stderr); }
    if (!early_big) {
        HIPCHECK(hipStreamWaitEvent(st, h->ev_zero, 0));   // (the cleared table)
        HIPCHECK(hipEventRecord(h->ev_p[0], st));
        for (int k = 0; k < 3; k++) { h->cls[k].ms = 0.0; h->cls[k].nh = 0; }
    }
    TRACE("run: ev recorded");
    for (int v = 0; v < NSV; v++) { h->small[v].ms = 0.0; h->small[v].nh = 0; }
    h->n_small_escalated = 0;
    // Hybrid (SOAP_ENGINE_PIPELINE=1): the pipeline takes the halos of the wavefront class -- nine in ten, and what saturates the GPU on
    // the single-kernel path -- while the large halos, whose cost is a long chain in one workgroup each, start at once on
    // the single-kernel path and run beside the pipeline's stages.  SOAP_ENGINE_PIPELINE=2 sends everything down the pipeline.
    const bool hybrid = pipeline && pmode == 1;
    // SOAP_ENGINE_PIPELINE=4: the large halos go to TEAMS of workgroups inside one persistent launch (k_team) instead of the
    // multi-launch pipeline; everything else stays on the single-kernel path
#ifdef SOAP_EXPERIMENTS
    const bool teams = pipeline && pmode == 4;
#else
    const bool teams = false;   // (SOAP_ENGINE_PIPELINE=4, k_team: built, tested, slower than the launches -- only in builds with -DSOAP_EXPERIMENTS)
    if (pipeline && pmode == 4) return fail(-103, "SOAP_ENGINE_PIPELINE=4 (teams) needs a library built with -DSOAP_EXPERIMENTS");
#endif
    if (teams || early_big) pipeline = false;   // (early_big: no halo of this run is the pipeline's -- decided before the large classes were launched)
    bool giants = pipeline && pmode == 3;
    if (giants) {
        // the pipeline works with the resources of the 1024-thread class only: the other two classes are busy with
        // the single-kernel path
        // (not a giant: a halo of a few dozen particles whose first sphere overlaps the mesh cells of a cluster core -- 1e5 candidates to look at,
        // a few hundred to keep -- and for which the radius guess expects no large sphere either.  Streaming its candidates is a millisecond on 256
        // threads; in the pipeline it is a slice of the tile kernels and, when its loop runs past the default first superset, a whole second
        // iteration for one row: rank of 8)
        auto dwarf = [&](int i) { return !(h->h_first[i] > 0.0) && h->h_nrbound[i] * 64 < est[i]; };
        std::vector<int> big;
        for (int k = 0; k < 3; k++) {
            for (int i : todo[k]) ((est[i] > nb_giant && !dwarf(i)) ? big : early[k]).push_back(i);
            todo[k].clear();
        }
        std::sort(big.begin(), big.end(), [&](int x, int y) { return est[x] != est[y] ? est[x] > est[y] : x < y; });
        if (big.empty()) {
            for (int k = 0; k < 3; k++) todo[k].swap(early[k]);
            pipeline = giants = false;
        } else {
            todo[2].swap(big);
            // the class-2 resources belong to the pipeline: its few small fry joins the 256-thread class
            early[1].insert(early[1].end(), early[2].begin(), early[2].end());
            early[2].clear();
            std::sort(early[1].begin(), early[1].end(), [&](int x, int y) { return est[x] != est[y] ? est[x] > est[y] : x < y; });
        }
    }
    if (hybrid) for (int k = 1; k < 3; k++) early[k].swap(todo[k]);
    // behind the small-halo kernels (and behind the wavefront class' own launch: same stream, same slab): the wavefront class once
    // more, for the rows that were handed on
    auto launch_cleanup = [&](bool join) -> int32_t {
        if (P.small_esc == nullptr || (todo_small[0].empty() && todo_small[1].empty() && todo_small[2].empty())) return 0;
        soap_engine::Class &C = h->cls[0];
        hipStream_t cs = h->own_streams ? C.stream : h->cls[2].stream;
        int rc2 = alloc_slab(C, caps[0], groups[0]);
        if (rc2) return rc2;
        for (int v = 0; v < NSV; v++) if (!todo_small[v].empty()) HIPCHECK(hipStreamWaitEvent(cs, h->small[v].done, 0));
        HIPCHECK(hipMemsetAsync(C.d_queue, 0, sizeof(unsigned), cs));
        const int nwg = (int)std::min<long long>(groups[0], (long long)(todo_small[0].size() + todo_small[1].size() + todo_small[2].size()));
        if (C.nh == 0) HIPCHECK(hipEventRecord(C.start, cs));   // (no launch of the class' own in this run: its clock starts here)
        // (the same halos on 256-thread groups -- 512 or 1024 of them on this class' slab -- take 7.5 ms instead of 3.9: measured, round 5)
        if (!want_inertia) (narrow64 ? soap_launch_halo_64n32 : soap_launch_halo_64n)(nwg, cs, h->d_params, C.slab, C.d_queue, (const int *)(h->d_small_esc + 1), -1);
        else (narrow64 ? soap_launch_halo_64i32 : soap_launch_halo_64i)(nwg, cs, h->d_params, C.slab, C.d_queue, (const int *)(h->d_small_esc + 1), -1);
        HIPCHECK(hipEventRecord(C.done, cs));
        if (join) HIPCHECK(hipStreamWaitEvent(st, C.done, 0));
        return 0;
    };
    bool small_early = false;
    if (hybrid || giants) {
        // (measured on a rank of 8 of the C3 box: with the pipeline's first build launches enqueued BEFORE these persistent kernels its
        // set-up kernels no longer wait 0.8 ms for a free CU, but the histogram and gather passes of the build then do -- the verdicts
        // of the first iteration arrive at the same moment either way)
        int rc = launch_single(early, h->ev_p[0], false);
        if (rc) return rc;
        // ... and the small-halo kernels with them: launched after the pipeline's iterations (whose host loop waits for every
        // iteration's verdicts) they and the clean-up launch behind them were a serial tail of ~2.4 ms on an eighth of the C3 box
        const bool small_late = expt_env("SOAP_ENGINE_SMALL_LATE") && atoi(expt_env("SOAP_ENGINE_SMALL_LATE")) != 0;
        if (!small_late && only_class < 0) {
            for (int v = 0; v < NSV && !rc; v++) rc = launch_small(v, todo_small[v], h->ev_p[0], false);
            if (!rc) rc = launch_cleanup(false);
            if (rc) return rc;
            small_early = true;
        }
    }
    if (pipeline) {
        TRACE("pipeline: begin");
        const size_t n = (size_t)h->n_halo;
        if (h->hdr_bytes < n * sizeof(HaloHdr)) {
            if (h->d_hdr) (void)hipFree(h->d_hdr);
            h->d_hdr = nullptr;
            HIPCHECK(hipMalloc((void **)&h->d_hdr, n * sizeof(HaloHdr)));
            h->hdr_bytes = n * sizeof(HaloHdr);
        }
        long long *d_off = nullptr, *d_est2 = nullptr; int *d_cap = nullptr, *d_state = nullptr; double *d_rbuild = nullptr;
        // the scratch goes back to the pool however this block is left (every HIPCHECK below may return); on an error path the class
        // streams are drained first, so that nothing in flight still reads it
        struct PipeScratch {
            soap_engine *h; size_t n; bool ok = false;
            long long *&off, *&est2; int *&cap, *&state; double *&rbuild;
            ~PipeScratch() {
                if (!ok) { (void)hipDeviceSynchronize(); (void)hipGetLastError(); }
                h->pool.release(off, n * 8); h->pool.release(est2, n * 8); h->pool.release(cap, n * 4);
                h->pool.release(state, n * 4); h->pool.release(rbuild, n * 8);
            }
        } pipe_scratch{h, n, false, d_off, d_est2, d_cap, d_state, d_rbuild};
        HIPCHECK(h->pool.alloc((void **)&d_off, n * 8));
        HIPCHECK(h->pool.alloc((void **)&d_est2, n * 8));
        HIPCHECK(h->pool.alloc((void **)&d_cap, n * 4));
        HIPCHECK(h->pool.alloc((void **)&d_state, n * 4));
        HIPCHECK(h->pool.alloc((void **)&d_rbuild, n * 8));
        P.hdr = h->d_hdr;
        P.rebuild_factor = giants ? 1.6 : 1.25;   // (stage A already extrapolates the enclosed density to the target; giants: every extra iteration is a round of launches for a handful of halos)
        { const char *e = expt_env("SOAP_ENGINE_REBUILD_FACTOR"); if (e) P.rebuild_factor = atof(e); }
        P.g_coarse = getenv("SOAP_ENGINE_G_COARSE") != nullptr && atoi(getenv("SOAP_ENGINE_G_COARSE")) != 0;
        HIPCHECK(hipMemcpyAsync(h->d_params_pipe, &P, sizeof(RunParams), hipMemcpyHostToDevice, st));
        auto launch_stage = [&](int mode, std::vector<int> (&lists)[3], hipEvent_t after, bool join, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr, int alt = -1) -> int32_t {
            const int order[3] = {2, 1, 0};
            for (int oi = 0; oi < 3; oi++) {
                const int k = order[oi];
                // alt: the stream, todo list and queue of another class carry this launch of the 1024-thread kernels (giants'
                // mode: the property passes of successive iterations run side by side)
                soap_engine::Class &C = (k == 2 && alt >= 0) ? h->cls[alt] : h->cls[k];
                const bool timed = !(k == 2 && alt >= 0);
                hipStream_t cs = h->own_streams ? C.stream : ((k == 1) ? h->cls[0].stream : h->cls[2].stream);
                if (lists[k].empty()) continue;
                const long long nt = (long long)lists[k].size();
                if (C.todo_cap < nt) {
                    if (C.d_todo) (void)hipFree(C.d_todo);
                    C.todo_cap = std::max<long long>(nt, 2 * h->n_halo);
                    HIPCHECK(hipMalloc((void **)&C.d_todo, (size_t)C.todo_cap * 4));
                }
                HIPCHECK(hipStreamWaitEvent(cs, after, 0));
                HIPCHECK(hipMemcpyAsync(C.d_todo, lists[k].data(), (size_t)nt * 4, hipMemcpyHostToDevice, cs));
                HIPCHECK(hipMemsetAsync(C.d_queue, 0, sizeof(unsigned), cs));
                const int nwg = (int)std::min<long long>(groups[k], nt);
                const RunParams *dp = h->d_params_pipe;
                if (timed) HIPCHECK(hipEventRecord(C.pstart[mode == 1 ? 0 : 1], cs));
                if (e0 && k == 2) HIPCHECK(hipEventRecord(e0, cs));
                if (mode == 1) {
                    if (k == 0) soap_launch_halo_64b(nwg, cs, dp, h->arena, C.d_queue, (const int *)C.d_todo, nt);
                    else if (k == 1) soap_launch_halo_256b(nwg, cs, dp, h->arena, C.d_queue, (const int *)C.d_todo, nt);
                    else soap_launch_halo_1024b(nwg, cs, dp, h->arena, C.d_queue, (const int *)C.d_todo, nt);
                } else {
                    if (k == 0) soap_launch_halo_64s(nwg, cs, dp, h->arena, C.d_queue, (const int *)C.d_todo, nt);
                    else if (k == 1) soap_launch_halo_256s(nwg, cs, dp, h->arena, C.d_queue, (const int *)C.d_todo, nt);
                    else soap_launch_halo_1024s(nwg, cs, dp, h->arena, C.d_queue, (const int *)C.d_todo, nt);
                }
                if (timed) HIPCHECK(hipEventRecord(C.pdone[mode == 1 ? 0 : 1], cs));
                if (e1 && k == 2) HIPCHECK(hipEventRecord(e1, cs));
                HIPCHECK(hipEventRecord(C.done, cs));
                if (join) HIPCHECK(hipStreamWaitEvent(st, C.done, 0));
                if (timed && C.pms[mode == 1 ? 0 : 1] == 0.0) C.pms[mode == 1 ? 0 : 1] = -1.0;   // first iteration: measured after the run
            }
            return 0;
        };
        for (int k = 0; k < 3; k++) h->cls[k].pms[0] = h->cls[k].pms[1] = 0.0;
        // stage times: summed over the iterations (read once the iteration's last kernel is through); per class: first iteration
        auto collect_times = [&]() {
            float a = 0, b = 0, c = 0;
            (void)hipEventElapsedTime(&a, h->ev_p[1], h->ev_p[2]);
            (void)hipEventElapsedTime(&b, h->ev_p[2], h->ev_p[3]);
            (void)hipEventElapsedTime(&c, h->ev_p[3], h->ev_p[4]);
            h->ms_build += a; h->ms_stage_a += b; h->ms_stage_b += c;
            for (int k = 0; k < 3; k++)
                for (int q = 0; q < 2; q++)
                    if (h->cls[k].pms[q] < 0.0) { float m = 0; (void)hipEventElapsedTime(&m, h->cls[k].pstart[q], h->cls[k].pdone[q]); h->cls[k].pms[q] = m > 0 ? m : 1e-6; }
        };
        long long arena_used = 0;
        int props_iters = 0;   // giants' mode: iterations whose property passes were timed with their own events
        std::vector<double> rbuild(h->h_first);     // 0: the default first superset; k_predict's guess otherwise
        std::vector<int> state(n), single[3];       // rows that end up on the single-kernel path
        std::vector<long long> off(n);
        std::vector<int> capv(n);
        // Iterations: every halo goes through build -> radius loop -> property passes; a halo whose loop ran past its
        // superset is built again, further out (twice the radius the loop had reached, the region that is resident
        // permitting), in the next iteration -- the arena is laid out afresh for the halos of each iteration.
        { const char *e = expt_env("SOAP_ENGINE_PIPE_ONLY_CLASS"); if (e) for (int k = 0; k < 3; k++) if (k != atoi(e)) todo[k].clear(); }   // diagnostics
        for (int iter = 0; iter < 8; iter++) {
            if (iter > 0 && expt_env("SOAP_ENGINE_PIPE_ONE_ITER")) { for (int k = 0; k < 3; k++) todo[k].clear(); break; }   // diagnostics
            long long nrows = 0;
            for (int k = 0; k < 3; k++) nrows += (long long)todo[k].size();
            if (nrows == 0) break;
            if (iter > 0) {
                // new work estimates for the enlarged spheres; the classes follow them
                HIPCHECK(hipMemcpyAsync(d_rbuild, rbuild.data(), n * 8, hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(k_estimate, dim3((unsigned)((h->n_halo + BLK / 64 - 1) / (BLK / 64))), dim3(BLK), 0, st, (const RunParams *)h->d_params_pipe, d_est2, (const double *)d_rbuild);
                HIPCHECK(hipMemcpyAsync(est.data(), d_est2, n * 8, hipMemcpyDeviceToHost, st));
                HIPCHECK(hipStreamSynchronize(st));
                std::vector<int> again[3];
                for (int k = 0; k < 3; k++)
                    for (int i : todo[k]) {
                        const long long nb = std::min<long long>(std::max<long long>(est[i], 0), 0xffffffffLL);
                        const int kc = giants ? 2 : (nb <= nb_small ? 0 : (nb <= nb_medium ? 1 : 2));
                        if (hybrid && kc > 0) single[kc].push_back(i);   // the class streams of the large halos are busy
                        else again[kc].push_back(i);
                    }
                for (int k = 0; k < 3; k++) {
                    std::sort(again[k].begin(), again[k].end(), [&](int x, int y) { return est[x] != est[y] ? est[x] > est[y] : x < y; });
                    todo[k].swap(again[k]);
                }
            }
            // arena slices: room for every candidate of the superset (what k_estimate counted), a power of two where
            // the fallback sort may pad, never below 256 entries (stage B parks per-calculation values in slab columns)
            long long total = 0;
            std::fill(capv.begin(), capv.end(), 0);
            for (int k = 0; k < 3; k++)
                for (int i : todo[k]) {
                    long long c = std::max<long long>(est[i], 256);
                    if (c > 1024) c = next_pow2(c);
                    off[i] = total; capv[i] = (int)c; total += c;
                }
            if (giants && iter > 0 && arena_used + total <= h->arena_cap) {
                // fresh space behind the slices of the earlier iterations, whose property passes may still be running
                for (int k = 0; k < 3; k++) for (int i : todo[k]) off[i] += arena_used;
                arena_used += total;
            } else {
                if (giants && iter > 0) { HIPCHECK(hipStreamSynchronize(h->cls[2].stream)); if (h->own_streams) HIPCHECK(hipStreamSynchronize(h->cls[1].stream)); }   // no room: wait, lay out afresh
                int rc0 = alloc_arena(h, giants ? total + total / 2 : total);
                if (rc0) return rc0;
                arena_used = total;
            }
            int rc = 0;
            HIPCHECK(hipMemcpyAsync(d_off, off.data(), n * 8, hipMemcpyHostToDevice, st));
            HIPCHECK(hipMemcpyAsync(d_cap, capv.data(), n * 4, hipMemcpyHostToDevice, st));
            if (iter == 0) HIPCHECK(hipMemcpyAsync(d_rbuild, rbuild.data(), n * 8, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_init_hdr, dim3((unsigned)std::min<size_t>((n + BLK - 1) / BLK, 2048)), dim3(BLK), 0, st, h->d_hdr, (long long)n, d_off, d_cap, (const double *)d_rbuild);
            TRACE("pipeline: records initialised");
            // stage 1: small halos one thread group each; large halos (everything beyond the wavefront class) with
            // several workgroups per halo (k_g* kernels), on the stream of the 1024-thread class
            HIPCHECK(hipEventRecord(h->ev_p[1], st));
            std::vector<int> small_lists[3], none;
            long long iter_tiles = 0;   // tiles of the large halos built in this iteration
            bool gbuild = expt_env("SOAP_ENGINE_GBUILD") == nullptr || atoi(expt_env("SOAP_ENGINE_GBUILD")) != 0;
            if (gbuild) small_lists[0] = todo[0];
            else for (int k = 0; k < 3; k++) small_lists[k] = todo[k];
            if (gbuild && (!todo[1].empty() || !todo[2].empty())) {
                std::vector<int> rows, first_tile, ntile, first_gblock, ngblock, first_sel, nsel, tile_li, gblock_li, sel_li, bin_first, nbinv;
                long long nbins_total = 0;
                for (int k = 1; k < 3; k++) rows.insert(rows.end(), todo[k].begin(), todo[k].end());
                const size_t L = rows.size();
                for (size_t li = 0; li < L; li++) {
                    // the selection holds at most est particles: tiles and bucket groups are sized by that
                    const long long e = std::max<long long>(est[rows[li]], 1);
                    const int nt = (int)((e + GTILE - 1) / GTILE);
                    // beyond a few 1e5 entries the sample of the sample sort (2048 entries for at most GNBMAX buckets: one or two
                    // per bucket) is too thin -- a bucket outgrows its LDS block in one run in ten, which costs a millisecond of
                    // rank sort: uniform bins in radius instead (~128 entries per bin), as many bucket groups as the entries need
                    const long long giant_sort_limit = getenv("SOAP_ENGINE_GIANT_SORT") ? atoll(getenv("SOAP_ENGINE_GIANT_SORT")) : 300000;
                    const bool giant_sort = e > giant_sort_limit;
                    const int ng = giant_sort ? (int)(e / 1024 + 2) : (int)std::min<long long>(GNBMAX, e / 1024 + 2);   // groups hold more than half a 2048-entry block
                    // (bins of ~30 entries on average: a bin may hold up to half a block, and the bins at the core of a concentrated halo --
                    // or across a massive satellite -- hold tens of times the average once the build radius has been doubled twice; with
                    // ~120 per bin two halos of the uncapped C3 box fell back to ONE 1024-thread group each, 15 ms at the end of the pass)
                    const long long giant_bin = expt_env("SOAP_ENGINE_GIANT_BIN") ? atoll(expt_env("SOAP_ENGINE_GIANT_BIN")) : 64;
                    const int nbin = giant_sort ? (int)std::max<long long>(next_pow2(e / giant_bin), 64) : 0;
                    bin_first.push_back(giant_sort ? (int)nbins_total : -1); nbinv.push_back(nbin);
                    if (giant_sort) nbins_total += 2ll * nbin + 1 + ng + 1;
                    first_tile.push_back((int)tile_li.size()); ntile.push_back(nt);
                    first_gblock.push_back((int)gblock_li.size()); ngblock.push_back(ng);
                    const int ns = (int)((e + 16383) / 16384);
                    first_sel.push_back((int)sel_li.size()); nsel.push_back(ns);
                    sel_li.insert(sel_li.end(), (size_t)ns, (int)li);
                    tile_li.insert(tile_li.end(), (size_t)nt, (int)li);
                    gblock_li.insert(gblock_li.end(), (size_t)ng, (int)li);
                }
                if (h->gb_count < L) {
                    if (h->d_gb) (void)hipFree(h->d_gb);
                    h->d_gb = nullptr;
                    h->gb_count = L + L / 4 + 16;
                    HIPCHECK(hipMalloc((void **)&h->d_gb, h->gb_count * sizeof(GBuild)));
                }
                if (nbins_total > 0x7fffffffll) return fail(-102, "giants: too many radial bins for one run");
                if (h->gbins_count < (size_t)nbins_total) {
                    if (h->d_gbins) (void)hipFree(h->d_gbins);
                    h->d_gbins = nullptr;
                    h->gbins_count = (size_t)nbins_total + (size_t)nbins_total / 4 + 64;
                    HIPCHECK(hipMalloc((void **)&h->d_gbins, h->gbins_count * sizeof(int)));
                }
                const size_t nints = 9 * L + tile_li.size() + gblock_li.size() + sel_li.size();
                if (h->gints_count < nints) {
                    if (h->d_gints) (void)hipFree(h->d_gints);
                    h->d_gints = nullptr;
                    h->gints_count = nints + nints / 4 + 64;
                    HIPCHECK(hipMalloc((void **)&h->d_gints, h->gints_count * sizeof(int)));
                }
                soap_engine::Class &C = h->cls[2];
                // (giants' mode: on the main stream, the class stream carries the property passes of the iteration before)
                hipStream_t cs = giants ? st : (h->own_streams ? C.stream : h->cls[2].stream);
                HIPCHECK(hipStreamWaitEvent(cs, h->ev_p[1], 0));
                int *d = h->d_gints;
                const std::vector<int> *parts[12] = {&rows, &first_tile, &ntile, &first_gblock, &ngblock, &tile_li, &gblock_li, &first_sel, &nsel, &sel_li,
                                                     &bin_first, &nbinv};
                int *dptr[12];
                // (ONE upload: twelve copies of a few KB from pageable memory were 0.6 ms of a rank-of-8 pass, each a staging copy and a copy kernel that
                // waits its turn behind the persistent groups)
                std::vector<int> &packed = h->gints_host;
                packed.clear();
                for (int q = 0; q < 12; q++) {
                    dptr[q] = d + packed.size();
                    packed.insert(packed.end(), parts[q]->begin(), parts[q]->end());
                }
                HIPCHECK(hipMemcpyAsync(d, packed.data(), packed.size() * sizeof(int), hipMemcpyHostToDevice, cs));
                P.gb = h->d_gb; P.tile_li = dptr[5]; P.gblock_li = dptr[6]; P.sel_li = dptr[9]; P.gbins = h->d_gbins;
                HIPCHECK(hipMemcpyAsync(h->d_params_pipe, &P, sizeof(RunParams), hipMemcpyHostToDevice, cs));
                HIPCHECK(hipStreamWaitEvent(h->cls[0].stream, h->ev_p[1], 0));   // (class 0 reads the same parameter block: upload first)
                HIPCHECK(hipEventRecord(C.pstart[0], cs));
                soap_launch_gbuild(cs, h->d_params_pipe, h->arena, (long long)L, (long long)tile_li.size(), (long long)gblock_li.size(),
                                   (long long)sel_li.size(), dptr[0], dptr[1], dptr[2], dptr[3], dptr[4], dptr[7], dptr[8], dptr[10], dptr[11]);
                HIPCHECK(hipEventRecord(C.pdone[0], cs));
                HIPCHECK(hipEventRecord(C.done, cs));
                HIPCHECK(hipStreamWaitEvent(st, C.done, 0));
                TRACE("pipeline: tile-parallel build launched");
                if (C.pms[0] == 0.0) C.pms[0] = -1.0;
                h->n_gbuild += (long long)L;
                iter_tiles = (long long)tile_li.size();
            }
            // (the small halos' groups are persistent and would hold all LDS if they started first: the large halos'
            // kernels, which come and go, are enqueued before them)
            rc = launch_stage(1, small_lists, h->ev_p[1], true);
            if (rc) return rc;
            HIPCHECK(hipEventRecord(h->ev_p[2], st));
            // stage 2: one lane per halo (records that are not freshly built are skipped)
            // the one-lane-per-halo radius loop knows nothing of neutrinos, property radii or the tensors' search-radius rule: runs
            // with them never get here (the gate above); should the gate and the kernel ever drift apart, fail instead of computing
            if (P.has_nu || want_inertia || want_property_radius || P.literal) return fail(-102, "pipeline: stage A cannot serve this run (neutrinos / inertia tensors / property radii / literal mode)");
            soap_launch_stage_a(st, h->d_params_pipe, h->arena, nullptr, (long long)n);
            // large halos whose loop has finished: moment sums of the whole tiles of their search spheres, by all CUs, so that
            // the one or two groups of the property stage only add them up (reduce_span)
            const bool use_bpartials = getenv("SOAP_ENGINE_BPARTIALS") == nullptr || atoi(getenv("SOAP_ENGINE_BPARTIALS")) != 0;
            if (use_bpartials && iter_tiles > 0) {
                const size_t need = (size_t)iter_tiles * 2 * (NACC + BSCAN_W);   // tile sums + the tiles' scan records (k_bscan)
                if (h->bpart_count[iter] < need) {
                    if (h->d_bpart[iter]) (void)hipFree(h->d_bpart[iter]);
                    h->d_bpart[iter] = nullptr;
                    h->bpart_count[iter] = need + need / 4;
                    HIPCHECK(hipMalloc((void **)&h->d_bpart[iter], h->bpart_count[iter] * sizeof(double)));
                }
                soap_launch_bpartials(st, h->d_params_pipe, h->arena, iter_tiles, h->d_bpart[iter]);
            }
            hipLaunchKernelGGL(k_collect_state, dim3((unsigned)std::min<size_t>((n + BLK - 1) / BLK, 2048)), dim3(BLK), 0, st, (const RunParams *)h->d_params_pipe, h->d_hdr, (long long)n, d_state, d_rbuild);
            HIPCHECK(hipEventRecord(h->ev_p[3], st));
            HIPCHECK(hipMemcpyAsync(state.data(), d_state, n * 4, hipMemcpyDeviceToHost, st));
            HIPCHECK(hipMemcpyAsync(rbuild.data(), d_rbuild, n * 8, hipMemcpyDeviceToHost, st));
            TRACE("pipeline: radius loop launched, waiting for the verdicts");
            HIPCHECK(hipStreamSynchronize(st));
            TRACE("pipeline: verdicts read");
            {
                hipError_t le = hipGetLastError();
                if (le != hipSuccess) return fail(-101, std::string("pipeline: ") + hipGetErrorString(le));
            }
            if (trace) {
                // why a halo leaves the pipeline: the scratch record of its tile-parallel build
                for (int k = 0; k < 3; k++)
                    for (int i : todo[k])
                        if (state[i] == PIPE_FALLBACK) {
                            fprintf(stderr, "[soap_engine] pipeline iteration %d: row %d (estimate %lld) falls back to the single-kernel path", iter, i, est[i]);
                            if (h->d_gb && iter_tiles > 0) {
                                std::vector<GBuild> gb((size_t)std::min<size_t>(h->gb_count, (size_t)nrows));
                                (void)hipMemcpy(gb.data(), h->d_gb, gb.size() * sizeof(GBuild), hipMemcpyDeviceToHost);
                                for (const GBuild &g : gb)
                                    if (g.row == i) fprintf(stderr, ": selected %u overflow %d unsorted %d bound %d+%d+%d+%d", g.n, g.overflow, g.q_unsorted, g.nbi[0], g.nbi[1], g.nbi[2], g.nbi[3]);
                            }
                            fprintf(stderr, "\n");
                        }
            }
            if (trace) {
                // who is built again, and how far the guess was off
                std::vector<double> h_search0(n), h_read0(n);
                std::vector<int> h_central(n);
                (void)hipMemcpy(h_search0.data(), h->d_search, n * 8, hipMemcpyDeviceToHost);
                (void)hipMemcpy(h_read0.data(), h->d_read, n * 8, hipMemcpyDeviceToHost);
                (void)hipMemcpy(h_central.data(), h->d_central, n * 4, hipMemcpyDeviceToHost);
                for (int k = 0; k < 3; k++)
                    for (int i : todo[k])
                        if (state[i] == PIPE_NEED_SUPERSET)
                            fprintf(stderr, "[soap_engine] pipeline iteration %d: row %d (estimate %lld, bound %lld, guess %.4g x search radius) is built again out to %.4g x (central %d, read radius %.4g x, search radius %.4g)\n", iter, i, est[i],
                                    h->h_nrbound[i], h->h_first[i] > 0.0 ? h->h_first[i] / h_search0[i] : 0.0, rbuild[i] / h_search0[i], h_central[i], h_read0[i] / h_search0[i], h_search0[i]);
            }
            // stage 3 for the halos whose loop finished
            std::vector<int> fin[3], rest[3];
            for (int k = 0; k < 3; k++)
                for (int i : todo[k]) {
                    if (state[i] == PIPE_DONE) fin[k].push_back(i);
                    else if (state[i] == PIPE_NEED_SUPERSET && iter < 7) rest[k].push_back(i);
                    else single[k].push_back(i);
                }
            // large halos of a run without gas / stars: two groups each, one per family of calculations (stage_b)
            std::vector<int> fin2[3];
            // (projected apertures are a pass of the undivided halo: stage_b runs them with part == 0 only)
            bool any_projected = false;
            for (int c = 0; c < n_calcs; c++) any_projected = any_projected || calcs[c].kind == SOAP_CALC_PROJECTED;
            const bool split = !P.hydro && !any_projected && (expt_env("SOAP_ENGINE_SPLIT_PROPS") == nullptr || atoi(expt_env("SOAP_ENGINE_SPLIT_PROPS")) != 0);
            for (int k = 0; k < 3; k++)
                for (int i : fin[k]) {
                    if (split && k > 0) { fin2[k].push_back(i | (1 << 29)); fin2[k].push_back(i | (2 << 29)); }
                    else fin2[k].push_back(i);
                }
            if (giants && expt_env("SOAP_ENGINE_GIANT_PROPS_CLASS") && atoi(expt_env("SOAP_ENGINE_GIANT_PROPS_CLASS")) == 1) {
                // experiment: the 256-thread variant of the property kernel (256 VGPRs, no spills in the sweeps)
                fin2[1].insert(fin2[1].end(), fin2[2].begin(), fin2[2].end());
                fin2[2].clear();
            }
            // giants' mode: the property passes run on the 1024-thread class' stream while the main stream goes on with the
            // next iteration (rebuilds of the few halos whose loop ran past their superset, in fresh arena space)
            if (giants && !fin2[2].empty()) {
                for (int q = 0; q < 2; q++) if (!h->ev_b[iter][q]) HIPCHECK(hipEventCreate(&h->ev_b[iter][q]));
                props_iters |= 1 << iter;
            }
            rc = launch_stage(2, fin2, h->ev_p[3], !giants, giants ? h->ev_b[iter][0] : nullptr, giants ? h->ev_b[iter][1] : nullptr,
                              (giants && h->own_streams && (iter & 1)) ? 1 : -1);
            if (rc) return rc;
            HIPCHECK(hipEventRecord(h->ev_p[4], st));
            for (int k = 0; k < 3; k++) { h->n_pipeline += (long long)fin[k].size(); h->cls[k].nh += (long long)fin[k].size(); todo[k].swap(rest[k]); }
            // the arena is laid out again for the next iteration: its property passes have to be through
            bool more = false;
            for (int k = 0; k < 3; k++) more = more || !todo[k].empty();
            HIPCHECK(hipStreamSynchronize(st));
            collect_times();
            (void)more;
            if (trace) {
                fprintf(stderr, "[soap_engine] pipeline iteration %d: %lld rows, finished %zu, again %zu, build %.2f ms a %.2f ms props %.2f ms (cumulative)\n", iter, nrows,
                        fin[0].size() + fin[1].size() + fin[2].size(), todo[0].size() + todo[1].size() + todo[2].size(), h->ms_build, h->ms_stage_a, h->ms_stage_b);
                fflush(stderr);
            }
        }
        if (giants) {
            HIPCHECK(hipStreamWaitEvent(st, h->cls[2].done, 0));
            HIPCHECK(hipStreamSynchronize(h->cls[2].stream));   // (the scratch below is reused by the next call)
            if (h->own_streams) HIPCHECK(hipStreamSynchronize(h->cls[1].stream));   // odd iterations' property passes
            h->ms_stage_b = 0.0;   // the passes ran beside the later iterations: their own events
            for (int it = 0; it < 8; it++)
                if (props_iters & (1 << it)) { float m = 0; (void)hipEventElapsedTime(&m, h->ev_b[it][0], h->ev_b[it][1]); h->ms_stage_b += m; }
        }
        for (int k = 0; k < 3; k++) { single[k].insert(single[k].end(), todo[k].begin(), todo[k].end()); todo[k].swap(single[k]); }
        pipe_scratch.ok = true;   // (released by the guard)
    }
    // ---- teams: the halos beyond nb_team candidates, heaviest first, T teams of K workgroups
    std::vector<int> team_rows;
    long long *d_toff = nullptr; int *d_tcap = nullptr, *d_tstate = nullptr; double *d_trb = nullptr;
    struct TeamScratch {
        soap_engine *h; size_t n, L; long long *&off; int *&cap, *&state; double *&rb;
        ~TeamScratch() { h->pool.release(off, n * 8); h->pool.release(cap, n * 4); h->pool.release(state, L * 4); h->pool.release(rb, n * 8); }
    } team_scratch{h, (size_t)h->n_halo, 0, d_toff, d_tcap, d_tstate, d_trb};
    h->n_team = h->n_team_failed = 0;
    h->ms_team = 0.0;
    if (teams) {
        long long nb_team = 49152;
        { const char *e = getenv("SOAP_ENGINE_NB_TEAM"); if (e) nb_team = atoll(e); }
        for (int k = 1; k < 3; k++) {
            std::vector<int> keep;
            for (int i : todo[k]) (est[i] > nb_team ? team_rows : keep).push_back(i);
            todo[k].swap(keep);
        }
        std::sort(team_rows.begin(), team_rows.end(), [&](int x, int y) { return est[x] != est[y] ? est[x] > est[y] : x < y; });
    }
    if (!team_rows.empty()) {
        const size_t n = (size_t)h->n_halo, L = team_rows.size();
        team_scratch.L = L;
        // workgroups of the launch (one per CU: the kernel's LDS): all of them walk every phase together
        int T = 128;
        { const char *e = getenv("SOAP_ENGINE_TEAMS"); if (e) T = atoi(e); }
        T = std::max(1, std::min(T, 256));
        // every group of the team spins at the team barrier until all have arrived: never more groups than can be resident together
        // (the kernel's LDS allows one per CU; a partitioned device has fewer CUs), less a margin for the persistent groups of the
        // other classes that may hold a CU when the team starts
#ifdef SOAP_EXPERIMENTS
        { const int fit = soap_team_max_groups(); if (fit > 0) T = std::max(1, std::min(T, fit - fit / 8)); }
#endif
        if (h->hdr_bytes < n * sizeof(HaloHdr)) {
            if (h->d_hdr) (void)hipFree(h->d_hdr);
            h->d_hdr = nullptr;
            HIPCHECK(hipMalloc((void **)&h->d_hdr, n * sizeof(HaloHdr)));
            h->hdr_bytes = n * sizeof(HaloHdr);
        }
        std::vector<long long> off(n, 0);
        std::vector<int> capv(n, 0);
        std::vector<int> rows, first_tile, ntile, first_gblock, ngblock, first_sel, nsel, tile_li, gblock_li, sel_li, bin_first, nbinv;
        long long total = 0, nbins_total = 0;
        const long long giant_sort_limit = getenv("SOAP_ENGINE_GIANT_SORT") ? atoll(getenv("SOAP_ENGINE_GIANT_SORT")) : 300000;
        const long long giant_bin = expt_env("SOAP_ENGINE_GIANT_BIN") ? atoll(expt_env("SOAP_ENGINE_GIANT_BIN")) : 64;
        for (size_t li = 0; li < L; li++) {
            const int i = team_rows[li];
            // room for the sphere the guess reaches and for one rebuild further out (the team checks before it builds again)
            long long c = next_pow2(std::max<long long>(2 * est[i], 4096));
            c = std::min<long long>(c, cap_max);
            off[i] = total; capv[i] = (int)c; total += c;
            const bool giant_sort = est[i] > giant_sort_limit;
            const int nt = (int)((c + GTILE - 1) / GTILE);
            const int ng = giant_sort ? (int)(c / 1024 + 2) : (int)std::min<long long>(GNBMAX, c / 1024 + 2);
            const int nbin = giant_sort ? (int)std::max<long long>(next_pow2(c / giant_bin), 64) : 0;
            const int ns = (int)((c + 16383) / 16384);
            rows.push_back(i);
            bin_first.push_back(giant_sort ? (int)nbins_total : -1); nbinv.push_back(nbin);
            if (giant_sort) nbins_total += 2ll * nbin + 1 + ng + 1;
            first_tile.push_back((int)tile_li.size()); ntile.push_back(nt);
            first_gblock.push_back((int)gblock_li.size()); ngblock.push_back(ng);
            first_sel.push_back((int)sel_li.size()); nsel.push_back(ns);
            sel_li.insert(sel_li.end(), (size_t)ns, (int)li);
            tile_li.insert(tile_li.end(), (size_t)nt, (int)li);
            gblock_li.insert(gblock_li.end(), (size_t)ng, (int)li);
        }
        { int rc0 = alloc_arena(h, total); if (rc0) return rc0; }
        if (h->gb_count < L) {
            if (h->d_gb) (void)hipFree(h->d_gb);
            h->d_gb = nullptr;
            h->gb_count = L + L / 4 + 16;
            HIPCHECK(hipMalloc((void **)&h->d_gb, h->gb_count * sizeof(GBuild)));
        }
        if (nbins_total > 0x7fffffffll) return fail(-102, "teams: too many radial bins for one run");
        if (h->gbins_count < (size_t)nbins_total) {
            if (h->d_gbins) (void)hipFree(h->d_gbins);
            h->d_gbins = nullptr;
            h->gbins_count = (size_t)nbins_total + (size_t)nbins_total / 4 + 64;
            HIPCHECK(hipMalloc((void **)&h->d_gbins, h->gbins_count * sizeof(int)));
        }
        const std::vector<int> *parts[12] = {&rows, &first_tile, &ntile, &first_gblock, &ngblock, &tile_li, &gblock_li, &first_sel, &nsel, &sel_li, &bin_first, &nbinv};
        size_t nints = 0;
        for (int q = 0; q < 12; q++) nints += parts[q]->size();
        if (h->gints_count < nints) {
            if (h->d_gints) (void)hipFree(h->d_gints);
            h->d_gints = nullptr;
            h->gints_count = nints + nints / 4 + 64;
            HIPCHECK(hipMalloc((void **)&h->d_gints, h->gints_count * sizeof(int)));
        }
        const size_t ntiles_total = tile_li.size();
        const size_t need_bp = ntiles_total * 2 * (NACC + BSCAN_W);
        if (h->bpart_count[0] < need_bp) {
            if (h->d_bpart[0]) (void)hipFree(h->d_bpart[0]);
            h->d_bpart[0] = nullptr;
            h->bpart_count[0] = need_bp + need_bp / 4;
            HIPCHECK(hipMalloc((void **)&h->d_bpart[0], h->bpart_count[0] * sizeof(double)));
        }
        HIPCHECK(h->pool.alloc((void **)&d_toff, n * 8));
        HIPCHECK(h->pool.alloc((void **)&d_tcap, n * 4));
        HIPCHECK(h->pool.alloc((void **)&d_tstate, L * 4));
        HIPCHECK(h->pool.alloc((void **)&d_trb, n * 8));
        hipStream_t ts = h->team_stream;
        HIPCHECK(hipStreamWaitEvent(ts, h->ev_p[0], 0));
        int *d = h->d_gints;
        int *dptr[12];
        for (int q = 0; q < 12; q++) {
            dptr[q] = d;
            HIPCHECK(hipMemcpyAsync(d, parts[q]->data(), parts[q]->size() * sizeof(int), hipMemcpyHostToDevice, ts));
            d += parts[q]->size();
        }
        HIPCHECK(hipMemcpyAsync(d_toff, off.data(), n * 8, hipMemcpyHostToDevice, ts));
        HIPCHECK(hipMemcpyAsync(d_tcap, capv.data(), n * 4, hipMemcpyHostToDevice, ts));
        HIPCHECK(hipMemcpyAsync(d_trb, h->h_first.data(), n * 8, hipMemcpyHostToDevice, ts));
        RunParams PT = P;
        PT.hdr = h->d_hdr; PT.gb = h->d_gb; PT.tile_li = dptr[5]; PT.gblock_li = dptr[6]; PT.sel_li = dptr[9]; PT.gbins = h->d_gbins;
        PT.rebuild_factor = 1.6;
        { const char *e = expt_env("SOAP_ENGINE_REBUILD_FACTOR"); if (e) PT.rebuild_factor = atof(e); }
        PT.g_coarse = getenv("SOAP_ENGINE_G_COARSE") != nullptr && atoi(getenv("SOAP_ENGINE_G_COARSE")) != 0;
        HIPCHECK(hipMemcpyAsync(h->d_params_pipe, &PT, sizeof(RunParams), hipMemcpyHostToDevice, ts));
        HIPCHECK(hipMemsetAsync(h->d_teamctl, 0, sizeof(TeamCtl), ts));
        hipLaunchKernelGGL(k_init_hdr, dim3((unsigned)std::min<size_t>((n + BLK - 1) / BLK, 2048)), dim3(BLK), 0, ts, h->d_hdr, (long long)n, d_toff, d_tcap, (const double *)d_trb);
        bool any_projected = false;
        for (int c = 0; c < n_calcs; c++) any_projected = any_projected || calcs[c].kind == SOAP_CALC_PROJECTED;
        const int split = (!P.hydro && !any_projected && !(expt_env("SOAP_ENGINE_SPLIT_PROPS") && atoi(expt_env("SOAP_ENGINE_SPLIT_PROPS")) == 0)) ? 1 : 0;
        const bool use_bpartials = getenv("SOAP_ENGINE_BPARTIALS") == nullptr || atoi(getenv("SOAP_ENGINE_BPARTIALS")) != 0;
        double *bpart = use_bpartials ? h->d_bpart[0] : nullptr;
        HIPCHECK(hipEventRecord(h->team_start, ts));
#ifdef SOAP_EXPERIMENTS
        soap_launch_team(ts, h->d_params_pipe, h->arena, h->d_teamctl, (long long)L, T, split, (long long)tile_li.size(), (long long)gblock_li.size(), (long long)sel_li.size(),
                         dptr[0], dptr[1], dptr[2], dptr[3], dptr[4], dptr[7], dptr[8], dptr[10], dptr[11], bpart, bpart ? bpart + ntiles_total * 2 * NACC : nullptr);
#endif
        hipLaunchKernelGGL(k_team_states, dim3((unsigned)std::min<size_t>((L + BLK - 1) / BLK, 256)), dim3(BLK), 0, ts, (const HaloHdr *)h->d_hdr, (const int *)dptr[0], (long long)L, d_tstate);
        HIPCHECK(hipEventRecord(h->team_done, ts));
        HIPCHECK(hipGetLastError());
        h->n_team = (long long)L;
        if (trace) { fprintf(stderr, "[soap_engine] teams: %zu halos above %lld candidates (largest %lld), %d workgroups, arena %lld entries, %zu tiles\n", L, (long long)est[team_rows.back()], (long long)est[team_rows[0]], T, total, ntiles_total); fflush(stderr); }
    }
    TRACE("run: single-kernel part");
    h->n_single = (long long)(todo[0].size() + todo[1].size() + todo[2].size() + early[0].size() + early[1].size() + early[2].size() + todo_small[0].size() + todo_small[1].size() + todo_small[2].size());
    std::vector<int> status(h->n_halo);
    double ms_total = 0.0;
    h->n_passes = 0;
    {
        // a sphere that is known not to fit the largest class' slab: grow the slab before the first attempt
        // instead of finding out by overflowing (the estimate counts every candidate of the first search sphere)
        long long need = 0;
        for (int i : todo[2]) need = std::max(need, est[i]);
        // (the radius loop may take the sphere a few x1.2 rounds further, x1.7 in content each: room for three of them --
        // a halo that overflows is done again in a pass of its own, which doubles the time of the largest halo)
        long long n_top = 0;
        for (int i : todo[2]) n_top += split_large ? 2 : 1;
        size_top_split(n_top, need);
        while (caps[2] < cap_max && caps[2] < 4 * need && groups[2] / 2 >= std::max<long long>(16, std::min<long long>(n_top, split_large ? 256 : 64))) {
            caps[2] = std::min(caps[2] * 2, cap_max);
            groups[2] = std::max(1, groups[2] / 2);
        }
        while (caps[2] < cap_max && caps[2] < need + need / 4) {
            caps[2] = std::min(caps[2] * 2, cap_max);
            groups[2] = std::max(1, groups[2] / 2);
        }
    }
    for (int pass = 0; pass < 12; pass++) {
        bool any = false;
        HIPCHECK(hipEventRecord(h->ev0, st));
        if (only_class >= 0) { for (int k = 0; k < 3; k++) if (k != only_class) todo[k].clear(); if (only_class != 3) { todo_small[0].clear(); todo_small[1].clear(); todo_small[2].clear(); } }   // diagnostics (3: the small-halo class)
        for (int k = 0; k < 3; k++) any = any || !todo[k].empty();
        {
            // the thread-group classes first, largest groups first: a 1024-thread group needs an empty CU, and the waves of the
            // small-halo class (three per SIMD fill its registers) would never leave one
            const bool small_first = expt_env("SOAP_ENGINE_SMALL_FIRST") && atoi(expt_env("SOAP_ENGINE_SMALL_FIRST")) != 0;
            int rc = 0;
            const bool launch_small_now = !(small_early && pass == 0);   // (giants' mode: they started before the pipeline)
            if (small_first && launch_small_now) for (int v = 0; v < NSV && !rc; v++) rc = launch_small(v, todo_small[v], h->ev0, true);
            if (rc) return rc;
            rc = launch_single(todo, h->ev0, true);
            if (rc) return rc;
            if (!small_first && launch_small_now) for (int v = 0; v < NSV && !rc; v++) rc = launch_small(v, todo_small[v], h->ev0, true);
            if (rc) return rc;
            if (!launch_small_now) for (int v = 0; v < NSV; v++) if (!todo_small[v].empty()) HIPCHECK(hipStreamWaitEvent(st, h->small[v].done, 0));
            for (int v = 0; v < NSV; v++) any = any || !todo_small[v].empty();
            if (launch_small_now) { const int rc3 = launch_cleanup(true); if (rc3) return rc3; }
            else if (P.small_esc != nullptr && (!todo_small[0].empty() || !todo_small[1].empty() || !todo_small[2].empty())) HIPCHECK(hipStreamWaitEvent(st, h->cls[0].done, 0));
        }
        if (pass == 0) {
            // the large halos launched before the pipeline stages: their statuses are checked with this pass
            for (int k = 0; k < 3; k++) {
                if (early[k].empty()) continue;
                any = true;
                todo[k].insert(todo[k].end(), early[k].begin(), early[k].end());
                HIPCHECK(hipStreamWaitEvent(st, h->cls[k].done, 0));
            }
        }
        if (pass == 0 && !team_rows.empty()) { any = true; HIPCHECK(hipStreamWaitEvent(st, h->team_done, 0)); }
        if (!any) break;
        h->n_passes++;
        HIPCHECK(hipEventRecord(h->ev1, st));
        HIPCHECK(hipMemcpyAsync(status.data(), h->d_status, (size_t)h->n_halo * 4, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        for (int k = 0; k < 3; k++) HIPCHECK(hipStreamSynchronize(h->cls[k].stream));
        for (int v = 0; v < NSV; v++) HIPCHECK(hipStreamSynchronize(h->small[v].stream));
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) return fail(-101, std::string("halo kernel: ") + hipGetErrorString(le));
        float ms = 0;
        (void)hipEventElapsedTime(&ms, h->ev0, h->ev1);
        ms_total += ms;
        for (int k = 0; k < 3; k++)
            if (!todo[k].empty()) { float m2 = 0; (void)hipEventElapsedTime(&m2, h->cls[k].start, h->cls[k].done); h->cls[k].ms += m2; }
        // escalate halos whose search sphere did not fit their class' slab
        std::vector<int> next[3];
        bool grow_top = false;
        if (pass == 0 && !team_rows.empty()) {
            // what the teams left undone (radius ties, bound counts that disagree with the catalogue, a sphere beyond the arena
            // slice) goes to the single-kernel path in the next pass
            HIPCHECK(hipStreamSynchronize(h->team_stream));
            std::vector<int> tstate(team_rows.size());
            unsigned ctl_head[2] = {0u, 0u};
            HIPCHECK(hipMemcpy(tstate.data(), d_tstate, team_rows.size() * 4, hipMemcpyDeviceToHost));
            HIPCHECK(hipMemcpy(ctl_head, h->d_teamctl, sizeof(ctl_head), hipMemcpyDeviceToHost));
            if (ctl_head[1] != 0u) return fail(-103, "team kernel: a team barrier timed out");
            if (trace) {
                TeamCtl hc;
                HIPCHECK(hipMemcpy(&hc, h->d_teamctl, sizeof(TeamCtl), hipMemcpyDeviceToHost));
                static const char *nm[15] = {"queue", "select", "split", "hist", "scan", "scatter", "sort", "gather", "cumsum", "post", "final+stage_a", "rebuild/idle", "tile sums", "tile scans", "props"};
                unsigned long long tot = 0; for (int k = 0; k < 15; k++) tot += hc.phase[k];
                fprintf(stderr, "[soap_engine] team phases (member 0, share of its clocks):");
                for (int k = 0; k < 15; k++) fprintf(stderr, " %s %.1f%%", nm[k], 100.0 * (double)hc.phase[k] / (double)std::max<unsigned long long>(tot, 1));
                fprintf(stderr, "\n"); fflush(stderr);
            }
            float m2 = 0;
            (void)hipEventElapsedTime(&m2, h->team_start, h->team_done);
            h->ms_team = m2;
            for (size_t li = 0; li < team_rows.size(); li++)
                if (tstate[li] != PIPE_DONE) {
                    const int i = team_rows[li];
                    next[est[i] <= nb_medium ? 1 : 2].push_back(i);
                    h->n_team_failed++;
                    if (trace) { fprintf(stderr, "[soap_engine] teams: row %d (estimate %lld) left in state %d: single-kernel path\n", i, (long long)est[i], tstate[li]); fflush(stderr); }
                }
        }
        if (pass == 0 && P.small_esc != nullptr && (!todo_small[0].empty() || !todo_small[1].empty() || !todo_small[2].empty())) {
            int nesc = 0;
            HIPCHECK(hipMemcpy(&nesc, h->d_small_esc, sizeof(int), hipMemcpyDeviceToHost));
            h->n_small_escalated += nesc;
            h->cls[0].nh += nesc;
            if (trace) {
                std::vector<int> rows((size_t)std::max(nesc, 0));
                if (nesc > 0) (void)hipMemcpy(rows.data(), h->d_small_esc + 1, (size_t)nesc * sizeof(int), hipMemcpyDeviceToHost);
                fprintf(stderr, "[soap_engine] small-halo class handed on %d rows:", nesc);
                for (int q = 0; q < std::min(nesc, 24); q++) fprintf(stderr, " %d(nb %lld)", rows[q], rows[q] >= 0 && rows[q] < h->n_halo ? h->h_nrbound[rows[q]] : -1ll);
                {
                    std::vector<int> srt(rows); std::sort(srt.begin(), srt.end());
                    int dup = 0; for (size_t q = 1; q < srt.size(); q++) dup += srt[q] == srt[q - 1];
                    std::vector<char> in0(h->n_halo, 0), in1(h->n_halo, 0), inc(h->n_halo, 0);
                    for (int i : todo_small[0]) in0[i] = 1; for (int i : todo_small[1]) in1[i] = 1; for (int i : todo[0]) inc[i] = 1;
                    int c0 = 0, c1 = 0, cc = 0, other = 0;
                    for (int r : rows) { if (r < 0 || r >= h->n_halo) { other++; continue; } c0 += in0[r]; c1 += in1[r]; cc += inc[r]; if (!in0[r] && !in1[r] && !inc[r]) other++; }
                    fprintf(stderr, "\n[soap_engine]   duplicates %d; of the 128-entry list %d (of %zu), of the 512-entry list %d (of %zu), of the wavefront class' own list %d (of %zu), other %d", dup, c0, todo_small[0].size(), c1, todo_small[1].size(), cc, todo[0].size(), other);
                }
                fprintf(stderr, "\n");
            }
            float m2 = 0;
            if (todo[0].empty()) { (void)hipEventElapsedTime(&m2, h->cls[0].start, h->cls[0].done); h->cls[0].ms += m2; }
        }
        for (int v = 0; v < NSV; v++) {
            if (todo_small[v].empty()) continue;
            float m2 = 0;
            (void)hipEventElapsedTime(&m2, h->small[v].start, h->small[v].done);
            h->small[v].ms += m2;
            // (handed on inside the pass: a row that still overflows has outgrown the wavefront class' slab too)
            std::vector<int> &nx = P.small_esc != nullptr ? next[1] : next[0];
            const size_t before = nx.size();
            for (int i : todo_small[v])
                if (status[i] == SOAP_HALO_FATAL_SLAB_OVERFLOW) nx.push_back(i);
            if (P.small_esc == nullptr) h->n_small_escalated += (long long)(nx.size() - before);
            todo_small[v].clear();
        }
        for (int k = 0; k < 3; k++)
            for (int i : todo[k])
                if (status[i] == SOAP_HALO_FATAL_SLAB_OVERFLOW) {
                    if (k < 2) next[k + 1].push_back(i);
                    else { next[2].push_back(i); grow_top = true; }
                }
        if (trace && (next[1].size() || next[2].size())) {
            long long emax = 0;
            for (int k = 1; k < 3; k++) for (int i : next[k]) emax = std::max(emax, est[i]);
            fprintf(stderr, "[soap_engine] pass %d: %zu halos outgrew the wavefront / 256-thread slabs, %zu the 1024-thread slab (largest estimate %lld; slabs %lld / %lld / %lld entries)\n",
                    pass, next[1].size() + (grow_top ? 0 : next[2].size()), grow_top ? next[2].size() : (size_t)0, emax, caps[0], caps[1], caps[2]);
            fflush(stderr);
        }
        if (grow_top) {
            if (caps[2] >= cap_max) break;   // cannot grow further: the status stays fatal
            caps[2] = std::min(caps[2] * 8, cap_max);
            groups[2] = std::max(1, groups[2] / 8);
        }
        for (int k = 0; k < 3; k++) todo[k].swap(next[k]);
    }
    (void)ms_total;
    TRACE("run: passes done");
    {
        // the halo phase on the device timeline: from the first launch of the run to the completion of the last kernel
        // (host gaps between the stages and passes included)
        for (int k = 0; k < 3; k++) {
            HIPCHECK(hipEventRecord(h->cls[k].done, h->cls[k].stream));
            HIPCHECK(hipStreamWaitEvent(st, h->cls[k].done, 0));
        }
        for (int v = 0; v < NSV; v++) {
            HIPCHECK(hipEventRecord(h->small[v].done, h->small[v].stream));
            HIPCHECK(hipStreamWaitEvent(st, h->small[v].done, 0));
        }
        HIPCHECK(hipEventRecord(h->ev1, st));
        HIPCHECK(hipStreamSynchronize(st));
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) return fail(-101, std::string("halo kernel: ") + hipGetErrorString(le));
        float ms = 0;
        (void)hipEventElapsedTime(&ms, h->ev_p[0], h->ev1);
        h->ms_halo = ms;

    }
    h->ms_host_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
    HIPCHECK(hipMemcpy(h->phase, h->d_phase, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    (void)hipGetLastError();   // (the elapsed-time queries above report events a class without work never recorded: not an error of the run)
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

// one workgroup per output row: the row's stored columns, scaled and cast (soap_engine_pack_catalogue)
__global__ void __launch_bounds__(BLK) k_pack_catalogue(const double *table, long long ncols, const long long *rows, long long n_rows,
                                                        const int *ncol, const double *nscale, int nn, float *out_n,
                                                        const int *wcol, const double *wscale, int nw, double *out_w) {
    for (long long r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const double *src = table + (rows ? rows[r] : r) * ncols;
        for (int k = threadIdx.x; k < nn; k += BLK) out_n[r * nn + k] = (float)(src[ncol[k]] * nscale[k]);
        for (int k = threadIdx.x; k < nw; k += BLK) out_w[r * nw + k] = src[wcol[k]] * wscale[k];
    }
}

int32_t soap_engine_pack_catalogue(soap_engine *h, const int64_t *rows, int64_t n_rows,
                                   const int32_t *narrow_cols, const double *narrow_scale, int32_t n_narrow, float *out_narrow,
                                   const int32_t *wide_cols, const double *wide_scale, int32_t n_wide, double *out_wide) {
    if (!h) return fail(-1, "null handle");
    if (!h->d_table) return fail(-2, "no run to pack");
    if (n_rows < 0 || n_rows > h->n_halo || n_narrow < 0 || n_wide < 0) return fail(-3, "pack_catalogue: bad sizes");
    if ((n_narrow > 0 && (!narrow_cols || !narrow_scale || !out_narrow)) || (n_wide > 0 && (!wide_cols || !wide_scale || !out_wide)))
        return fail(-3, "pack_catalogue: null argument");
    if (n_rows == 0) return 0;
    HIPCHECK(hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(k_pack_catalogue, dim3((unsigned)std::min<long long>(n_rows, 256 * 16)), dim3(BLK), 0, h->stream, (const double *)h->d_table,
                       (long long)h->ncols, (const long long *)rows, (long long)n_rows, (const int *)narrow_cols, narrow_scale, (int)n_narrow, out_narrow,
                       (const int *)wide_cols, wide_scale, (int)n_wide, out_wide);
    HIPCHECK(hipGetLastError());
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
    if (stats) HIPCHECK(hipMemcpyAsync(stats, h->d_stats, n * 8 * SOAP_STATS_COLUMNS, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    return 0;
}

int64_t soap_engine_membership(soap_engine *h, int64_t row, int32_t ts, double radius,
                               int32_t strict, int32_t bound_only, int64_t *out, int64_t cap) {
    if (!h) return fail(-1, "null handle");
    if (row < 0 || row >= h->n_halo) return fail(-2, "row out of range");
    if (ts < 0 || ts >= SOAP_NTYPES) return fail(-3, "type_slot out of range");
    HIPCHECK(hipSetDevice(h->cfg.device));
    { const int32_t frc = finish_stage(h); if (frc) return frc; }
    const TypeHost &T = h->t[ts];
    if (T.n == 0) return 0;
    double c[3];
    long long hindex;
    HIPCHECK(hipMemcpy(c, h->d_cofp + 3 * row, 24, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(&hindex, h->d_index + row, 8, hipMemcpyDeviceToHost));
    TypeDev D;
    memset(&D, 0, sizeof(D));
    D.n = T.n; D.grnr64 = T.grnr64; D.x = T.x; D.y = T.y; D.z = T.z; D.grnr32 = T.g32; D.grnr64p = T.g64; D.pay = T.pay;
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

int32_t soap_engine_phase_cycles(const soap_engine *h, uint64_t *out32) {
    if (!h || !out32) return fail(-1, "null argument");
    for (int i = 0; i < 32; i++) out32[i] = h->phase[i];
    return 0;
}

double soap_engine_kernel_ms(const soap_engine *h, const char *name) {
    if (!h || !name) return -1.0;
    if (!strcmp(name, "halo")) return h->ms_halo;
    if (!strcmp(name, "run_setup")) return h->ms_host_setup;
    if (!strcmp(name, "run_pre")) return h->ms_host_pre;
    if (!strcmp(name, "run_classify")) return h->ms_host_classify;
    if (!strcmp(name, "run_total")) return h->ms_host_total;
    if (!strcmp(name, "halo_wave")) return h->cls[0].ms;
    if (!strcmp(name, "halo_wg256")) return h->cls[1].ms;
    if (!strcmp(name, "halo_wg1024")) return h->cls[2].ms;
    if (!strcmp(name, "halo_small")) return h->small[0].ms;
    if (!strcmp(name, "halo_small512")) return h->small[1].ms;
    if (!strcmp(name, "halo_small1024")) return h->small[2].ms;
    if (!strcmp(name, "n_small1024")) return (double)h->small[2].nh;
    if (!strcmp(name, "n_small")) return (double)(h->small[0].nh + h->small[1].nh + h->small[2].nh);
    if (!strcmp(name, "n_small128")) return (double)h->small[0].nh;
    if (!strcmp(name, "n_small512")) return (double)h->small[1].nh;
    if (!strcmp(name, "n_small_escalated")) return (double)h->n_small_escalated;
    if (!strcmp(name, "n_wave")) return (double)h->cls[0].nh;
    if (!strcmp(name, "n_wg256")) return (double)h->cls[1].nh;
    if (!strcmp(name, "n_wg1024")) return (double)h->cls[2].nh;
    if (!strncmp(name, "pipe_build_", 11) || !strncmp(name, "pipe_props_", 11)) {
        const int q = name[5] == 'b' ? 0 : 1;
        const int k = !strcmp(name + 11, "wave") ? 0 : (!strcmp(name + 11, "wg256") ? 1 : (!strcmp(name + 11, "wg1024") ? 2 : -1));
        return k < 0 ? -1.0 : h->cls[k].pms[q];
    }
    if (!strcmp(name, "pipe_build")) return h->ms_build;
    if (!strcmp(name, "pipe_stage_a")) return h->ms_stage_a;
    if (!strcmp(name, "pipe_stage_b")) return h->ms_stage_b;
    if (!strcmp(name, "n_pipeline")) return (double)h->n_pipeline;
    if (!strcmp(name, "n_gbuild")) return (double)h->n_gbuild;
    if (!strcmp(name, "n_single")) return (double)h->n_single;
    if (!strcmp(name, "n_predicted")) return (double)h->n_predicted;
#ifdef SOAP_EXPERIMENTS
    if (!strcmp(name, "experiments")) return 1.0;   // the library carries the experiment knobs and the measured-slower paths (teams, 1024-entry small-halo kernel)
#else
    if (!strcmp(name, "experiments")) return 0.0;
#endif
    if (!strcmp(name, "n_team")) return (double)h->n_team;
    if (!strcmp(name, "n_team_failed")) return (double)h->n_team_failed;
    if (!strcmp(name, "halo_team")) return h->ms_team;
    if (!strcmp(name, "n_passes")) return (double)h->n_passes;   // launches of the single-kernel path (more than one: slab overflows)
    if (!strcmp(name, "mesh") || !strcmp(name, "reorder")) {
        // (the events of a staging call still in flight.  This call cannot report a failure of that call: it is kept in the handle and the
        // next soap_engine_run / stage_* / membership returns it)
        soap_engine *hm = const_cast<soap_engine *>(h);
        (void)hipSetDevice(hm->cfg.device);
        const int32_t frc = finish_stage(hm);
        if (frc) hm->stage_error = frc;
        return !strcmp(name, "mesh") ? h->ms_mesh : h->ms_reorder;
    }
    return -1.0;
}

}  // extern "C"
