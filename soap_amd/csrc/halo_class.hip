// One thread-group class of the per-halo kernels (halo_kernel.inc), selected with -DGCLASS=64|256|1024 and
// -DGINERTIA=0|1, -DGMODE=0|1|2 (single kernel / pipeline build stage / pipeline property stage): compiled as separate translation units so that the classes build in parallel.  Exports a plain
// launcher per variant; engine.hip owns everything on the host side.
#include "engine_common.h"
#include <algorithm>

#if GCLASS == 64
// a wavefront-level "barrier": LDS and global accesses of one wave are issued in order, so only the
// compiler and outstanding memory operations have to be fenced
#define GRP 64
#define GNW 1
#define GWG 64
#ifdef G64_CH
#define GCH G64_CH         // (experiments: scripts/build_variant.sh)
#else
#define GCH 1024
#endif
#define SORT_NBMAX 64
#define GLB 4
#define SCAN_E 2           // consecutive particles per lane in the radial scan (fused_scan): most halos of this class fit one or two blocks anyway (4: 13.9 ms for the class alone on C3, 2: 13.6, 1: 13.6)
#ifndef G64_WPE
#define G64_WPE 2
#endif
#define GATTR __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(G64_WPE, 8)))   // at most 256 registers (VGPR + AGPR): two wavefronts per SIMD
#define GSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
#elif GCLASS == 256
#define GRP 256
#define GNW 4
#define GWG 256
#if GMODE == 3
#define GCH 2048           // the tile-parallel build sorts bucket groups of at most this many entries: less LDS, more groups per CU
#elif defined(G256_CH)
#define GCH G256_CH        // (experiments: scripts/build_variant.sh)
#else
#define GCH 4096
#endif
#define SORT_NBMAX 256
#ifdef G256_LB
#define GLB G256_LB
#else
#define GLB 2
#endif
#define SCAN_E 4
#define GATTR __launch_bounds__(GRP, GLB)
#define GSYNC() __syncthreads()
#elif GCLASS == 1024
// threads of the largest class (512 threads x 256 VGPRs measured slower: 35 vs 29 ms on C3)
#define GRP 1024
#define GNW 16
#define GWG 1024
#define GCH 8192
#define SORT_NBMAX 1024
#define GLB 4
#define SCAN_E 4
#define GATTR __launch_bounds__(GRP, GLB)
#define GSYNC() __syncthreads()
#else
#error "GCLASS must be 64, 256 or 1024"
#endif

#define SOAP_CAT3_(a, b, c) a##b##c
#define SOAP_CAT3(a, b, c) SOAP_CAT3_(a, b, c)
#ifndef GMODE
#define GMODE 0
#endif
#if GMODE == 3
#define SOAP_VARIANT g      // pipeline stage 1 for large halos: several workgroups per halo
#elif GMODE == 1
#define SOAP_VARIANT b      // pipeline stage 1: build the sorted superset in the arena
#elif GMODE == 2
#define SOAP_VARIANT s      // pipeline stage 3: the property passes
#elif GINERTIA && defined(LCALC)
#define SOAP_VARIANT i32    // ... with LDS tables for LCALC (32) calculations instead of MAXCALC: one more resident group per CU
#elif GINERTIA
#define SOAP_VARIANT i      // the single-kernel path with the inertia tensor passes compiled in (their registers and
#elif defined(LCALC)        // stack stay out of the variant every other run uses)
#define SOAP_VARIANT n32
#else
#define SOAP_VARIANT n
#endif

namespace {
#include "halo_kernel.inc"
}

#if GMODE == 3
// the whole tile-parallel build of `nlarge` large halos: ten launches on one stream, kernel boundaries are the
// only synchronisation between the phases
extern "C" void soap_launch_gbuild(hipStream_t stream, const RunParams *params, SlabDev arena, long long nlarge, long long ntiles,
                                   long long ngblocks, long long nselparts, const int *rows, const int *first_tile, const int *ntile,
                                   const int *first_gblock, const int *ngblock, const int *first_sel, const int *nsel,
                                   const int *bin_first, const int *nbin) {
    if (nlarge <= 0) return;
    const unsigned g64 = (unsigned)((nlarge + 63) / 64);
    hipLaunchKernelGGL(k_ginit, dim3(g64), dim3(64), 0, stream, params, nlarge, rows, first_tile, ntile, first_gblock, ngblock, first_sel, nsel, bin_first, nbin);
    // fixed grids that walk their tiles / bucket groups (a workgroup per tile would cost more in dispatch than in work)
    const unsigned gt = (unsigned)std::min<long long>(ntiles, 256 * 8),
                   gg = (unsigned)std::min<long long>(ngblocks, 256 * 2);
    hipLaunchKernelGGL(k_gselect, dim3((unsigned)std::min<long long>(nselparts, 256 * 2)), dim3(GRP), 0, stream, params, arena, nselparts);
    hipLaunchKernelGGL(k_gsplit, dim3((unsigned)nlarge), dim3(GRP), 0, stream, params, arena);
    hipLaunchKernelGGL(k_ghist, dim3(gt), dim3(256), 0, stream, params, arena, ntiles);
    hipLaunchKernelGGL(k_gscan, dim3((unsigned)nlarge), dim3(64), 0, stream, params, arena);
    hipLaunchKernelGGL(k_gscatter, dim3(gt), dim3(256), 0, stream, params, arena, ntiles);
    hipLaunchKernelGGL(k_gsortgroups, dim3(gg), dim3(GRP), 0, stream, params, arena, ngblocks);
    hipLaunchKernelGGL(k_ggather, dim3(gt), dim3(256), 0, stream, params, arena, ntiles);
    hipLaunchKernelGGL(k_gcumsum, dim3((unsigned)nlarge), dim3(64), 0, stream, params, arena);
    hipLaunchKernelGGL(k_gpost, dim3(gt), dim3(256), 0, stream, params, arena, ntiles);
    hipLaunchKernelGGL(k_gfinal, dim3(g64), dim3(64), 0, stream, params, arena, nlarge);
}
#ifdef SOAP_EXPERIMENTS
// workgroups of the team kernel that can be resident on the device at the same time (its groups meet at a spinning barrier)
extern "C" int soap_team_max_groups() {
    int per_cu = 0, dev = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_team, GRP, 0) != hipSuccess || hipGetDevice(&dev) != hipSuccess) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    return per_cu * prop.multiProcessorCount;
}
// records of the large halos of a team run (k_ginit), then the team kernel: T teams of K workgroups, one persistent launch
extern "C" void soap_launch_team(hipStream_t stream, const RunParams *params, SlabDev arena, TeamCtl *ctl, long long nlarge, int groups, int split,
                                 long long ntiles, long long ngblocks, long long nselparts,
                                 const int *rows, const int *first_tile, const int *ntile, const int *first_gblock, const int *ngblock,
                                 const int *first_sel, const int *nsel, const int *bin_first, const int *nbin, double *bpart, double *bscan) {
    if (nlarge <= 0 || groups <= 0) return;
    hipLaunchKernelGGL(k_ginit, dim3((unsigned)((nlarge + 63) / 64)), dim3(64), 0, stream, params, nlarge, rows, first_tile, ntile, first_gblock, ngblock, first_sel, nsel, bin_first, nbin);
    hipLaunchKernelGGL(k_team, dim3((unsigned)groups), dim3(GRP), 0, stream, params, arena, ctl, nlarge, 0, split, bpart, bscan, ntiles, ngblocks, nselparts);
}
#endif   // SOAP_EXPERIMENTS
// tile sums for the property stage of the large halos whose radius loop has finished (k_bpartials)
extern "C" void soap_launch_bpartials(hipStream_t stream, const RunParams *params, SlabDev arena, long long ntiles, double *bpart) {
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(k_bpartials, dim3((unsigned)std::min<long long>(ntiles, 256 * 8)), dim3(GRP), 0, stream, params, arena, ntiles, bpart);
    // the tiles' cumulative weights and v_max candidates, behind the sums in the same allocation
    double *bscan = bpart + ntiles * 2 * NACC;
    hipLaunchKernelGGL(k_bscan, dim3((unsigned)std::min<long long>(ntiles, 256 * 16)), dim3(256), 0, stream, params, arena, ntiles, (const double *)bpart, bscan);
}
#else
extern "C" void SOAP_CAT3(soap_launch_halo_, GCLASS, SOAP_VARIANT)(int groups, hipStream_t stream, const RunParams *params, SlabDev slab,
                                                                 unsigned *queue, const int *todo, long long n_todo) {
    hipLaunchKernelGGL(k_halo, dim3(groups), dim3(GRP), 0, stream, params, slab, queue, todo, n_todo);
}
#endif
