// One thread-group class of the per-halo kernels (halo_kernel.inc), selected with -DGCLASS=64|256|1024 and
// -DGINERTIA=0|1, -DGMODE=0|1|2 (single kernel / pipeline build stage / pipeline property stage): compiled as separate translation units so that the classes build in parallel.  Exports a plain
// launcher per variant; engine.hip owns everything on the host side.
#include "engine_common.h"

#if GCLASS == 64
// a wavefront-level "barrier": LDS and global accesses of one wave are issued in order, so only the
// compiler and outstanding memory operations have to be fenced
#define GRP 64
#define GNW 1
#define GWG 64
#define GCH 1024
#define SORT_NBMAX 64
#define GLB 4
#define GATTR __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8)))   // at most 256 registers (VGPR + AGPR): two wavefronts per SIMD
#define GSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
#elif GCLASS == 256
#define GRP 256
#define GNW 4
#define GWG 256
#define GCH 4096
#define SORT_NBMAX 256
#define GLB 2
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
#if GMODE == 1
#define SOAP_VARIANT b      // pipeline stage 1: build the sorted superset in the arena
#elif GMODE == 2
#define SOAP_VARIANT s      // pipeline stage 3: the property passes
#elif GINERTIA
#define SOAP_VARIANT i      // the single-kernel path with the inertia tensor passes compiled in (their registers and
#else                       // stack stay out of the variant every other run uses)
#define SOAP_VARIANT n
#endif

namespace {
#include "halo_kernel.inc"
}

extern "C" void SOAP_CAT3(soap_launch_halo_, GCLASS, SOAP_VARIANT)(int groups, hipStream_t stream, const RunParams *params, SlabDev slab,
                                                                 unsigned *queue, const int *todo, long long n_todo) {
    hipLaunchKernelGGL(k_halo, dim3(groups), dim3(GRP), 0, stream, params, slab, queue, todo, n_todo);
}
