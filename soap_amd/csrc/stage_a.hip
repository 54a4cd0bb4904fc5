// Pipeline stage 2: the search-radius loop of process_single_halo (SOAP/core/halo_tasks.py:74-187) and the
// success / SearchRadiusTooSmallError logic of every calculate() method for ALL halos at once -- one LANE per halo.
//
// The single-kernel path evaluates this with lanes = calculations inside the halo's own thread group: at most as
// many busy lanes as the list has calculations, three kinds of calculation taking turns, once per halo.  None of it
// touches particles -- it is binary searches and a root solve on the halo's sorted (r, q, cum, dens) columns -- so
// here 64 halos share one instruction stream and the calculations of the list are walked in order, exactly as the
// reference does (halo_tasks.py:117-166): the first calculation that needs a larger sphere ends the round.
//
// What each calculation uses is a prefix of the radially sorted slab the build stage left in the arena; its length,
// radius and SO mass go to the halo's record, the property passes (stage 3) do the sums.
//
//   SOAP/core/halo_tasks.py:74-187, :393-402          radius loop, read-radius update
//   SOAP/particle_selection/SO_properties.py:80-217, :356-457, :3880-3926   SO radius / mass, radius multiples
//   SOAP/particle_selection/aperture_properties.py:4082-4143                skip / copy rule, search-radius check
//   SOAP/particle_selection/subhalo_properties.py:2618-2646                 bound-particle gate
//   SOAP/core/category_filter.py:69-106                                     calculation filters
#include "engine_common.h"

#define SA_BLK 256

namespace {
#include "stage_a.inc"
}  // namespace

__global__ void __launch_bounds__(SA_BLK) k_stage_a(const RunParams *Pp, SlabDev arena, const int *todo, long long n_todo) {
    __shared__ soap_calc_desc s_calcs[MAXCALC];
    __shared__ int s_soslot[MAXCALC];
    const GAS RunParams &P = *(const GAS RunParams *)Pp;
    const int ncalc = P.ncalc;
    {
        const GAS unsigned *csrc = (const GAS unsigned *)P.calcs;
        unsigned *cdst = (unsigned *)&s_calcs[0];
        for (int i = threadIdx.x; i < (int)(sizeof(soap_calc_desc) / 4) * ncalc; i += SA_BLK) cdst[i] = csrc[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int k = 0;
        for (int c = 0; c < ncalc; c++) s_soslot[c] = (s_calcs[c].kind == SOAP_CALC_SO_DENSITY) ? k++ : -1;
    }
    __syncthreads();
    const long long task = blockIdx.x * (long long)SA_BLK + threadIdx.x;
    if (task >= n_todo) return;
    const long long hrow = todo ? (long long)((const GAS int *)todo)[task] : task;
    stage_a_lane(P, arena, hrow, s_calcs, s_soslot);
}

extern "C" void soap_launch_stage_a(hipStream_t stream, const RunParams *params, SlabDev arena, const int *todo, long long n_todo) {
    if (n_todo <= 0) return;
    hipLaunchKernelGGL(k_stage_a, dim3((unsigned)((n_todo + SA_BLK - 1) / SA_BLK)), dim3(SA_BLK), 0, stream, params, arena, todo, n_todo);
}
