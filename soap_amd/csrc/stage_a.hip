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
#define RES_OK 0
#define RES_TOO_SMALL 1

namespace {

struct SlabView {   // this lane's halo: columns of its arena slice
    const GAS double *r, *q, *cum, *dens;
};

// entries of [0, n) with r <= R / r < R / q <= R2 (sorted columns)
SOAP_DEV int count_le(const GAS double *a, int n, double v) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1; else hi = mid; }
    return lo;
}
SOAP_DEV int count_lt(const GAS double *a, int n, double v) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

SOAP_DEV double cum_nu(const SlabView &S, int i, double nuK) {
    return (nuK == 0.0) ? S.cum[i] : S.cum[i] + nuK * cube(S.r[i]);
}

// SO radius and mass for the search sphere [0, n) (SO_properties.py:356-457, :80-217, RadiusMultiple :3912-3923);
// the same walk as so_eval_lane of the single-kernel path.
SOAP_DEV int so_eval(const SlabView &S, const soap_calc_desc &cd, double Rcur, int n, int nskip, double parent_R, int fb,
                     double nuK, double mpc20, double a, double mpc, double *required, int *state, int *n_in, double *R_out,
                     double *M_out) {
    *state = 2; *n_in = 0; *R_out = 0.0; *M_out = 0.0;
    if (n == 0) return RES_OK;
    double R_phys = 0.0;
    if (cd.kind == SOAP_CALC_SO_MULTIPLE) {
        // parent radius as stored (float32); zero if the parent SO does not exist
        // the FLOAT32 product multiple * R is the radius of this SO (SO_properties.py:3917-3920; see so_eval_lane)
        const float r32 = (float)parent_R;
        const float to_mpc32 = (float)(a / mpc), from_mpc32 = (float)(mpc / a);
        const float r_mpc32 = (float)cd.radius * (r32 * to_mpc32);
        R_phys = (double)(r_mpc32 * from_mpc32);
        *required = R_phys;
        if ((double)((float)cd.radius * r32) > Rcur) return RES_TOO_SMALL;
    } else if (cd.kind == SOAP_CALC_SO_PHYSICAL) {
        R_phys = cd.radius;
    }
    const int nr_parts = n - nskip;
    double SO_r = 0.0, SO_mass = 0.0;
    int have = 0;
    if (cd.kind == SOAP_CALC_SO_DENSITY) {
        const double rho = cd.reference_density;
        if (nr_parts > 0) {
            const bool above0 = S.dens[nskip] > rho;
            int res = 0;
            if (above0) {
                int ib = -1;   // first index beyond nskip whose enclosed density is not above the threshold
                if (fb > nskip) {
                    if (fb < n) ib = fb;   // tabulated while the density profile was built
                } else {
                    for (int i = nskip + 1; i < n; i++)
                        if (!(S.dens[i] > rho)) { ib = i; break; }
                }
                if (ib < 0) {
                    res = (S.r[n - 1] > mpc20) ? SOAP_HALO_FATAL_SO_BEYOND_20MPC : 1;
                } else {
                    int i = ib;
                    double r1 = S.r[i - 1], r2 = S.r[i];
                    double M1 = cum_nu(S, i - 1, nuK), M2 = cum_nu(S, i, nuK);
                    bool a1 = S.dens[i - 1] > rho, a2 = S.dens[i] > rho;
                    while (r1 == r2 || a1 == a2) {
                        i++;
                        if (i >= n) { res = (S.r[n - 1] > mpc20) ? SOAP_HALO_FATAL_SO_BEYOND_20MPC : 1; break; }
                        r1 = r2; r2 = S.r[i];
                        M1 = M2; a1 = a2;
                        a2 = S.dens[i] > rho;
                        M2 = cum_nu(S, i, nuK);
                    }
                    if (res == 0) {
                        const double rho_dim = rho * cube(r1) / M1;
                        const double slope_dim = (M2 - M1) / (r2 - r1) * (r1 / M1);
                        SOIntersection f;
                        f.c3 = 4.0 * M_PI / 3.0 * rho_dim;
                        f.s = slope_dim;
                        int err;
                        const double u = brentq(f, 1.0, r2 / r1, &err);
                        SO_r = r1 * u;
                        const double vol = 4.0 / 3.0 * M_PI * cube(SO_r);
                        SO_mass = vol * rho;
                    }
                }
            } else {
                int ip = -1;
                for (int i = nskip; i < n; i++)
                    if (!(cum_nu(S, i, nuK) < 0.0)) { ip = i; break; }
                if (ip < 0) res = SOAP_HALO_FATAL_SO_BEYOND_20MPC;   // "Should never happen"
                else {
                    const double Mi = cum_nu(S, ip, nuK), ri = S.r[ip];
                    SO_r = sqrt(0.75 * Mi / (M_PI * ri * rho));
                    SO_mass = Mi * SO_r / ri;
                }
            }
            if (res == 1) return RES_TOO_SMALL;
            if (res < 0) return res;
            have = 1;
        }   // else: the reference leaves SO_r undefined; documented divergence -> zeros
    } else if (R_phys > 0.0) {
        SO_r = R_phys;
        if (nr_parts > 0) {
            const int nin = count_le(S.r, n, SO_r);
            // SO_properties.py:436-452 on the profile with the first nskip entries dropped
            const int i = nin < nskip ? nskip : nin;   // first retained element with r > SO_r
            if (i >= n) SO_mass = cum_nu(S, n - 1, nuK);
            else if (i == nskip) SO_mass = cum_nu(S, i, nuK);
            else {
                const double M1 = cum_nu(S, i - 1, nuK), M2 = cum_nu(S, i, nuK), r1 = S.r[i - 1], r2 = S.r[i];
                SO_mass = M1 + (SO_r - r1) / (r2 - r1) * (M2 - M1);
            }
            have = 1;
        }
    }
    if (have && SO_r > 0.0 && SO_mass > 0.0) {
        *state = 1;
        *n_in = count_lt(S.r, n, SO_r);   // r < SO_r (strict, SO_properties.py:485)
        *R_out = SO_r; *M_out = SO_mass;
    }
    return RES_OK;
}

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
    GAS HaloHdr &H = ((GAS HaloHdr *)P.hdr)[hrow];
    if (H.pipe_state != PIPE_BUILT) return;
    SlabView S;
    S.r = (const GAS double *)arena.r + H.off; S.q = (const GAS double *)arena.q + H.off;
    S.cum = (const GAS double *)arena.cum + H.off; S.dens = (const GAS double *)arena.dens + H.off;
    const double nuK = ((P.nu_density * 4.0) / 3.0) * M_PI;
    const double mpc20 = P.mpc20, a = P.a, mpc = P.mpc, target = P.target_density;
    const int n_have = H.n_have;
    const double R_have = H.R_have;
    const int i_firstpos = H.i_firstpos, i_lastbound = H.i_lastbound, i_firstdm = H.i_firstdm;
    const double nb_tot = ((H.nb[0] + H.nb[1]) + H.nb[2]) + H.nb[3];
    const float enc32 = (float)H.encl;   // EncloseRadius as stored (aperture_properties.py:4086-4123)
    double Rread = ((const GAS double *)P.halo.read_radius)[hrow];
    double Rsearch = ((const GAS double *)P.halo.search_radius)[hrow];
    const double Rresident = P.halo.resident ? ((const GAS double *)P.halo.resident)[hrow] : Rread;
    const int central = ((const GAS int *)P.halo.is_central)[hrow];
    const long long nr_bound = ((const GAS long long *)P.halo.nr_bound)[hrow];
    const bool use_target = central == 1 && target > 0.0;
    const unsigned long long alldone = (ncalc >= 64) ? ~0ull : ((1ull << ncalc) - 1ull);
    unsigned long long done = 0ull, cached = 0ull;
    double Rcur = Rsearch;
    int status = SOAP_HALO_OK, n_restart = 0, n_final = 0;
    long long n_loop = 0, n_loop_all = 0;
    int pipe = PIPE_DONE;
    double R_need = 0.0;
    double over_R = 0.0, over_density = 0.0;   // last round whose enclosed density was still above the target
    while (true) {
        n_loop++; n_loop_all++;
        if (!(Rcur <= Rread)) { status = SOAP_HALO_FATAL_RADIUS_ASSERT; break; }
        if (Rcur > R_have) {
            // the build stage has to reach further: at least to this round's radius; when the loop is still waiting for
            // the enclosed density to fall below the target, to where it will have -- beyond a halo the enclosed density
            // falls about as R^-2 (R^-3 once nothing more is added) -- so that ONE rebuild is enough
            pipe = PIPE_NEED_SUPERSET; R_need = Rcur;
            if (over_R > 0.0 && target > 0.0) R_need = fmax(R_need, over_R * sqrt(over_density / target));
            break;
        }
        const int n = (Rcur == R_have) ? n_have : count_le(S.q, n_have, Rcur * Rcur);
        n_final = n;
        double required = 0.0;
        // density = mass_total / (4/3 pi R^3) <= target  (halo_tasks.py:97-103)
        const double mass_total = n > 0 ? S.cum[n - 1] : 0.0;
        const double density = mass_total / (4.0 / 3.0 * M_PI * cube(Rcur));
        if (use_target && density > target) { over_R = Rcur; over_density = density; } else over_R = 0.0;
        if (!use_target || density <= target) {
            int fatal = 0;
            bool stop = false;
            if (!(done & 1ull)) {
                // bound-particle gate: every bound particle of the superset lies inside the sphere once the outermost
                // one does; the build stage has sent halos with MORE bound particles than expected elsewhere
                const bool all_in = n > i_lastbound;
                if (nb_tot < (double)nr_bound || !all_in) { required = fmax(required, s_calcs[0].physical_radius_mpc); stop = true; }
                else { done |= 1ull; H.cs_state[0] = 1; H.cs_n[0] = n; H.cs_R[0] = 0.0; H.cs_M[0] = 0.0; }
            }
            if (!stop && done != alldone) {
                // nskip = max(1, argmax(r > 0))  (SO_properties.py:416)
                int nskip = 1;
                if (central && n > 0) { nskip = i_firstpos < n ? i_firstpos : 0; nskip = nskip < 1 ? 1 : nskip; }
                for (int c = 1; c < ncalc; c++) {
                    if ((done >> c) & 1ull) continue;
                    const soap_calc_desc &cd = s_calcs[c];
                    int res = RES_OK, st = 2, nin = 0;
                    double Rv = 0.0, Mv = 0.0, creq = cd.physical_radius_mpc;
                    bool filt = true;
                    if (cd.filter_limit > 0) {   // category filter (category_filter.py:69-106)
                        double v = 0.0;
                        for (int t = 0; t < 4; t++) if (cd.filter_types & (1 << t)) v += H.nb[t];
                        filt = v >= (double)cd.filter_limit;
                    }
                    const bool is_cached = (cached >> c) & 1ull;
                    if (is_cached) { st = H.cs_state[c]; nin = H.cs_n[c]; Rv = H.cs_R[c]; Mv = H.cs_M[c]; }
                    else if (cd.kind == SOAP_CALC_SO_DENSITY || cd.kind == SOAP_CALC_SO_PHYSICAL) {
                        if (central && filt)
                            res = so_eval(S, cd, Rcur, n, nskip, 0.0, s_soslot[c] >= 0 ? H.cs_fb[s_soslot[c]] : -1, nuK, mpc20, a, mpc,
                                          &creq, &st, &nin, &Rv, &Mv);
                    } else if (cd.kind == SOAP_CALC_SO_MULTIPLE) {
                        if (central && filt) {
                            // every earlier calculation is finished when the walk gets here: the parent's committed radius
                            const double rp = (H.cs_state[cd.parent] == 1) ? H.cs_R[cd.parent] : 0.0;
                            res = so_eval(S, cd, Rcur, n, nskip, rp, -1, nuK, mpc20, a, mpc, &creq, &st, &nin, &Rv, &Mv);
                        }
                    } else if (cd.kind == SOAP_CALC_APERTURE_INCL || cd.kind == SOAP_CALC_APERTURE_EXCL) {
                        bool skip = false;
                        int copy_from = -1;
                        if (cd.parent >= 0 && cd.prev_radius > (double)enc32) {
                            skip = true;
                            if (cd.kind == SOAP_CALC_APERTURE_EXCL && filt) copy_from = cd.parent;
                        }
                        if (filt && !skip) {
                            if (Rcur < cd.radius) res = RES_TOO_SMALL;   // aperture_properties.py:4140-4143
                            else { st = 1; nin = count_le(S.r, n, cd.radius); Rv = cd.radius; }   // r <= R_ap (:310)
                        } else if (copy_from >= 0) { st = 3; nin = copy_from; }
                    } else if (cd.kind == SOAP_CALC_PROJECTED) {
                        // bound particles only: no search-radius check (projected_aperture_properties.py:1889-1891)
                        int copy_from = -1;
                        bool skip = false;
                        if (cd.parent >= 0 && cd.prev_radius > (double)enc32) { skip = true; if (filt) copy_from = cd.parent; }
                        if (filt && !skip) { st = 1; Rv = cd.radius; }
                        else if (copy_from >= 0) { st = 3; nin = copy_from; }
                    }
                    // a selection that was found stays what it is when the sphere grows (a prefix of the sorted slab): kept
                    // for the rounds in which this calculation still waits (for its flow-rate shell, below)
                    if (!is_cached && res == RES_OK && (st == 1 || st == 3) && cd.kind != SOAP_CALC_SO_MULTIPLE) {
                        cached |= 1ull << c;
                        H.cs_state[c] = st; H.cs_n[c] = nin; H.cs_R[c] = Rv; H.cs_M[c] = Mv;
                    }
                    // DarkMatterMassFlowRate: the shell around R_SO reaches R + dR/2, dR = 0.1 R (SO_properties.py:2846-2850),
                    // unless no dark matter lies inside R_SO (:2923); checked in every round
                    if (res == RES_OK && st == 1 && cd.kind == SOAP_CALC_SO_DENSITY && (cd.flags & SOAP_CALC_FLAG_DM_FLOW)) {
                        const double dR = 0.1 * Rv;
                        if (Rv + (dR / 2) > Rcur && i_firstdm < nin) res = RES_TOO_SMALL;
                    }
                    if (res != RES_OK) {   // first failure in list order ends the round (halo_tasks.py:129-139)
                        if (res == RES_TOO_SMALL) required = fmax(required, creq);
                        else fatal = res;
                        break;
                    }
                    done |= 1ull << c;
                    H.cs_state[c] = st; H.cs_n[c] = nin; H.cs_R[c] = Rv; H.cs_M[c] = Mv;
                }
            }
            if (fatal) { status = fatal; break; }
            if (done == alldone) break;
        }
        // halo_tasks.py:167-187
        if (required > Rread || Rcur >= Rread) {
            Rsearch = fmax(Rsearch, required > Rread ? required : Rcur);
            const double Rread_new = fmax(Rread * 1.5, Rsearch);   // halo_tasks.py:393-402
            if (Rread_new <= Rresident && n_restart < 64) {
                // the enlarged region is resident: start again from the new search radius with every calculation open
                Rread = Rread_new;
                n_restart++;
                done = 0ull; cached = 0ull;
                Rcur = Rsearch;
                n_loop = 0;
                continue;
            }
            status = SOAP_HALO_NEED_LARGER_READ_RADIUS;
            break;
        }
        Rcur = fmin(Rcur * 1.2, Rread);
        Rcur = fmax(Rcur, required);
    }
    if (pipe == PIPE_NEED_SUPERSET) {
        H.R_need = R_need;
        H.pipe_state = PIPE_NEED_SUPERSET;
        return;
    }
    H.status = status;
    H.n_final = n_final;
    H.n_loop = n_loop;
    H.n_loop_all = n_loop_all;
    H.R_final = Rcur;
    H.n_restart = n_restart;
    H.R_search = Rsearch;
    H.R_read = status == SOAP_HALO_NEED_LARGER_READ_RADIUS ? fmax(Rread * 1.5, Rsearch) : Rread;
    if (status != SOAP_HALO_OK)
        for (int c = 0; c < ncalc; c++) H.cs_state[c] = 0;
    H.pipe_state = PIPE_DONE;
}

extern "C" void soap_launch_stage_a(hipStream_t stream, const RunParams *params, SlabDev arena, const int *todo, long long n_todo) {
    if (n_todo <= 0) return;
    hipLaunchKernelGGL(k_stage_a, dim3((unsigned)((n_todo + SA_BLK - 1) / SA_BLK)), dim3(SA_BLK), 0, stream, params, arena, todo, n_todo);
}
