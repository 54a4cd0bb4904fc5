/*
 * soap_engine.h -- C ABI of the MI355X halo-property engine.
 *
 * This is the drop-in boundary for SOAP's per-halo hot path.  One engine handle
 * owns one GPU and replaces, for one chunk of particles, the work the reference
 * does inside
 *     process_halos()            SOAP/core/halo_tasks.py:276-430
 *       process_single_halo()    SOAP/core/halo_tasks.py:23-273
 *         SharedMesh.query_radius_periodic()   SOAP/core/shared_mesh.py:122-200
 *         HaloProperty.calculate()             SOAP/particle_selection/<...>.py
 *           property_calculation/<...>.py kernels
 * and the mesh build that precedes it (SharedMesh.__init__, shared_mesh.py:11-114,
 * called from ChunkTask.__call__, SOAP/core/chunk_tasks.py:293-304).
 *
 * Conventions
 *   - plain C types only, no exceptions, no callbacks; every function returns 0 on
 *     success and a negative code on failure (soap_engine_last_error() has the text).
 *   - the caller keeps ownership of every buffer it passes in; the library copies
 *     what it needs into HBM during the stage_* calls.
 *   - a handle is not re-entrant; use one handle per GPU per process (or thread).
 *   - unit system: lengths are COMOVING code lengths, densities are mass per comoving
 *     length^3, velocities/masses are code units.  Physical radii (kpc apertures,
 *     20 Mpc limit) are converted by the caller ONCE (see soap_amd/units.py).
 */
#ifndef SOAP_ENGINE_H
#define SOAP_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOAP_ENGINE_ABI_VERSION 5
#define SOAP_STATS_COLUMNS 6 /* int64 per halo in the stats array of soap_engine_fetch */

/* particle types the selections know about, in the reference's concatenation order
 * (SO_properties.py:3520-3550).  "slot" below is the position in this list. */
#define SOAP_NTYPES 4 /* PartType0, PartType1, PartType4, PartType5 */

typedef struct soap_engine soap_engine; /* opaque */

/* What HaloProperty.__init__ copies from the cellgrid
 * (SOAP/particle_selection/halo_properties.py:8-35), already converted. */
typedef struct soap_engine_config {
    int32_t abi_version;    /* SOAP_ENGINE_ABI_VERSION */
    int32_t device;         /* HIP device ordinal */
    double boxsize;         /* comoving code length */
    double G_eff;           /* Newton G / a : v^2 = G_eff * M / r_comoving */
    double nu_density;      /* mean neutrino density, mass / comoving length^3 */
    double mpc20;           /* 20 Mpc in comoving code length (SO_properties.py:150) */
    double a;               /* expansion factor */
    double mpc;             /* one PHYSICAL Mpc in physical code length.  Only used to mirror the
                               reference's round trip of a radius-multiple SO's required radius
                               through physical Mpc (SO_properties.py:3917, halo_tasks.py:168) */
    double softening[SOAP_NTYPES]; /* comoving code length per type slot */
    int32_t n_workgroups;   /* 0 = library default (persistent grid size) */
    int32_t slab_capacity;  /* 0 = default; max particles in one halo's search sphere */
    /* pseudo-evolution term of the flow rates through spherical shells
     * (SO_properties.py:2875-2884, cosmology from :3446-3451):
     *   R_dot = (2/3) (G M_SO H / 100)^(1/3) (2 Omega_g + 3/2 Omega_m) R_frac */
    double G_phys;          /* Newton G in physical code units */
    double hubble;          /* H(z) in 1 / code time */
    double rdot_omega;      /* 2 Omega_g + 3/2 Omega_m */
} soap_engine_config;

/* calculation kinds: one per HaloProperty subclass on the path */
enum {
    SOAP_CALC_SUBHALO = 0,         /* SubhaloProperties      subhalo_properties.py:2345 */
    SOAP_CALC_SO_DENSITY = 1,      /* SOProperties mean/crit/BN98   SO_properties.py:3215 */
    SOAP_CALC_SO_PHYSICAL = 2,     /* SOProperties type="physical" */
    SOAP_CALC_SO_MULTIPLE = 3,     /* RadiusMultipleSOProperties  SO_properties.py:3797 */
    SOAP_CALC_APERTURE_INCL = 4,   /* InclusiveSphereProperties   aperture_properties.py:4324 */
    SOAP_CALC_APERTURE_EXCL = 5,   /* ExclusiveSphereProperties   aperture_properties.py:4247 */
    SOAP_CALC_PROJECTED = 6        /* ProjectedApertureProperties projected_aperture_properties.py:1580 */
};

/* one entry of halo_prop_list, in the reference's list order
 * (SOAP/compute_halo_properties.py:232-505) */
typedef struct soap_calc_desc {
    int32_t kind;
    int32_t parent;              /* SO_MULTIPLE: position of the parent SO in the list;
                                    APERTURE_EXCL / PROJECTED: position of the next-smaller
                                    aperture of the same family (skip/copy rule,
                                    aperture_properties.py:4082-4123), -1 = none */
    double reference_density;    /* SO_DENSITY: threshold, mass / comoving length^3 */
    double radius;               /* apertures, SO_PHYSICAL: comoving code length;
                                    SO_MULTIPLE: the multiple */
    double physical_radius_mpc;  /* HaloProperty.physical_radius_mpc converted to comoving
                                    code length: radius the scheduler jumps to when this
                                    calculation raises SearchRadiusTooSmallError
                                    (halo_tasks.py:129-139, :168, :187) */
    double prev_radius;          /* comoving radius of `parent` aperture (copy rule) */
    int32_t virial_definition;   /* SO_properties.py:3458-3469 (concentration only then) */
    int32_t filter_limit;        /* category filter of the calculation: minimum number of
                                    bound particles (0 = "basic"), category_filter.py:69-106 */
    int32_t filter_types;        /* bit mask over type slots counted by the filter */
    int32_t flags;               /* SOAP_CALC_FLAG_* */
} soap_calc_desc;

/* soap_calc_desc.flags */
enum {
    SOAP_CALC_FLAG_DM_FLOW = 1,    /* SO_DENSITY with a virial definition: DarkMatterMassFlowRate is wanted
                                      (SO_properties.py:2919-2932); the outer shell reaches 1.05 R_SO, beyond
                                      the search radius the calculation raises SearchRadiusTooSmallError */
    SOAP_CALC_FLAG_INERTIA = 4,    /* bound subhalo / SO: the eight inertia tensors of the total and the dark
                                      matter mass are wanted (inertia_tensors.py:19-132): up to 20 passes per
                                      tensor family over the particles of the search sphere */
    SOAP_CALC_FLAG_ROTATION = 2,   /* gas / star / baryon kappa_corot and disc-to-total fractions are wanted
                                      (kinematic_properties.py:266-425): one more pass over the selection */
    SOAP_CALC_FLAG_PROPERTY_RADIUS = 16, /* APERTURE_INCL / APERTURE_EXCL whose radius is a multiple of a property of the
                                      bound subhalo (aperture_properties.py:3973-3981, :4131-4138;
                                      compute_halo_properties.py:403-440): `radius` holds the multiple, `parent` the
                                      property (SOAP_RADIUS_*), physical_radius_mpc is 0 and no copy rule applies */
    SOAP_CALC_FLAG_STRICT_COPY = 8 /* parameters.strict_halo_copy(): an aperture that repeats the next-smaller one
                                      (EncloseRadius rule) leaves the properties that depend on the aperture size
                                      itself at zero instead of copying them (aperture_properties.py:4117-4120,
                                      projected_aperture_properties.py:1880-1883).  Of the properties this engine
                                      emits those are the iterative 2D inertia tensors of projected apertures */
};

/* soap_calc_desc.parent of a calculation with SOAP_CALC_FLAG_PROPERTY_RADIUS: the BoundSubhalo property, as stored
 * (float32), that is multiplied by `radius` */
enum {
    SOAP_RADIUS_HALF_MASS_TOTAL = 0,   /* BoundSubhalo/HalfMassRadiusTotal */
    SOAP_RADIUS_HALF_MASS_DM = 1,      /* BoundSubhalo/HalfMassRadiusDarkMatter */
    SOAP_RADIUS_HALF_MASS_GAS = 2,     /* BoundSubhalo/HalfMassRadiusGas */
    SOAP_RADIUS_HALF_MASS_STARS = 3,   /* BoundSubhalo/HalfMassRadiusStars */
    SOAP_RADIUS_HALF_MASS_BARYONS = 4, /* BoundSubhalo/HalfMassRadiusBaryons */
    SOAP_RADIUS_ENCLOSE = 5,           /* BoundSubhalo/EncloseRadius */
    SOAP_RADIUS_NPROPS = 6
};

/* per-halo status column */
enum {
    SOAP_HALO_OK = 0,
    SOAP_HALO_NEED_LARGER_READ_RADIUS = 1, /* process_single_halo returned None
                                              (halo_tasks.py:169-181); new radii fetched */
    SOAP_HALO_FATAL_TOO_MANY_BOUND = -1,   /* subhalo_properties.py:2642-2646 */
    SOAP_HALO_FATAL_SO_BEYOND_20MPC = -2,  /* SO_properties.py:150-153 */
    SOAP_HALO_FATAL_SLAB_OVERFLOW = -3,    /* search sphere larger than slab_capacity */
    SOAP_HALO_FATAL_RADIUS_ASSERT = -4     /* halo_tasks.py:78 */
};

/* number of fp64 output columns a calculation kind writes (fixed layout, see
 * soap_amd/layout.py which mirrors csrc/layout.h) */
int32_t soap_engine_calc_columns(int32_t kind);
/* columns a calculation occupies in THIS run: soap_engine_calc_columns(kind), minus the 96 inertia tensor
 * columns at the end of the subhalo / SO blocks when SOAP_CALC_FLAG_INERTIA is not set; plus the 108 columns of
 * the 2D tensors after the three projection blocks of a projected aperture when it is set */
int32_t soap_engine_desc_columns(const soap_calc_desc *desc);

int32_t soap_engine_abi_version(void);
const char *soap_engine_last_error(void);

int32_t soap_engine_create(const soap_engine_config *cfg, soap_engine **out);
int32_t soap_engine_destroy(soap_engine *h);

/*
 * Stage one particle type of the chunk and build its cell-list mesh.
 * Replaces share_array() + SharedMesh(comm, pos, resolution)
 * (chunk_tasks.py:165-170, :293-304; shared_mesh.py:11-114).
 *   type_slot      0..3  (PartType0,1,4,5)
 *   pos            [n][3] doubles, as data[ptype]["Coordinates"]
 *   vel            [n][3] data[ptype]["Velocities"], float64 or float32 (value_bytes = 8 or 4)
 *   mass           [n]    Masses (DynamicalMasses for PartType5), the same width as vel
 *   value_bytes    8 or 4: SWIFT snapshots hold Velocities and Masses in float32 and Coordinates in float64
 *                  (tests/dummy_halo_generator.py:806-811); float32 values are widened on the device, on their way
 *                  into the payload records, and every sum of the engine is fp64.  The reference keeps float32
 *                  arrays float32 (its sums then round in float32, SO_properties.py:400-402): with float32 inputs
 *                  the two agree to float32 rounding, not to 1e-10.
 *   grnr           [n]    GroupNr_bound, int32 or int64 (grnr_bytes = 4 or 8)
 *   fof            [n]    FOFGroupIDs int64
 *   resolution     mesh cells per dimension (chunk_tasks.py:299-302 computes it)
 *   on_device      0: host pointers, 1: device (HBM) pointers on cfg.device, 2: device pointers and an ASYNCHRONOUS call -- it
 *                  returns once the mesh is known and the placement pass (the copy into cell order and payload records, most of
 *                  the staging time) is queued on the engine's stream; the next call on the handle that needs the mesh
 *                  (soap_engine_run, another soap_engine_stage_*, soap_engine_membership, the "mesh" / "reorder" clocks,
 *                  soap_engine_destroy) completes it and reports its errors.  soap_engine_stage_halos does not wait for it: the
 *                  host prepares the halo table while the device places the particles.
 * The call copies everything it needs: the coordinates into cell order (structure of arrays, what the candidate stream of
 * a halo reads) and velocity, mass, GroupNr_bound, FOF id, coordinates and original index of every particle into one
 * 80-byte payload record at the particle's cell-ordered position (what the slab build of a halo gathers).  Lifetime: every
 * input array, host or device, may be reused or freed as soon as the call returns (on_device == 2: as soon as the call that
 * completes the staging returns; until then the arrays must stay allocated and unmodified); with on_device == 0 the device
 * copies made for the mesh build go back to the handle's allocation cache before it returns.
 */
int32_t soap_engine_stage_particles(soap_engine *h, int32_t type_slot, int64_t n,
                                    const double *pos, const void *vel, const void *mass, int32_t value_bytes,
                                    const void *grnr, int32_t grnr_bytes, const int64_t *fof,
                                    int32_t resolution, int32_t on_device);

/*
 * Device inputs (on_device == 1) are read on the engine's own HIP stream.  Call this before staging them with the
 * stream (hipStream_t; NULL = the default stream) their producer ran on -- whatever wrote, converted or copied the
 * arrays: the engine's stream is ordered behind everything submitted to that stream so far, with an event and
 * without synchronising the host.  Host inputs need nothing of the kind.
 */
int32_t soap_engine_wait_stream(soap_engine *h, void *producer_stream);

/*
 * Stage the chunk's neutrino particles (PartType6) and build their mesh.  They enter the spherical-overdensity
 * mass profile with mass * weight (weights may be negative) and feed NumberOfNeutrinoParticles /
 * RawNeutrinoMass / NoiseSuppressedNeutrinoMass and the neutrino terms of the concentration; nothing else
 * sees them (SO_properties.py:380-395, :507-511, :2689-2753).  All three arrays are consumed by the call (weighted and raw
 * masses go into the payload records).  mass and weight are float64 or float32 (value_bytes = 8 or 4, as above).
 * n == 0 removes them.
 */
int32_t soap_engine_stage_neutrinos(soap_engine *h, int64_t n, const double *pos, const void *mass,
                                    const void *weight, int32_t value_bytes, int32_t resolution, int32_t on_device);

/*
 * Stage the chunk's halo table (halo_arrays of process_halos, halo_tasks.py:276-287),
 * in processing order (the reference sorts by nr_bound_part descending,
 * chunk_tasks.py:118-120).  Host pointers.
 */
int32_t soap_engine_stage_halos(soap_engine *h, int64_t n_halo, const double *cofp /*[n][3]*/,
                                const double *search_radius, const double *read_radius,
                                const int64_t *index, const int32_t *is_central,
                                const int64_t *nr_bound_part);

/*
 * Optional, after soap_engine_stage_halos: how far around each halo centre the STAGED particles are complete
 * (resident_radius[i] >= read_radius[i]; host pointer, n_halo doubles; NULL = read_radius, the reference's
 * contract).  A halo whose calculations ask for more than its read radius is then not handed back with
 * SOAP_HALO_NEED_LARGER_READ_RADIUS as long as the enlarged read radius max(1.5 read_radius, search_radius)
 * (halo_tasks.py:393-402) stays within resident_radius: the engine applies the update itself and starts the
 * halo again from the new search radius, exactly what the next process_halos call of ChunkTask's loop
 * (chunk_tasks.py:189-367) would do with a re-read region -- without the re-read.  The radii fetched
 * afterwards are the updated ones.
 */
int32_t soap_engine_set_resident_radius(soap_engine *h, const double *resident_radius);

/*
 * Verification probe, before soap_engine_run: record which particles calculation `calc` of halo row `row`
 * USES -- the prefix of the halo's radially sorted slab the property passes sum over (bound particles only
 * for the bound subhalo and exclusive spheres).  After the run soap_engine_fetch_members returns how many
 * were recorded (<0 on error, 0 if the calculation was filtered / copied / zero) and writes up to `cap`
 * pairs (type_slot, index in the staged arrays of that type), sorted by (type_slot, index).
 * row < 0 switches the probe off.  Projected apertures are not supported (their members are chosen per axis).
 */
int32_t soap_engine_set_member_probe(soap_engine *h, int64_t row, int32_t calc, int64_t cap);
int64_t soap_engine_fetch_members(soap_engine *h, int32_t *type_slot, int64_t *index, int64_t cap);

/*
 * process_halos(): run every calculation of `calcs` for every staged halo.
 * target_density: minimum over the calculations' density multiples
 * (halo_tasks.py:306-317) in mass / comoving length^3; <= 0 means None.
 */
int32_t soap_engine_run(soap_engine *h, const soap_calc_desc *calcs, int32_t n_calcs,
                        double target_density);

/* row length of the result table of the last run (sum of calc columns) */
int64_t soap_engine_row_columns(const soap_engine *h);

/*
 * Copy results of the last run to caller-allocated host buffers (any may be NULL):
 *   table          [n_halo][row_columns] fp64 pre-cast values (zeros where not computed)
 *   status         [n_halo]
 *   search_radius, read_radius  [n_halo] updated radii (halo_tasks.py:393-402)
 *   stats          [n_halo][SOAP_STATS_COLUMNS] int64: n_loop of the last attempt, n_sel (final sphere),
 *                  n_cand (final), shader clocks spent on the halo, n_loop summed over every attempt made
 *                  inside the engine, number of those attempts (1 + restarts inside the resident radius): the
 *                  last two are what halo_tasks.py:188-191, :411-414 add to InputHalos/n_loop and n_process
 */
int32_t soap_engine_fetch(soap_engine *h, double *table, int32_t *status, double *search_radius,
                          double *read_radius, int64_t *stats);

/* device pointer of the result table (for the multi-GPU catalogue gather), or NULL */
const double *soap_engine_table_device(const soap_engine *h);

/* device-to-device copy of the result table into a caller-owned HBM buffer of
 * n_halo*row_columns doubles (e.g. the send buffer of the RCCL catalogue all-gather that
 * replaces combine_chunks' parallel sort, SOAP/core/combine_chunks.py:33-61) */
int32_t soap_engine_copy_table_device(soap_engine *h, double *dst_device);

/*
 * The STORED catalogue of the last run straight from the engine's table, in the caller's row order: for every output row r
 *   out_narrow[r][k] = (float)(table[rows[r]][narrow_cols[k]] * narrow_scale[k])     k < n_narrow
 *   out_wide[r][k]   =         table[rows[r]][wide_cols[k]]   * wide_scale[k]        k < n_wide
 * -- the columns of the table that become datasets, multiplied by their comoving -> stored-unit factor in float64 and cast
 * ONCE to the stored dtype (what ResultSet does on the host, SOAP/core/result_set.py:96-130 with
 * SO_properties.py:3693-3699), in the order of the final catalogue (combine_chunks.py:38-52: `rows` is that permutation;
 * NULL = the table's own order).  One pass over the table instead of a copy, a column gather and a row gather.
 * Every pointer is a DEVICE pointer on cfg.device; the call returns when the outputs are complete.
 */
int32_t soap_engine_pack_catalogue(soap_engine *h, const int64_t *rows, int64_t n_rows,
                                   const int32_t *narrow_cols, const double *narrow_scale, int32_t n_narrow, float *out_narrow,
                                   const int32_t *wide_cols, const double *wide_scale, int32_t n_wide, double *out_wide);

/*
 * Debug/verification entry: original particle indices (positions in the staged arrays) of
 * the particles of `type_slot` whose wrapped radius about halo row `row` satisfies
 * r < radius (strict != 0) or r <= radius (strict == 0), optionally bound-only.
 * Uses the same device arithmetic as the property kernels.  Returns the count (<0 on
 * error); at most `cap` indices are written, sorted ascending.
 */
int64_t soap_engine_membership(soap_engine *h, int64_t row, int32_t type_slot, double radius,
                               int32_t strict, int32_t bound_only, int64_t *out, int64_t cap);

/* average duration in ms of the named internal kernel over the last run, measured with
 * HIP events on the engine's stream ("halo", "reorder", "mesh"); <0 if unknown */
double soap_engine_kernel_ms(const soap_engine *h, const char *name);

/* diagnostic: shader cycles (summed over workgroups, thread 0's clock) the last run spent in
 * each phase of the halo kernel (32 slots): [0] select, [1] sort, [2] materialize+cumsum, [3] moment
 * reductions, [4] scans, [5] SO radius search, [6] whole halo, [7] other; [8..31] sub-phases of those
 * (names in soap_amd/engine.py::PHASE_NAMES) */
int32_t soap_engine_phase_cycles(const soap_engine *h, uint64_t *out32);

#ifdef __cplusplus
}
#endif
#endif /* SOAP_ENGINE_H */
