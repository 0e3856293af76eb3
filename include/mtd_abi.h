/* include/mtd_abi.h — C ABI of libmtd_hip.so: the MI355X (gfx950) metadynamics hot path.
 *
 * This is the drop-in boundary.  Every entry point is `extern "C"`, takes plain pointers, sizes
 * and PODs (no torch / pybind / HOOMD types), enqueues work on an explicit HIP stream and returns
 * an int status (0 = success, >0 = hipError_t, <0 = MTD_ERR_*).  No entry point synchronises the
 * device unless its comment says so.  The host classes in metadynamics-plugin_amd/host (the mirror
 * of the reference's CollectiveVariable / IntegratorMetaDynamics C++ classes) and the test / bench
 * harness (ctypes) call nothing else.
 *
 * Each block cites the reference interface it replaces (file:line under
 * /root/reference/metadynamics/).  The reference's kernel drivers are C++ free functions taking
 * HOOMD PODs (BoxDim, Index2D, GPUPartition) and raw device pointers borrowed from ArrayHandle;
 * here BoxDim becomes `mtd_box`, the default stream becomes an explicit `mtd_stream_t`, and the
 * tiny per-CV configuration arrays (Miller indices, per-type mode coefficients, grid geometry) are
 * passed as HOST pointers and travel in the kernel-argument segment instead of device arrays.
 *
 * Particle arrays keep HOOMD's layout: postype = Scalar4 (x, y, z, type bit-cast into w),
 * force = Scalar4 (fx, fy, fz, energy), with Scalar selected per call by `dtype`
 * (MTD_F32: w holds the int32 type bit pattern; MTD_F64: the LOW 32 bits of w hold it, HOOMD's
 * __scalar_as_int convention).  All reductions and the whole bias grid are double precision.
 */
#ifndef MTD_ABI_H
#define MTD_ABI_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTD_ABI_VERSION 1

typedef void *mtd_stream_t; /* hipStream_t */

enum mtd_dtype { MTD_F32 = 0, MTD_F64 = 1 };

enum mtd_status
    {
    MTD_SUCCESS = 0,
    MTD_ERR_INVALID_ARGUMENT = -1, /* the reference throws std::runtime_error for these */
    MTD_ERR_UNSUPPORTED = -2,
    MTD_ERR_NO_DEVICE = -3,
    MTD_ERR_COMM_TIMEOUT = -4,     /* a mailbox wait expired: the step is poisoned (NaN), sticky until the mailbox is destroyed */
    MTD_ERR_COLLECTIVE = -5        /* an RCCL call failed (mtd_rccl_last_error says which and why), or the walkers of a multiple-walker
                                      run disagree on stride / add_hills / timestep (they would enter different collectives) */
    };

/* HOOMD BoxDim as a POD (reference: hoomd/BoxDim.h, passed by const-ref to every driver) */
typedef struct mtd_box
    {
    double L[3];
    double lo[3];
    double xy, xz, yz;
    unsigned char periodic[3];
    unsigned char _pad[5];
    } mtd_box;

int mtd_abi_version(void);
const char *mtd_status_string(int status);
/* number of visible HIP devices, or a negative mtd_status */
int mtd_device_count(void);

/* ================================================================================================
 * Lamellar order parameter
 * replaces LamellarOrderParameterGPU.cuh:21-54 (gpu_calculate_fourier_modes, gpu_compute_sq_forces)
 * ============================================================================================== */

#define MTD_MAX_CV 8     /* lamellar CVs fused into one pass over the particles */
#define MTD_MAX_MODES 64 /* total Fourier modes over the fused CVs */
#define MTD_MAX_TYPES 16

/* n_cv lamellar CVs evaluated in ONE pass over the positions (the reference launches one
 * kernel grid per CV and re-reads every position n_wave times, LamellarOrderParameterGPU.cu:130). */
typedef struct mtd_lamellar_set
    {
    unsigned int n_cv;
    unsigned int n_types;
    unsigned int n_modes;                       /* total, CV-major: modes of CV c are [first[c], first[c+1]) */
    unsigned int first[MTD_MAX_CV + 1];
    int hkl[MTD_MAX_MODES][3];                  /* Miller indices (cv.py:251-256, std_vector_int3) */
    double coeff[MTD_MAX_CV][MTD_MAX_TYPES];    /* per-type mode coefficients (cv.py:242-249) */
    int trig_mode;                              /* MTD_TRIG_DEFAULT (0: the process default, mtd_lamellar_set_fast_trig), MTD_TRIG_HARDWARE
                                                   or MTD_TRIG_ACCURATE for THIS set's kernels — see below */
    } mtd_lamellar_set;

enum mtd_trig_mode { MTD_TRIG_DEFAULT = 0, MTD_TRIG_HARDWARE = 1, MTD_TRIG_ACCURATE = 2 };

/* workspace (device doubles) needed for n_particles: block partial sums */
size_t mtd_lamellar_scratch_doubles(unsigned int n_particles);

/* gpu_calculate_fourier_modes (LamellarOrderParameterGPU.cuh:21-29, .cu:104-147), one CV:
 * d_fourier_modes[2*n_wave] = sum_j a(type_j) (cos, sin)(q_k . r_j), k < n_wave, as doubles.
 * lattice_vectors / mode are HOST arrays (int[3*n_wave], double[n_types]). */
int mtd_calculate_fourier_modes(unsigned int n_wave, const int *lattice_vectors, unsigned int n_particles,
                                const void *d_postype, int dtype, const double *mode, unsigned int n_types,
                                double *d_fourier_modes, double *d_scratch, const mtd_box *global_box,
                                mtd_stream_t stream);

/* Fused hot path (no reference counterpart; replaces computeCV's kernel + D2H + host sum,
 * LamellarOrderParameterGPU.cc:34-96): for every CV c of the set,
 *   d_partials[b * n_cv + c] = sum over the particles of block b of a_c(type_j) sum_k cos(q_k . r_j)
 * for b < *n_partials.  The CV value is s_c = (1/N_global) sum_b d_partials[b*n_cv + c]; that last
 * tiny sum is done by the consumer (mtd_metad_set_cv_source) or by mtd_reduce_partials. */
int mtd_lamellar_cv_partials(const mtd_lamellar_set *set, unsigned int n_particles, const void *d_postype,
                             int dtype, const mtd_box *global_box, double *d_partials,
                             unsigned int *n_partials, mtd_stream_t stream);

/* out[c] = shift + scale * sum_{b<n_partials} d_partials[b*stride + c], c < count (one tiny block,
 * fixed summation order => bitwise reproducible) */
int mtd_reduce_partials(const double *d_partials, unsigned int n_partials, unsigned int stride,
                        unsigned int count, double scale, double shift, double *d_out, mtd_stream_t stream);

/* gpu_compute_sq_forces (LamellarOrderParameterGPU.cuh:43-54, .cu:198-236), one CV, bias as a host
 * scalar exactly like the reference: F_j = bias * (2/n_global) a(type_j) sum_k q_k sin(q_k . r_j), w = 0 */
int mtd_compute_sq_forces(unsigned int n_particles, const void *d_postype, void *d_force, int dtype,
                          unsigned int n_wave, const int *lattice_vectors, const double *mode,
                          unsigned int n_types, unsigned int n_global, double bias,
                          const mtd_box *global_box, mtd_stream_t stream);

/* Fused hot path: forces of all CVs of the set in one pass; the bias factors dV/ds_c are read from
 * DEVICE memory (d_bias[c], written by mtd_metad_update_bias) so no host round trip separates the
 * CV reduction from the force pass.  d_force[c] (host array of n_cv device pointers) receives CV c's
 * Scalar4 force array (the reference's per-CV ForceCompute::m_force). */
int mtd_lamellar_forces(const mtd_lamellar_set *set, unsigned int n_particles, const void *d_postype,
                        void *const *d_force, int dtype, unsigned int n_global, const double *d_bias,
                        const mtd_box *global_box, mtd_stream_t stream);

/* Trigonometry of the lamellar kernels.  MTD_TRIG_HARDWARE: v_sin_f32 / v_cos_f32 on the phase in turns — what the reference's
 * GPU kernels do (fast::sin / fast::cos, LamellarOrderParameterGPU.cu:36-37, 180); MTD_TRIG_ACCURATE: ocml sinpi / cospi.  The
 * mode is a property of the CV SET (mtd_lamellar_set::trig_mode); sets that leave it at MTD_TRIG_DEFAULT take the process default,
 * which this call sets (1, the initial value: hardware; 0: accurate) — two sets with different modes can run in one process.
 * PRECONDITION of the hardware mode: the instructions' domain is +-256 turns, outside it they return cos = 1 / sin = 0 without
 * any error.  A phase is sum_i hkl_i * g_i with g_i = b_i . r the fractional coordinate: positions inside the box give
 * |phase| <= (|h| + |k| + |l|) / 2.  The library therefore takes the hardware path only for mode sets with |h| + |k| + |l| <= 100
 * per mode (others run the accurate path whatever the mode says), which leaves room for particles up to FIVE box lengths outside
 * the box; a caller that hands over UNWRAPPED coordinates further out than that must select MTD_TRIG_ACCURATE (correct for any
 * range) — HOOMD keeps its particles wrapped, the reference's kernels rely on the same. */
int mtd_lamellar_set_fast_trig(int enable);
int mtd_lamellar_get_fast_trig(void);

/* ================================================================================================
 * Bias grid ("IntegratorMetaDynamics" grid engine), device resident
 * replaces IntegratorMetaDynamics.cuh:1-11 (gpu_update_grid) and moves the host passes of
 * IntegratorMetaDynamics.cc:314-588, 663-776, 1053-1155 onto the device
 * ============================================================================================== */

#define MTD_METAD_MAX_CV 6

enum mtd_metad_mode { MTD_MODE_STANDARD = 0, MTD_MODE_WELL_TEMPERED = 1 };

/* gpu_update_grid (IntegratorMetaDynamics.cuh:1-11, .cu:67-93):
 * d_grid_delta[g] += W * scal * exp(-1/2 sum_ij d_i d_j sigma_inv_ij^2) over all num_elements cells,
 * d = node coordinate - *d_current_val.  lengths/cv_min/cv_max/sigma_inv are HOST arrays,
 * d_current_val and d_grid_delta are device doubles. */
int mtd_update_grid(unsigned int num_elements, const unsigned int *lengths, unsigned int dim,
                    const double *d_current_val, double *d_grid_delta, const double *cv_min,
                    const double *cv_max, const double *sigma_inv, double scal, double W,
                    mtd_stream_t stream);

typedef struct mtd_metad mtd_metad; /* opaque; owns the ten grid arrays (IntegratorMetaDynamics.h:276-330) */

/* constructor + registerCollectiveVariable + setGrid(true) + prepRun/setupGrid
 * (IntegratorMetaDynamics.cc:23-72, .h:117-134, .cc:778-815, 121-200, 590-661).
 * Returns MTD_ERR_INVALID_ARGUMENT where the reference throws (cv_min >= cv_max, num_points < 2). */
int mtd_metad_create(mtd_metad **out, unsigned int n_cv, const double *sigma, const double *cv_min,
                     const double *cv_max, const unsigned int *num_points, double W, double T_shift,
                     double T, unsigned int stride, int mode, int add_bias);
int mtd_metad_destroy(mtd_metad *m);

int mtd_metad_set_stride(mtd_metad *m, unsigned int stride);   /* setStride  .h:205 */
int mtd_metad_set_add_hills(mtd_metad *m, int add_bias);       /* setAddHills .h:237 */
int mtd_metad_set_mode(mtd_metad *m, int mode);                /* setMode    .h:197 */
int mtd_metad_set_sigma_inv(mtd_metad *m, const double *sigma_inv /* n_cv^2, host */);
int mtd_metad_reset_histogram(mtd_metad *m, mtd_stream_t stream); /* resetHistogram .cc:1195-1203 */

/* Where the engine takes CV c's current value from (replaces the host
 * `Scalar val = cv->getCurrentValue(timestep)` of .cc:323-327):
 *   s_c = shift + scale * sum_{b<n_partials} d_partials[b*stride + offset]
 * d_partials stays owned by the caller and must stay valid; n_partials may be 1 (a plain value). */
int mtd_metad_set_cv_source(mtd_metad *m, unsigned int cv, const double *d_partials, unsigned int n_partials,
                            unsigned int stride, unsigned int offset, double scale, double shift);

/* CV c's value is a host scalar (box-shape CVs such as AspectRatio / Density): it travels in the
 * kernel arguments of the next update.  This is also the default source (value 0). */
int mtd_metad_set_cv_value(mtd_metad *m, unsigned int cv, double value);

/* device double[n_cv]: dV/ds_c after the most recent mtd_metad_update_bias — what the reference
 * hands to CollectiveVariable::setBiasFactor (.cc:578-584).  Force kernels read it in place. */
const double *mtd_metad_bias_device(const mtd_metad *m);
/* device double[n_cv]: the CV values the engine used in the most recent update */
const double *mtd_metad_cv_device(const mtd_metad *m);

/* updateBiasPotential (.cc:314-588), grid branch, entirely on the device and asynchronous:
 * histogram; on deposit steps (add_bias && timestep % stride == 0) sigma grid, well-tempered scale,
 * Gaussian deposit, reweighted estimator, accumulate + clear; then dV/ds_c (finite difference of
 * the multilinear interpolant), V(s) and the reweighting factor w(s). */
int mtd_metad_update_bias(mtd_metad *m, unsigned int timestep, mtd_stream_t stream);

/* The same split at the multiple-walker exchange (.cc:393-409): phase A fills the four delta arrays
 * (packed contiguously, see mtd_metad_delta_buffers) and returns *deposited; the caller all-reduces
 * them over the walkers (RCCL); phase B reweights, accumulates and evaluates. */
int mtd_metad_update_phase_a(mtd_metad *m, unsigned int timestep, int *deposited, mtd_stream_t stream);
int mtd_metad_update_phase_b(mtd_metad *m, int deposited, mtd_stream_t stream);
/* d_real: {grid_delta[G], sigma_grid_delta[G]} doubles; d_count: {hist_delta[G], hist_gauss_delta[G]} uint32 */
int mtd_metad_delta_buffers(mtd_metad *m, double **d_real, unsigned int **d_count, unsigned int *num_elements);

/* Host-visible state (SYNCHRONISES the stream): what getCurrentValue / the log quantities
 * "bias", "weight" (.h:161-189) report.  Any output pointer may be NULL. */
int mtd_metad_get_state(mtd_metad *m, double *cv, double *bias, double *bias_potential, double *weight,
                        unsigned int *num_gaussians, unsigned int *num_out_of_bounds, mtd_stream_t stream);
double mtd_metad_sigma_determinant(const mtd_metad *m); /* sigmaDeterminant .cc:1296-1313 */
unsigned int mtd_metad_num_elements(const mtd_metad *m);

/* raw grid arrays for writeGrid / readGrid (.cc:831-1000); SYNCHRONISE.  which: 0 grid, 1 grid_delta,
 * 2 reweighted, 3 weight, 4 sigma_grid, 5 sigma_grid_delta (double[G]); 6 hist, 7 hist_delta,
 * 8 hist_gauss, 9 hist_gauss_delta (uint32[G]) */
int mtd_metad_get_array(mtd_metad *m, int which, void *host_out, mtd_stream_t stream);
int mtd_metad_set_array(mtd_metad *m, int which, const void *host_in, mtd_stream_t stream);
int mtd_metad_set_num_gaussians(mtd_metad *m, unsigned int n, mtd_stream_t stream);
/* The device address of one of the arrays above (NULL for a bad index).  A caller that WRITES the bias grid (which = 0) through
 * the pointer must say so after every such write, on the stream the write ran on, with mtd_metad_grid_touched: the engine keeps
 * a small patch of grid values around the last CV values (the scalar chain's preload) which is then rewritten from the grid
 * before its next use.  Fetching the pointer for which = 0 invalidates the patch once, on the NULL stream (kept for callers
 * that write before their next update on that stream); returns NULL when that fails. */
void *mtd_metad_device_array(mtd_metad *m, int which);
int mtd_metad_grid_touched(mtd_metad *m, mtd_stream_t stream);

/* ================================================================================================
 * Fused bias step for lamellar CVs — the headline path (no reference counterpart: it replaces the
 * whole of IntegratorMetaDynamics::updateBiasPotential + the CVs' computeCV / computeBiasForces,
 * IntegratorMetaDynamics.cc:314-588, LamellarOrderParameterGPU.cc:34-132, by TWO launches)
 * ============================================================================================== */

/* Launch A: per-CV partial sums over the particles (as mtd_lamellar_cv_partials) and, in the same
 * launch, the deferred second reweighting pass + accumulate of the previous deposit if one is pending.
 * Between the two passes the caller registers the sums as CV sources (mtd_metad_set_cv_source, once)
 * or, multi-GPU, reduces them (mtd_reduce_partials), all-reduces over RCCL and registers the result. */
int mtd_fused_cv_pass(mtd_metad *m, const mtd_lamellar_set *set, unsigned int n_particles, const void *d_postype,
                      int dtype, const mtd_box *global_box, double *d_partials, unsigned int *n_partials,
                      mtd_stream_t stream);

/* Launch B: updateBiasPotential(timestep) for the CV values defined by the registered sources
 * (histogram; on deposit steps sigma grid, well-tempered scale, Gaussian increment, first reweighting
 * pass; dV/ds_c, V(s)) and, in the same launch, the bias forces of every CV of the set
 * (CV c of the set must be CV c of the grid).  The second reweighting pass + accumulate stay pending
 * until the next mtd_fused_cv_pass or any call that reads the grid (get_state / get_array flush it). */
int mtd_fused_force_pass(mtd_metad *m, const mtd_lamellar_set *set, unsigned int n_particles, const void *d_postype,
                         void *const *d_force, int dtype, unsigned int n_global, const mtd_box *global_box,
                         unsigned int timestep, mtd_stream_t stream);

/* The same launch for a MIXED set of collective variables: `set` holds the lamellar CVs among the grid's variables,
 * slots[c] is the grid variable behind CV c of the set (any order, each < the grid's n_cv); the other variables' values
 * arrive through their registered sources as in mtd_metad_update_bias, which this call replaces for the step, and their
 * force kernels read mtd_metad_bias_device afterwards.  slots == NULL: identity (set->n_cv must equal the grid's). */
int mtd_fused_force_pass_slots(mtd_metad *m, const mtd_lamellar_set *set, const unsigned int *slots, unsigned int n_particles,
                               const void *d_postype, void *const *d_force, int dtype, unsigned int n_global,
                               const mtd_box *global_box, unsigned int timestep, mtd_stream_t stream);

/* The whole step in ONE launch (fused_step.hip): a persistent kernel, one 1024-thread block per compute unit, keeps its
 * particles in registers between the CV phase and the force phase (the positions are read once) and hands the two grid-wide
 * sums of a step (CV sums; sum R dV, sum R of the reweighting) from block to block inside the launch in the mailbox's wire
 * format.  Both reweighting passes run in the launch: the grid arrays are final when it ends (nothing stays pending).
 * Sharded run (mtd_metad_set_comm): block 0 sends the local sums, every block's chain polls the local mailbox.
 * Envelope: n_cv <= 3, CV c of the set is CV c of the grid, at most 4096 particles per compute unit, the whole grid of
 * blocks resident at once, and the form selected (mtd_fused_step_set_mode); otherwise the call runs mtd_fused_cv_pass + mtd_fused_force_pass with
 * d_scratch (>= mtd_lamellar_scratch_doubles) as the partial-sum buffer.  Every in-launch wait is bounded
 * (MTD_COMM_TIMEOUT_MS): an expired one poisons the step (NaN) and later calls return MTD_ERR_COMM_TIMEOUT.
 * Replaces per step: LamellarOrderParameterGPU.cc:34-132 for every CV + IntegratorMetaDynamics.cc:314-588. */
int mtd_fused_step(mtd_metad *m, const mtd_lamellar_set *set, unsigned int n_particles, const void *d_postype,
                   void *const *d_force, int dtype, unsigned int n_global, const mtd_box *global_box, double *d_scratch,
                   unsigned int timestep, mtd_stream_t stream);
/* launches the last mtd_fused_step of this engine took: 1 (persistent kernel) or 2 (two-launch form); 0 before the first */
unsigned int mtd_fused_step_launches(const mtd_metad *m);
/* which form mtd_fused_step takes for this engine: 1 the persistent kernel wherever its envelope allows, 0 always the
 * two-launch form, -1 (default) the environment (MTD_FUSED_STEP=1 / 0) and without it the two-launch form, which measures
 * faster at 10^6 particles */
int mtd_fused_step_set_mode(mtd_metad *m, int mode);

/* Measurement aid (bench.py): the next n_launches launches of mtd_fused_force_pass record their own begin and end through the
 * start / stop events of hipExtLaunchKernelGGL — the dispatch's time stamps, i.e. the duration a kernel trace reports for the
 * launch, without a profiler attached.  mtd_profile_force_end SYNCHRONISES, returns the durations (microseconds) of the
 * launches made since _begin and releases the events. */
int mtd_profile_force_begin(unsigned int n_launches);
int mtd_profile_force_end(double *durations_us, unsigned int capacity, unsigned int *n_out);

/* ================================================================================================
 * RCCL all-reduce of large device buffers (rccl.hip; RCCL is bound at run time, no link-time dependency)
 * replaces: the multiple-walker exchange IntegratorMetaDynamics.cc:393-409 (four host-staged MPI_Allreduce over
 * m_partition_comm) and, for a particle-sharded mesh CV, the mesh exchange of OrderParameterMesh.cc:263-316, 659-746.
 * ============================================================================================== */
#define MTD_RCCL_ID_BYTES 128
#define MTD_ELEM_F64 1
#define MTD_ELEM_U32 2
typedef struct mtd_rccl mtd_rccl;
/* rank 0: a fresh unique id (MTD_RCCL_ID_BYTES bytes) for the caller's control plane to hand to every rank */
int mtd_rccl_unique_id(void *out_id);
/* collective over the `world` ranks holding the same id; one communicator per process and GPU */
int mtd_rccl_create(mtd_rccl **out, const void *unique_id, unsigned int rank, unsigned int world);
/* in-place sum over ranks of `count` elements (MTD_ELEM_F64 doubles or MTD_ELEM_U32 unsigned ints) on `stream` */
int mtd_comm_allreduce_large(mtd_rccl *r, void *d_buffer, size_t count, int elem, mtd_stream_t stream);
unsigned int mtd_rccl_world(const mtd_rccl *r);
unsigned int mtd_rccl_rank(const mtd_rccl *r);
int mtd_rccl_destroy(mtd_rccl *r);
/* text of the last RCCL failure of this process ("ncclAllReduce: unhandled system error", ...), "" when there was none */
const char *mtd_rccl_last_error(void);
/* Multiple walkers in one call (IntegratorMetaDynamics.cc:363-451 with m_multiple_walkers): histogram / Gaussian increments
 * of this walker (mtd_metad_update_phase_a), sum of {grid_delta, sigma_grid_delta} and {hist_delta, hist_gauss_delta} over
 * the walkers (two all-reduces: the delta groups are contiguous), reweighting + accumulate + evaluation
 * (mtd_metad_update_phase_b).  Every walker must call it with the same timestep.  walkers == NULL: one walker (the reference in
 * a run with a single partition, as test/test_2d.py:29 does). */
int mtd_metad_update_bias_walkers(mtd_metad *m, mtd_rccl *walkers, unsigned int timestep, mtd_stream_t stream);

/* ================================================================================================
 * xGMI mailbox: all-reduce (sum) of a few doubles between the GPUs of one node, one process per GPU
 * replaces the host-staged MPI_Allreduce of the per-step CV sums (LamellarOrderParameterGPU.cc:69-77,
 * SteinhardtQl.cc:183-191, WellTemperedEnsemble.cc:57-63) without a collective-library call on the
 * critical path: every rank's kernels store their values straight into the peers' mailboxes over xGMI
 * (hipIpc-mapped uncached device memory) and poll their own.  Large buffers (replicated mesh, packed
 * walker deltas) stay on RCCL.  Every wait is bounded: a peer that never arrives is a counted timeout
 * (MTD_COMM_TIMEOUT_MS, default 5000), not a hang.
 * ============================================================================================== */

#define MTD_COMM_MAX_RANKS 8     /* the GPUs of one node */
#define MTD_COMM_HANDLE_BYTES 64 /* sizeof(hipIpcMemHandle_t) */

typedef struct mtd_comm mtd_comm; /* opaque; owns this rank's mailbox and the mappings of the peers' */

/* max_doubles: largest message (per rank) in doubles, <= 4096.  SYNCHRONISES (allocation + clear). */
int mtd_comm_create(mtd_comm **out, unsigned int rank, unsigned int world, unsigned int max_doubles);
/* the IPC handle of this rank's mailbox (MTD_COMM_HANDLE_BYTES bytes) for the caller's control plane to gather */
int mtd_comm_handle(mtd_comm *c, void *out_handle);
/* handles: world * MTD_COMM_HANDLE_BYTES bytes in rank order (the own entry is ignored).  world == 1 needs no connect. */
int mtd_comm_connect(mtd_comm *c, const void *handles);
/* d_values[0..n) <- sum over ranks, added up in rank order (the same bits on every rank).  Every rank must issue the
 * same sequence of exchanges (this call and the fused step's) on its comm. */
int mtd_comm_allreduce_small(mtd_comm *c, double *d_values, unsigned int n, mtd_stream_t stream);
/* number of timed-out waits so far; SYNCHRONISES the stream */
int mtd_comm_status(mtd_comm *c, unsigned int *timeouts, mtd_stream_t stream);
/* Bulk buffers readable by every rank (the slab-decomposed mesh pulls its peers' slabs straight out of them over xGMI):
 * mtd_comm_share allocates `bytes` of uncached device memory on this rank (peers' reads and this GPU's writes must not
 * sit in a non-coherent L2) and returns its address, its slot number and its IPC handle; after the caller's control plane
 * has gathered the handles of that slot from all ranks, mtd_comm_open maps them: peers[r] is rank r's buffer as seen from
 * this process (peers[rank] = the local address).  At most MTD_COMM_MAX_SHARED buffers; released by mtd_comm_destroy. */
#define MTD_COMM_MAX_SHARED 8
int mtd_comm_share(mtd_comm *c, size_t bytes, void **d_local, unsigned int *slot, void *out_handle);
int mtd_comm_open(mtd_comm *c, unsigned int slot, const void *handles, void **peers);
/* Large all-reduce (sum, doubles, in place) through the mailbox's exported buffers instead of a collective library — for the
 * MB-sized per-step buffers of a domain-decomposed run (the replicated mesh of cv.mesh: M + 1 doubles; replaces the ghost-cell
 * exchange + distributed FFT of OrderParameterMesh.cc:263-316, 659-746 and the MPI_Allreduce of :630) where no RCCL communicator
 * is at hand (mtd_comm_allreduce_large is the RCCL form).  Every rank stages its buffer in an exported one, rank r sums slice r
 * of all ranks' staging buffers in rank order by remote loads (a reduce-scatter), every rank copies all slices home (an
 * all-gather); two mailbox exchanges are the barriers.  Same bits on every rank; an expired wait poisons the whole result (NaN)
 * and the communicator (MTD_ERR_COMM_TIMEOUT from then on).
 * Set-up: two buffers of mtd_comm_pull_bytes(max_doubles) bytes per rank from mtd_comm_share, opened by every rank
 * (mtd_comm_open), handed over once with mtd_comm_pull_attach (in_peers / out_peers: rank r's buffer as mapped here). */
size_t mtd_comm_pull_bytes(size_t max_doubles);
int mtd_comm_pull_attach(mtd_comm *c, size_t max_doubles, void *const *in_peers, void *const *out_peers);
int mtd_comm_allreduce_pull(mtd_comm *c, double *d_buffer, size_t count, mtd_stream_t stream);
unsigned int mtd_comm_world(const mtd_comm *c);
unsigned int mtd_comm_rank(const mtd_comm *c);
int mtd_comm_destroy(mtd_comm *c);

/* Particle-sharded fused step: with a comm attached, every CV block of mtd_fused_cv_pass also posts its sums for a collector
 * (the first CV block), which adds them up in a fixed order and sends the n_cv totals to every rank, and
 * mtd_fused_force_pass's scalar chain takes
 * s_c = scale_c * (sum over ranks) + shift_c from the mailbox instead of the registered partial sums (scale / shift
 * as registered with mtd_metad_set_cv_source): still two launches per step, no collective call.  n_cv <= 3.
 * comm == NULL detaches. */
int mtd_metad_set_comm(mtd_metad *m, mtd_comm *comm);

/* ================================================================================================
 * Particle-mesh order parameter (cv.mesh)
 * replaces OrderParameterMeshGPU.cuh:9-109 (gpu_bin_particles, gpu_assign_binned_particles_to_mesh,
 * gpu_update_meshes, gpu_compute_cv, gpu_compute_forces, gpu_compute_mode_sq) and the cuFFT plans of
 * OrderParameterMeshGPU.cc:57-152; single rank (no ghost cells), double precision meshes
 * ============================================================================================== */

typedef struct mtd_mesh mtd_mesh; /* opaque; owns the meshes, the cell list and the FFT twiddles */

/* OrderParameterMesh constructor + setupMesh + initializeFFT (OrderParameterMesh.cc:18-122, 191-229, 263-342).
 * mode: host double[n_types].  Mesh points per axis: 4 ... 256 (any), or a power of two up to 1024 (MTD_ERR_UNSUPPORTED otherwise). */
int mtd_mesh_create(mtd_mesh **out, unsigned int nx, unsigned int ny, unsigned int nz, const double *mode,
                    unsigned int n_types, unsigned int max_particles);
int mtd_mesh_destroy(mtd_mesh *m);
/* 1 (default): interpolation function with the reference's unsigned integer division; 0: as intended */
int mtd_mesh_set_bug_compat(mtd_mesh *m, int on);
/* The normalised Fourier mesh f = FFT(mesh) / N (fourier_mesh of the reference, OrderParameterMesh.cc:697-712) is only read by
 * mtd_mesh_qmax, mtd_mesh_virial and mtd_mesh_get_array(1): the spectral step writes it when `on` (default 1; 18.9 MB per
 * step at 128^3 otherwise saved).  Those three calls return MTD_ERR_INVALID_ARGUMENT when the last spectral step kept none:
 * switch it on and run mtd_mesh_spectral again (the real mesh of the step is still in place). */
int mtd_mesh_set_keep_fourier(mtd_mesh *m, int on);
/* Overlap hook: when `hip_event` (a hipEvent_t, NULL to clear) is set, mtd_mesh_spectral / mtd_mesh_compute_cv record it on
 * their stream right after the pass that completes the CV partial sums (the fused z pass) — the two inverse passes that
 * follow only feed the force pass.  A caller makes the launch that consumes the CV (the grid engine) wait for this event on
 * ANOTHER stream, so that it runs beside the inverse passes (host classes: mixed CV sets). */
int mtd_mesh_set_cv_event(mtd_mesh *m, void *hip_event);
unsigned int mtd_mesh_num_cells(const mtd_mesh *m);

/* getCurrentValue (OrderParameterMesh.cc:925-968): assignParticles -> FFT -> updateMeshes -> iFFT -> computeCV.
 * The CV is s = 1/2 * sum_{b < *n_partials} (*d_partials)[b]  (device doubles owned by the mesh): consume with
 * mtd_metad_set_cv_source(engine, slot, *d_partials, *n_partials, 1, 0, 0.5, 0.0) or mtd_reduce_partials. */
int mtd_mesh_compute_cv(mtd_mesh *m, unsigned int n_particles, const void *d_postype, int dtype, const mtd_box *box,
                        unsigned int n_global, const double **d_partials, unsigned int *n_partials, mtd_stream_t stream);

/* The two halves of mtd_mesh_compute_cv for a particle-sharded system with a replicated mesh (replaces the
 * ghost-cell exchange + distributed FFT of OrderParameterMesh.cc:263-316, 659-746 and the mode_sq all-reduce of :630):
 * _assign spreads this rank's particles, the ranks all-reduce (sum) the *count doubles at *d_buffer — the real mesh
 * followed by the sum of mode^2 — and _spectral runs FFT -> updateMeshes -> iFFT -> computeCV on the reduced mesh. */
/* Riders on the NEXT mtd_mesh_compute_cv / mtd_mesh_assign of this mesh (one-shot, tile pipeline only — MTD_ERR_UNSUPPORTED
 * otherwise, then call mtd_fused_cv_pass): the kernel that bins the particles also forms, from the positions it holds anyway, the
 * block partial sums of the lamellar CVs of `set` (at most 3; what mtd_fused_cv_pass leaves in d_partials: feed them to
 * mtd_metad_set_cv_source with n_partials = *n_partials, stride = set->n_cv, offset = c, scale 1 / N_global), and the assignment's
 * launches carry the deferred second grid pass of `engine`'s previous deposit (may be NULL).  In the bin pipeline (every assignment
 * of a mesh but its first) the sums are formed while the binning blocks wait for their atomics and the grid pass travels as extra
 * blocks of the scatter launch; the counting pipeline forms them in its particle loop / carries the pass in its row-scan launch.  A
 * mixed set (cv.lamellar + cv.mesh: LamellarOrderParameter.cc:143-179 and OrderParameterMesh.cc:517-640 both stream the positions)
 * then reads the positions once for both and spends one launch less per step.  n_particles, the stream and the position array of
 * that next call must be the ones the CVs share; d_partials: mtd_lamellar_scratch_doubles(n_particles) doubles. */
int mtd_mesh_set_lamellar_rider(mtd_mesh *mesh, mtd_metad *engine, const mtd_lamellar_set *set, const mtd_box *global_box,
                                unsigned int n_particles, double *d_partials, unsigned int *n_partials, mtd_stream_t stream);
/* disarm riders that no assignment has consumed (a caller whose mesh CV turned out to be up to date for this step);
 * *was_armed (may be NULL): 1 when they were still waiting — their sums were then NOT formed */
int mtd_mesh_clear_rider(mtd_mesh *mesh, int *was_armed);
int mtd_mesh_assign(mtd_mesh *m, unsigned int n_particles, const void *d_postype, int dtype, const mtd_box *box, mtd_stream_t stream);
/* How the last assignment (mtd_mesh_assign / mtd_mesh_compute_cv / mtd_mesh_slab_*) grouped the particles by tile; waits for `stream`.
 * *pipeline: 0 the cell-level pipeline (meshes beyond 256^3, MTD_MESH_ASSIGN=cells), 1 the counting pipeline (count -> row scan ->
 * place: the first assignment of a mesh, a changed particle number, riders), 2 the bin pipeline (one launch on tile segments with
 * slack planned from the previous snapshot's exact counts).  *n_overflow: particles of a bin step that did not fit
 * their tile's planned segment and travelled through the overflow list (correct, only slower; 0 in a well-planned step). */
int mtd_mesh_assign_info(mtd_mesh *m, int *pipeline, unsigned int *n_overflow, mtd_stream_t stream);
/* The end of a step of a MIXED set — one mesh variable (grid variable `mesh_slot`) and the lamellar CVs of `set` (grid variables
 * slots[0 .. set->n_cv)) — in ONE launch: what mtd_fused_force_pass_slots (scalar chain on the CV sums -> bias factors, first grid
 * pass, lamellar forces) followed by mtd_mesh_forces with the mesh's device bias factor do in two (OrderParameterMesh.cc:925-968,
 * IntegratorMetaDynamics.cc:329-588, LamellarOrderParameterGPU.cu:60-118).  The chain runs in wave 0 of every block of the mesh's
 * force pass while the other waves stage their tile; every block streams a share of the particles for the lamellar forces and the
 * first blocks take the grid pass.  Same results as the two calls (engine state and lamellar forces to the last bits of a
 * contraction, mesh forces bitwise).  MTD_ERR_UNSUPPORTED where the shapes do not allow it (cell-level mesh pipeline, an engine with a
 * mailbox, more than three grid variables, more 256-cell grid blocks than mesh tiles, MTD_MESH_FORCE_MERGED=0): make the two calls.
 * set == NULL (slots, d_force_lamellar ignored): no lamellar CVs — the mesh variable alone on the grid, or beside variables of other
 * kinds whose sources are registered: mtd_metad_update_bias + mtd_mesh_forces in one launch. */
int mtd_mesh_forces_update_bias(mtd_mesh *mesh, mtd_metad *engine, unsigned int mesh_slot, const mtd_lamellar_set *set,
                                const unsigned int *slots, unsigned int n_particles, const void *d_postype, void *d_force_mesh,
                                void *const *d_force_lamellar, int dtype, unsigned int n_global, const mtd_box *global_box,
                                unsigned int timestep, mtd_stream_t stream);
/* Which kernels ran the last forward transform (mtd_mesh_spectral / mtd_mesh_compute_cv): 0 separate x and y passes (meshes whose
 * planes do not fit the LDS, sizes that are not powers of two), 1 x and y of a plane in one launch on the combined mesh, 2 the same
 * launch reading the assignment's per-tile images itself (meshes 128 cells wide after mtd_mesh_compute_cv: no combine launch;
 * MTD_FFT_FROM_TILES=0 switches it off).  The results are the same bits in all three. */
int mtd_mesh_transform_info(mtd_mesh *m, int *forward);
int mtd_mesh_exchange_buffer(mtd_mesh *m, double **d_buffer, size_t *count);
int mtd_mesh_spectral(mtd_mesh *m, const mtd_box *box, unsigned int n_global, const double **d_partials, unsigned int *n_partials,
                      mtd_stream_t stream);

/* interpolateForces (OrderParameterMesh.cc:749-864) from the inverse mesh AND the cell-sorted particle records of the last
 * mtd_mesh_compute_cv, i.e. of the same snapshot (the reference recomputes the CV first when needed, :1055-1056);
 * bias = *d_bias when d_bias != NULL (device resident), else bias_host */
int mtd_mesh_forces(mtd_mesh *m, unsigned int n_particles, const void *d_postype, void *d_force, int dtype,
                    const mtd_box *box, unsigned int n_global, const double *d_bias, double bias_host, mtd_stream_t stream);

/* setTable (OrderParameterMesh.cc:148-189): convolution kernel K(k) and its derivative on n equidistant points of
 * [k_min, k_max].  Like the reference, K itself is stored and never applied to the mesh; K' enters the virial. */
int mtd_mesh_set_table(mtd_mesh *m, const double *K, const double *d_K, unsigned int n, double k_min, double k_max);
int mtd_mesh_set_use_table(mtd_mesh *m, int use_table);

/* computeQmax (:1108-1179) on the Fourier mesh of the last compute_cv / spectral call (SYNCHRONISES):
 * out[0..2] = wave vector of the cell with the largest |f|^2 (first cell wins ties, DC bin included), out[3] = that
 * amplitude times N_global — the log quantities qx_max, qy_max, qz_max, sq_max. */
int mtd_mesh_qmax(mtd_mesh *m, const mtd_box *box, unsigned int n_global, double *out, mtd_stream_t stream);

/* computeVirial (:970-1050) (SYNCHRONISES): virial[6] (xx, xy, xz, yy, yz, zz) = bias * sum_{k != 0} |f|^4 / N^2 *
 * K'(|k|) / (2 |k|) * k_a k_b — identically zero without a table in use, exactly like the reference. */
int mtd_mesh_virial(mtd_mesh *m, const mtd_box *box, unsigned int n_global, double bias, double *virial, mtd_stream_t stream);

/* raw arrays for parity tests (SYNCHRONISES): which = 0 real mesh double[M]; 1 fourier_mesh (normalised) complex double[M];
 * 3 Re(inv_fourier_mesh) double[M] (the imaginary part is never used, OrderParameterMesh.cc:851-857); 7 sum of mode^2 */
int mtd_mesh_get_array(mtd_mesh *m, int which, void *host_out, mtd_stream_t stream);


/* Slab decomposition of the mesh over the ranks of an xGMI mailbox (replaces the ghost-cell exchange and the
 * distributed FFT of OrderParameterMesh.cc:263-316, 659-746 — dfftlib over MPI — for meshes whose replicated transform no
 * longer pays).  Rank r owns the z planes [r nz/W, (r+1) nz/W) for the x and y passes and the y rows [r ny/W, (r+1) ny/W)
 * for the z pass; the transposes between them are remote loads out of four buffers every rank exports
 * (mtd_comm_share, sizes from mtd_mesh_slab_bytes: local assignment, transformed slab, pencils of G, slab of Re(inv)).
 * nz and ny must be multiples of the number of ranks.  mtd_mesh_slab_compute_cv runs the whole forward / spectral / inverse
 * sequence with four mailbox exchanges as barriers (one carries sum mode^2, one the CV sum) and leaves the complete Re(inv)
 * on every rank for mtd_mesh_forces; *d_cv_sum is a device double, the CV is half of it.  The normalised Fourier mesh
 * (log quantities, virial) is not kept in slab runs. */
int mtd_mesh_slab_bytes(const mtd_mesh *m, unsigned int world, size_t *bytes /* [4] */);
int mtd_mesh_slab_attach(mtd_mesh *m, mtd_comm *comm, void *const *rho_peers, void *const *f_peers, void *const *g_peers,
                         void *const *inv_peers);
int mtd_mesh_slab_compute_cv(mtd_mesh *m, unsigned int n_particles, const void *d_postype, int dtype, const mtd_box *box,
                             unsigned int n_global, const double **d_cv_sum, mtd_stream_t stream);
/* ================================================================================================
 * Steinhardt Q_l (cv.steinhardt) — the reference has only a host implementation (SteinhardtQl.cc); these entry points
 * are what a GPU class of it would call.  Neighbour list in HOOMD's layout (NeighborList::getHeadList / getNNeighArray /
 * getNListArray, device uint32 arrays); half_nlist: 0 = storage mode full, 1 = storage mode half (third-law path,
 * SteinhardtQl.cc:80, 173-179, 328-333), 2 = a full list that is symmetric ((i, j) listed <=> (j, i) listed, as HOOMD builds
 * them) and indexes no ghost particle: the CV pass then visits every pair once, from its lower index — Y_lm(-d) = (-1)^l
 * Y_lm(d), so the reference's two visits add up to twice the even degrees and cancel in the odd ones, the scaling of :173-179;
 * the force pass treats 2 like 0.  Not for mtd_ql_accumulate_local with ghost particles.
 * ============================================================================================== */

/* device doubles the two calls share (block partial sums, Q_lm tables, Q_l, CV value) */
size_t mtd_ql_scratch_doubles(unsigned int lmax);

/* SteinhardtQl::computeCV (SteinhardtQl.cc:62-201).  Ql_ref: host double[lmax+1].  On return *d_value points at the CV value,
 * *d_Ql at Q_l[lmax+1], *d_Qlm at the complex Q_lm[(lmax+1)^2] table in the reference's order (all inside d_scratch);
 * consume the value with mtd_metad_set_cv_source(engine, slot, *d_value, 1, 1, 0, 1.0, 0.0).  lmax <= 12. */
int mtd_ql_accumulate(unsigned int n_particles, const void *d_postype, int dtype, const mtd_box *box, const unsigned int *d_head_list,
                      const unsigned int *d_n_neigh, const unsigned int *d_nlist, int half_nlist, double rcut, double ron,
                      unsigned int lmax, unsigned int type, const double *Ql_ref, unsigned int n_global, double *d_scratch,
                      const double **d_value, const double **d_Ql, const double **d_Qlm, mtd_stream_t stream);

/* The two halves of mtd_ql_accumulate for a particle-sharded system (the reference class has no MPI path):
 * _local leaves this rank's sums Q'_lm (m >= 0, (lmax+1)(lmax+2) doubles) at *d_sums inside d_scratch — the ranks
 * all-reduce exactly that buffer — and _finalize turns the reduced sums into Q_lm, Q_l and the CV value.
 * Neighbour indices >= n_particles address ghost particles stored behind the local ones in d_postype. */
int mtd_ql_accumulate_local(unsigned int n_particles, const void *d_postype, int dtype, const mtd_box *box,
                            const unsigned int *d_head_list, const unsigned int *d_n_neigh, const unsigned int *d_nlist, int half_nlist,
                            double rcut, double ron, unsigned int lmax, unsigned int type, unsigned int n_global, double *d_scratch,
                            double **d_sums, unsigned int *n_sums, mtd_stream_t stream);
int mtd_ql_finalize(int half_nlist, unsigned int lmax, const double *Ql_ref, unsigned int n_global, double *d_scratch,
                    const double **d_value, const double **d_Ql, const double **d_Qlm, mtd_stream_t stream);

/* cv.steinhardt as the ONLY variable of a bias grid: mtd_ql_finalize and mtd_metad_update_bias(engine, timestep) in ONE launch
 * (SteinhardtQl.cc:173-199 + IntegratorMetaDynamics.cc:314-588): every block of the grid engine's launch forms Q_lm, Q_l and the
 * value from the sums mtd_ql_accumulate_local left in d_scratch (all-reduced by the caller in a particle-sharded run) and hands the
 * value to the engine's scalar chain in registers; the value is also registered as the variable's source.  The engine's deferred
 * pass of the deposit then travels in the next mtd_ql_forces on the same stream (full lists) instead of a launch of its own.
 * MTD_ERR_UNSUPPORTED when the grid has more than one variable or a mailbox attached (call mtd_ql_finalize +
 * mtd_metad_update_bias then).  Between the two pair passes of a step this leaves two launches where there were three. */
int mtd_ql_finalize_update_bias(mtd_metad *engine, int half_nlist, unsigned int lmax, const double *Ql_ref, unsigned int n_global,
                                double *d_scratch, unsigned int timestep, const double **d_value, const double **d_Ql,
                                const double **d_Qlm, mtd_stream_t stream);

/* SteinhardtQl::computeBiasForces (:203-339) with the Q_lm the last mtd_ql_accumulate left in d_scratch (Q20);
 * bias = *d_bias when d_bias != NULL, else bias_host.  Writes d_force[0..n_particles); with half lists the reaction force goes
 * to LOCAL partners only (j < n_particles, SteinhardtQl.cc:328), so particle-sharded runs use full lists. */
/* Half lists: how the reaction forces of the pairs are summed into the partner particles (SteinhardtQl.cc:328-333; the
 * reference's serial loop has one order, a GPU has none).  1 (default): as exact integers — bitwise reproducible, independent
 * of the order; 0: floating-point atomics — sums in arrival order, about 2.5 x faster.  Full lists need neither. */
int mtd_ql_set_half_list_exact(int enable);

/* Half lists without atomics.  Builds — once per neighbour-list update, synchronous — the symmetric full list a half list stands
 * for: every pair at both of its particles, each particle's partners in ascending order (so the result does not depend on any
 * arrival order), into caller-provided device arrays d_full_head[n], d_full_n_neigh[n], d_full_nlist[full_capacity].
 * *n_full_entries (host) receives the number of entries (twice the pairs); when it exceeds full_capacity nothing is written and
 * MTD_ERR_INVALID_ARGUMENT comes back (call again with room).  Then pass the full arrays with half_nlist = 2 to
 * mtd_ql_accumulate / mtd_ql_forces: the CV pass visits every pair once and scales like the half-list branch
 * (SteinhardtQl.cc:173-179), the force pass gathers like the full-list pass — the third-law sum of :328-333 with no atomics,
 * bitwise reproducible, at the full-list pass's cost (two pair visits instead of one visit + atomics).  A half list that
 * indexes ghost particles (j >= n_particles) is refused (MTD_ERR_UNSUPPORTED): its reaction forces would be lost (:328). */
int mtd_ql_symmetrize_half_list(unsigned int n_particles, const unsigned int *d_head_list, const unsigned int *d_n_neigh,
                                const unsigned int *d_nlist, unsigned int *d_full_head, unsigned int *d_full_n_neigh,
                                unsigned int *d_full_nlist, size_t full_capacity, size_t *n_full_entries, mtd_stream_t stream);
/* The same with a caller-owned workspace of mtd_ql_symmetrize_workspace_uints(n_particles) device unsigned ints (a host class
 * keeps it across the list updates of a run): no allocation, ONE synchronisation (the entry count must reach the host before the
 * arrays can be trusted); the fill itself is only stream-ordered. */
size_t mtd_ql_symmetrize_workspace_uints(unsigned int n_particles);
int mtd_ql_symmetrize_half_list_ws(unsigned int n_particles, const unsigned int *d_head_list, const unsigned int *d_n_neigh,
                                   const unsigned int *d_nlist, unsigned int *d_full_head, unsigned int *d_full_n_neigh,
                                   unsigned int *d_full_nlist, size_t full_capacity, size_t *n_full_entries, unsigned int *d_workspace,
                                   mtd_stream_t stream);

int mtd_ql_forces(unsigned int n_particles, const void *d_postype, void *d_force, int dtype, const mtd_box *box,
                  const unsigned int *d_head_list, const unsigned int *d_n_neigh, const unsigned int *d_nlist, int half_nlist,
                  double rcut, double ron, unsigned int lmax, unsigned int type, const double *Ql_ref, unsigned int n_global,
                  const double *d_scratch, const double *d_bias, double bias_host, mtd_stream_t stream);

/* ================================================================================================
 * WellTemperedEnsemble (potential energy as CV)
 * replaces WellTemperedEnsemble.cuh:3-19 (gpu_scale_netforce, gpu_reduce_potential_energy)
 * ============================================================================================== */

size_t mtd_wte_scratch_doubles(unsigned int n_particles);

/* gpu_reduce_potential_energy (.cu:195-243): block partial sums of net_force.w.
 * PE = external_energy + sum_b d_partials[b]; consume with mtd_metad_set_cv_source / mtd_reduce_partials */
int mtd_wte_energy_partials(unsigned int n_particles, const void *d_net_force, int dtype,
                            double *d_partials, unsigned int *n_partials, mtd_stream_t stream);

/* gpu_scale_netforce (.cu:53-89): net_force.xyz, net_torque.xyz and the six net_virial rows are
 * multiplied by (1 + bias); bias = *d_bias when d_bias != NULL (device resident), else bias_host.
 * scale_torque_w != 0 also scales net_torque.w like the reference CPU path (WellTemperedEnsemble.cc:169). */
int mtd_wte_scale_netforce(unsigned int n_particles, void *d_net_force, void *d_net_torque,
                           void *d_net_virial, unsigned int virial_pitch, int dtype, const double *d_bias,
                           double bias_host, int scale_torque_w, mtd_stream_t stream);

/* ================================================================================================
 * CollectiveWrapper (energy of an arbitrary ForceCompute as CV)
 * replaces CollectiveWrapper.cc:74-134 (computeCVGPU / computeBiasForcesGPU, which reuse the WTE drivers)
 * ============================================================================================== */

/* energy = external_energy + sum_j force_j.w: use mtd_wte_energy_partials on the wrapped force array.
 * computeBiasForces (:125, :153): force.xyz, torque.xyz(w) and the six virial rows of the WRAPPED compute are
 * multiplied by the bias factor itself (not 1 + bias as in the WTE) */
int mtd_wrapper_scale_forces(unsigned int n_particles, void *d_force, void *d_torque, void *d_virial,
                             unsigned int virial_pitch, int dtype, const double *d_bias, double bias_host,
                             int scale_torque_w, mtd_stream_t stream);

/* ================================================================================================
 * Adaptive Gaussians
 * replaces IntegratorMetaDynamics::computeSigma (IntegratorMetaDynamics.cc:1205-1294)
 * ============================================================================================== */

size_t mtd_sigma_scratch_doubles(void);

/* sigmasq[i*n_cv+j] = sigma_g^2 * sum_n f_i(n).f_j(n) over this rank's particles (:1241-1246), for the CVs whose
 * d_force[c] != NULL (those that can compute derivatives); all other entries come back 0 and are the caller's
 * (:1249: diagonal sigma^2 on the root rank).  Synchronous: sigmasq is a HOST array of n_cv^2 doubles; in a
 * domain-decomposed run it is all-reduced before mtd_sigma_inverse (:1259-1268). */
int mtd_sigma_products(unsigned int n_cv, const void *const *d_force, unsigned int n_particles, int dtype, double sigma_g,
                       double *d_scratch, double *sigmasq, mtd_stream_t stream);

/* sigma_inv = inverse(element-wise sqrt(sigmasq)) (:1273-1286), host arrays of n_cv^2 doubles */
int mtd_sigma_inverse(unsigned int n_cv, const double *sigmasq, double *sigma_inv);

/* ================================================================================================
 * Diagnostic exports: the two pieces of device code whose reference counterparts COULD be compiled (oracle/_ref), exposed so
 * that tests hold the kernels' own arithmetic — not only the oracle's — against the vectors the reference code produced
 * (tests/golden/, tests/test_gpu_golden.py).  Host arrays in and out, synchronous; not part of the hot path.
 * ============================================================================================== */

/* Y_lm of n directions (h_separations: n x 3, any length) as the Steinhardt pair kernels evaluate them; h_out:
 * n x (lmax+1)^2 x (re, im) in fsph's order (spherical_harmonics.hpp:229-246 with full_m) */
int mtd_debug_sph_harmonics(unsigned int lmax, unsigned int n, const double *h_separations, double *h_out);
/* IndexGrid::getCoordinates (IndexGrid.cc:46-58) and getIndex (:20-44) as the grid kernels compute them */
int mtd_debug_index_decode(unsigned int n_cv, const unsigned int *lengths, unsigned int n, const unsigned int *h_indices,
                           unsigned int *h_coords, unsigned int *h_index_back);
/* The agreement check of mtd_metad_update_bias_walkers as two pure-host functions (no device needed; tests/test_abi_exports.py):
 * _pack writes the 12 doubles a walker contributes for (stride, add_bias, timestep) — each quantity split into 16-bit halves,
 * every half followed by its square, so that sums over <= 2^16 walkers stay exact integers below 2^53 whatever the order of
 * the reduction; _verify returns 1 when `sums` (the all-reduced 12 doubles) say that all `world` walkers sent what `mine` holds. */
void mtd_debug_walker_check_pack(unsigned int stride, int add_bias, unsigned int timestep, double *out12);
int mtd_debug_walker_check_verify(const double *sums12, const double *mine12, unsigned int world);

#ifdef __cplusplus
}
#endif
#endif /* MTD_ABI_H */
