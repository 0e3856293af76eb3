/* oracle/mtd_ref.h — CPU restatement of the reference's metadynamics hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / reported CPU baseline.  The product path (metadynamics-plugin_amd/) never links
 * or calls it and fails loudly when its HIP library is missing.
 *
 * PARITY PINNING.  The reference (jglaser/metadynamics-plugin) ships no golden vectors, no
 * asserts and no unit tests (test/test_2d.py, test/test_mesh.py are eyeball scripts that need a
 * full HOOMD-blue v2 install, which is absent here).  Only IndexGrid.cc and the header-only
 * spherical_harmonics.hpp compile standalone; they are built from the sources where they lie
 * into oracle/_ref/ (see oracle/Makefile, oracle/ref_shim.cc) and pin ref_index_* and ref_sph_*
 * below.  Every other function in this file is a statement-by-statement restatement of the cited
 * .cc lines, pinned only by analytic known-answer tests (tests/test_oracle_*.py):
 *     ==> lamellar, bias grid, WTE, mesh: "parity unpinned" (no reference-run vectors exist).
 *
 * All arithmetic is double precision ("Scalar" = double, HOOMD's default build) and serial,
 * exactly like the reference CPU path (which has no threading, only MPI ranks).
 * Citations are file:line relative to /root/reference/metadynamics/.
 */
#ifndef MTD_REF_H
#define MTD_REF_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* HOOMD BoxDim restated: lattice vectors
 * a1=(Lx,0,0), a2=(xy*Ly,Ly,0), a3=(xz*Lz,yz*Lz,Lz); lo = lower corner. */
typedef struct ref_box
    {
    double L[3];
    double lo[3];
    double xy, xz, yz;
    } ref_box;

/* ---------------- IndexGrid (IndexGrid.cc:20-58) ---------------- */
unsigned int ref_index_get(unsigned int dim, const unsigned int *lengths, const unsigned int *coords);
void ref_index_coords(unsigned int dim, const unsigned int *lengths, unsigned int idx, unsigned int *coords);

/* ---------------- Lamellar (LamellarOrderParameter.cc) ---------------- */
/* postype: double[4*N] = (x,y,z,type as double holding the integer type id).
 * lattice: int[3*n_wave].  mode: double[ntypes].  modes_out: double[2*n_wave] = (Re,Im). */
void ref_lamellar_fourier_modes(unsigned int n_wave, const int *lattice, unsigned int N,
                                const double *postype, const double *mode, const ref_box *global_box,
                                double *modes_out);                         /* .cc:143-179 */
double ref_lamellar_cv(unsigned int n_wave, const double *modes, unsigned int N_global); /* .cc:58-68 */
void ref_lamellar_forces(unsigned int n_wave, const int *lattice, unsigned int N,
                         const double *postype, const double *mode, const ref_box *global_box,
                         unsigned int N_global, double bias, double *force_out /*4*N*/); /* .cc:77-140 */

/* ---------------- Bias-grid engine (IntegratorMetaDynamics.cc) ---------------- */
enum { REF_MODE_STANDARD = 0, REF_MODE_WELL_TEMPERED = 1 };

typedef struct ref_metad ref_metad;

ref_metad *ref_metad_create(unsigned int n_cv, const double *sigma, const double *cv_min,
                            const double *cv_max, const unsigned int *num_points,
                            double W, double T_shift, double T, unsigned int stride,
                            int mode, int add_bias);                        /* ctor :23-72, prepRun :121-217, setupGrid :590-661 */
void ref_metad_destroy(ref_metad *m);
unsigned int ref_metad_num_elements(const ref_metad *m);
void ref_metad_set_stride(ref_metad *m, unsigned int stride);
void ref_metad_set_add_bias(ref_metad *m, int add_bias);
void ref_metad_set_mode(ref_metad *m, int mode);
void ref_metad_set_sigma_inv(ref_metad *m, const double *sigma_inv /*n_cv^2*/);
void ref_metad_reset_histogram(ref_metad *m);                               /* :1195-1203 */

/* updateBiasPotential (:314-588), grid branch.  bias_out[n_cv] = dV/ds_i. */
void ref_metad_update_bias(ref_metad *m, unsigned int timestep, const double *current_val, double *bias_out);

/* the same, split at the multiple-walker all-reduce (:393-409): phase A = histogram, sigma grid,
 * scal, updateGrid (fills the four delta arrays); phase B = reweight + accumulate + evaluate. */
int  ref_metad_update_phase_a(ref_metad *m, unsigned int timestep, const double *current_val);
void ref_metad_update_phase_b(ref_metad *m, int deposited, const double *current_val, double *bias_out);

double ref_metad_interpolate(ref_metad *m, const double *val, int reweight);  /* :663-736 */
double ref_metad_derivative(ref_metad *m, unsigned int cv, const double *val);/* :738-776 */
double ref_metad_sigma_determinant(const ref_metad *m);                       /* :1296-1313 */
double ref_metad_curr_bias(const ref_metad *m);      /* log quantity "bias"   (:448) */
double ref_metad_curr_weight(const ref_metad *m);    /* log quantity "weight" (:451) */
unsigned int ref_metad_num_gaussians(const ref_metad *m);
unsigned int ref_metad_num_oob_warnings(const ref_metad *m);

/* raw array access for comparisons: which = 0 grid, 1 grid_delta, 2 reweighted, 3 weight,
 * 4 sigma_grid, 5 sigma_grid_delta (double*); 6 hist, 7 hist_delta, 8 hist_gauss, 9 hist_gauss_delta (unsigned*) */
void *ref_metad_array(ref_metad *m, int which);

/* writeGrid (:831-926) / readGrid (:928-1000).  names: n_cv C strings.  Returns 0 on success. */
int ref_metad_write_grid(ref_metad *m, const char *filename, unsigned int timestep, const char *const *names);
int ref_metad_read_grid(ref_metad *m, const char *filename);

/* standalone pieces, for unit comparisons */
void ref_update_grid(unsigned int dim, const unsigned int *lengths, const double *cv_min, const double *cv_max,
                     const double *sigma_inv, const double *current_val, double scal, double W,
                     double *grid_delta);                                     /* updateGrid :1002-1047 */

/* ---------------- Umbrella (CollectiveVariable.cc:22-106) ---------------- */
enum { REF_NO_UMBRELLA = 0, REF_LINEAR, REF_HARMONIC, REF_WALL, REF_GAUSSIAN };
double ref_umbrella_bias(int umbrella, double val, double bias_in, double cv0, double kappa,
                         double width_flat, double scale);                   /* :22-60 */
double ref_umbrella_energy(int umbrella, double val, double cv0, double kappa,
                           double width_flat, double scale);                 /* :68-106 */

/* ---------------- Box CVs needed by test_2d.py (AspectRatio.cc:24-57, Density.cc:20-27) -------- */
double ref_aspect_ratio(const ref_box *box, unsigned int dir1, unsigned int dir2);
double ref_density(const ref_box *box, unsigned int N_group);

/* ---------------- WellTemperedEnsemble (WellTemperedEnsemble.cc:30-68, 135-188) ---------------- */
double ref_wte_potential_energy(unsigned int N, const double *net_force /*4N*/, double external_energy);
void ref_wte_scale(unsigned int N, double *net_force /*4N*/, double *net_torque /*4N*/,
                   double *net_virial /*6*pitch*/, unsigned int pitch, double *external_virial /*6*/, double bias);

/* ---------------- CollectiveWrapper (CollectiveWrapper.cc:31-72, 136-179, CPU path) ------------------------ */
double ref_wrapper_energy(unsigned int N, const double *force /*4N*/, double external_energy);
/* force.xyz, torque.xyzw and the six virial rows of the wrapped compute *= bias (:153-171) */
void ref_wrapper_scale(unsigned int N, double *force /*4N*/, double *torque /*4N*/, double *virial /*6*pitch*/,
                       unsigned int pitch, double bias);

/* ---------------- adaptive Gaussians (IntegratorMetaDynamics.cc:1205-1294, single rank) --------------------- */
/* forces: n_cv arrays of 4N doubles (the derivative arrays after computeDerivatives); can_derive[c] != 0 when CV c
 * provides derivatives; sigma[c] the registered widths.  Writes sigmasq (n_cv^2) and sigma_inv (n_cv^2). */
void ref_compute_sigma(unsigned int n_cv, unsigned int N, const double *const *forces, const int *can_derive,
                       const double *sigma, double sigma_g, double *sigmasq, double *sigma_inv);

/* ---------------------------------------------------------------- OrderParameterMesh (mtd_ref_mesh.c) */
typedef struct ref_mesh ref_mesh;
ref_mesh *ref_mesh_create(unsigned int nx, unsigned int ny, unsigned int nz, unsigned int n_types, const double *mode);
void ref_mesh_destroy(ref_mesh *m);
/* 1 (default): interpolation function exactly as the reference computes it (unsigned division, Q6); 0: as intended */
void ref_mesh_set_bug_compat(ref_mesh *m, int on);
/* getCurrentValue (OrderParameterMesh.cc:925-968): assignParticles -> FFT -> updateMeshes -> iFFT -> computeCV */
double ref_mesh_cv(ref_mesh *m, unsigned int N, const double *postype, const ref_box *box, unsigned int N_global);
/* convolution kernel table (setTable :148-189; K is stored in inf_f and never applied, Q7), q_max log quantities (:1108-1179),
 * virial (:970-1050, nonzero only with a table in use) — all on the Fourier mesh of the last ref_mesh_cv */
int ref_mesh_set_table(ref_mesh *m, const double *K, const double *d_K, unsigned int n, double kmin, double kmax);
void ref_mesh_set_use_table(ref_mesh *m, int on);
void ref_mesh_qmax(const ref_mesh *m, unsigned int N_global, double *out /*qx,qy,qz,sq_max*/);
void ref_mesh_virial(const ref_mesh *m, unsigned int N_global, double bias, double *virial /*6*/);
/* halves of ref_mesh_cv for particle-sharded checks: spread one shard; (sum mesh + mode_sq over shards); spectral part */
void ref_mesh_assign(ref_mesh *m, unsigned int N, const double *postype, const ref_box *box);
void ref_mesh_set_mode_sq(ref_mesh *m, double mode_sq);
double ref_mesh_spectral(ref_mesh *m, unsigned int N_global);
/* interpolateForces (:749-864); call after ref_mesh_cv on the same snapshot */
void ref_mesh_forces(ref_mesh *m, unsigned int N, const double *postype, const ref_box *box, unsigned int N_global,
                     double bias, double *force_out /*4N*/);
double ref_mesh_mode_sq(const ref_mesh *m);
/* which: 0 mesh, 1 fourier_mesh (normalised), 2 fourier_mesh_G, 3 inv_fourier_mesh (complex double[M], (re,im));
 *        4 interpolation_f, 5 inf_f (double[M]); 6 k (double[3M]) */
void *ref_mesh_array(ref_mesh *m, int which);

/* ---------------------------------------------------------------- Steinhardt Q_l (mtd_ref_steinhardt.c) */
/* fsph::evaluate_SPH (spherical_harmonics.hpp:229-246), argument order of the header: (phi = polar, theta = azimuth).
 * out: (re,im) pairs, per point (lmax+1)^2 values (full_m) or (lmax+1)(lmax+2)/2; per l: m = 0..l then -1..-l. */
void ref_sph_evaluate(double *out, unsigned int lmax, const double *phi, const double *theta, unsigned int N, int full_m);
/* SteinhardtQl::computeCV (SteinhardtQl.cc:62-201): neighbour list in HOOMD layout (head_list[N], n_neigh[N], nlist[]). */
double ref_ql_compute_cv(unsigned int N, const double *postype, const ref_box *box, const unsigned int *head_list,
                         const unsigned int *n_neigh, const unsigned int *nlist, int half_nlist, double rcut, double ron,
                         unsigned int lmax, unsigned int type, const double *Ql_ref, unsigned int N_global,
                         double *Qlm_out /*2 (lmax+1)^2*/, double *Ql_out /*lmax+1*/);
/* tail of computeCV (:181-194) from a summed Q_lm table (particle-sharded checks) */
double ref_ql_from_qlm(unsigned int lmax, const double *Qlm_in, const double *Ql_ref, unsigned int N_global, double *Ql_out);
/* SteinhardtQl::computeBiasForces (:203-339) with the Q_lm computeCV left behind (Q20) */
void ref_ql_compute_forces(unsigned int N, const double *postype, const ref_box *box, const unsigned int *head_list,
                           const unsigned int *n_neigh, const unsigned int *nlist, int half_nlist, double rcut, double ron,
                           unsigned int lmax, unsigned int type, const double *Ql_ref, unsigned int N_global,
                           const double *Qlm_in, double bias, double *force_out /*4N*/);

#ifdef __cplusplus
}
#endif
#endif
