// A C++ caller of the drop-in boundary, written the way INTEGRATION.md tells a maintainer of the reference to call it from
// LamellarOrderParameterGPU::computeCV / computeBiasForces (LamellarOrderParameterGPU.cc:34-132) and from
// IntegratorMetaDynamics::updateBiasPotential (IntegratorMetaDynamics.cc:314-588): raw device pointers borrowed from the
// caller's own allocations (hipMalloc here, GPUArray/ArrayHandle there), host arrays for lattice vectors and mode
// coefficients, an explicit stream, int return codes.  Checked against the CPU oracle (oracle/mtd_ref.h: test infrastructure).
// Built and run by tests/test_gpu_cpp_caller.py with g++ (no device code: HIP runtime API + libmtd_hip.so + libmtd_ref.so).
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mtd_abi.h"
#include "mtd_ref.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define MTD_OK(x) do { int r_ = (x); if (r_ != 0) { std::fprintf(stderr, "%s returned %d\n", #x, r_); return 3; } } while (0)

static double rnd(unsigned long long &s)
    {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(s >> 11) / 9007199254740992.0;
    }

int main()
    {
    const unsigned int N = 20000, n_wave = 3, n_types = 2;
    const double L = 25.0;
    const int lattice[3 * n_wave] = {0, 0, 2, 1, 1, 0, 2, 0, 1};
    const double mode[n_types] = {1.0, -0.7};

    // a Scalar4 postype array in single precision, type id bit-cast into w like ParticleData's (x, y, z, __int_as_scalar(type))
    std::vector<float> h_pos(4 * N);
    std::vector<double> ref_pos(4 * N);
    unsigned long long seed = 20241004;
    for (unsigned int i = 0; i < N; ++i)
        {
        const int type = (int)(i & 1);
        float xyz[3];
        for (int d = 0; d < 3; ++d) xyz[d] = (float)((rnd(seed) - 0.5) * L);
        xyz[2] += 0.4f * (type ? -1.f : 1.f) * std::sin(2.0f * 3.14159265f * 2.0f * xyz[2] / (float)L);   // a density wave: the CV is not noise
        for (int d = 0; d < 3; ++d)
            {
            h_pos[4 * i + d] = xyz[d];
            ref_pos[4 * i + d] = (double)xyz[d];
            }
        union { int i; float f; } w;
        w.i = type;
        h_pos[4 * i + 3] = w.f;
        ref_pos[4 * i + 3] = (double)type;
        }
    mtd_box box;
    ref_box rbox;
    for (int d = 0; d < 3; ++d)
        {
        box.L[d] = rbox.L[d] = L;
        box.lo[d] = rbox.lo[d] = -L / 2;
        box.periodic[d] = 1;
        }
    box.xy = box.xz = box.yz = rbox.xy = rbox.xz = rbox.yz = 0.0;

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    float *d_pos, *d_force;
    double *d_modes, *d_scratch;
    HIP_OK(hipMalloc((void **)&d_pos, sizeof(float) * 4 * N));
    HIP_OK(hipMalloc((void **)&d_force, sizeof(float) * 4 * N));
    HIP_OK(hipMalloc((void **)&d_modes, sizeof(double) * 2 * n_wave));
    HIP_OK(hipMalloc((void **)&d_scratch, sizeof(double) * mtd_lamellar_scratch_doubles(N)));
    HIP_OK(hipMemcpyAsync(d_pos, h_pos.data(), sizeof(float) * 4 * N, hipMemcpyHostToDevice, stream));

    // computeCV: gpu_calculate_fourier_modes -> modes on the host -> s = sum_k Re F_k / N_global (.cc:58-68)
    MTD_OK(mtd_calculate_fourier_modes(n_wave, lattice, N, d_pos, MTD_F32, mode, n_types, d_modes, d_scratch, &box, stream));
    double modes[2 * n_wave], modes_ref[2 * n_wave];
    HIP_OK(hipMemcpyAsync(modes, d_modes, sizeof(modes), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    double cv = 0.0;
    for (unsigned int k = 0; k < n_wave; ++k) cv += modes[2 * k];
    cv /= (double)N;
    ref_lamellar_fourier_modes(n_wave, lattice, N, ref_pos.data(), mode, &rbox, modes_ref);
    const double cv_ref = ref_lamellar_cv(n_wave, modes_ref, N);
    int bad = 0;
    double worst_mode = 0.0;
    for (unsigned int j = 0; j < 2 * n_wave; ++j) worst_mode = std::fmax(worst_mode, std::fabs(modes[j] - modes_ref[j]));
    std::printf("cv %.12g (oracle %.12g), worst Fourier-mode deviation %.3g of %u particles\n", cv, cv_ref, worst_mode, N);
    if (std::fabs(cv - cv_ref) > 1e-6 * std::fabs(cv_ref)) ++bad;                   // 1e-6 on the modulated snapshot

    // updateBiasPotential with the grid on the device: the CV arrives as a host scalar (the path of box-shape CVs)
    const double sigma[1] = {0.05}, cv_min[1] = {-1.0}, cv_max[1] = {1.0};
    const unsigned int num_points[1] = {128};
    mtd_metad *engine = nullptr;
    MTD_OK(mtd_metad_create(&engine, 1, sigma, cv_min, cv_max, num_points, 1.0, 7.0, 1.0, 1, MTD_MODE_WELL_TEMPERED, 1));
    ref_metad *oracle = ref_metad_create(1, sigma, cv_min, cv_max, num_points, 1.0, 7.0, 1.0, 1, REF_MODE_WELL_TEMPERED, 1);
    double bias = 0.0, bias_ref = 0.0, V = 0.0, w = 0.0;
    for (unsigned int t = 0; t < 5; ++t)
        {
        const double s = cv + 0.013 * t;                                          // the CV moves a little every step
        MTD_OK(mtd_metad_set_cv_value(engine, 0, s));
        MTD_OK(mtd_metad_update_bias(engine, t, stream));
        ref_metad_update_bias(oracle, t, &s, &bias_ref);
        }
    unsigned int n_gauss = 0;
    MTD_OK(mtd_metad_get_state(engine, nullptr, &bias, &V, &w, &n_gauss, nullptr, stream));
    std::printf("after 5 deposits: dV/ds %.12g (oracle %.12g), V %.12g (%.12g), w %.12g (%.12g), %u hills\n", bias, bias_ref, V,
                ref_metad_curr_bias(oracle), w, ref_metad_curr_weight(oracle), n_gauss);
    if (std::fabs(bias - bias_ref) > 1e-9 * std::fabs(bias_ref) || std::fabs(V - ref_metad_curr_bias(oracle)) > 1e-9 * std::fabs(V) ||
        std::fabs(w - ref_metad_curr_weight(oracle)) > 1e-9 * std::fabs(w) || n_gauss != 5)
        ++bad;

    // computeBiasForces: gpu_compute_sq_forces with the bias factor as a host scalar (setBiasFactor, .cc:578-584)
    MTD_OK(mtd_compute_sq_forces(N, d_pos, d_force, MTD_F32, n_wave, lattice, mode, n_types, N, bias, &box, stream));
    std::vector<float> h_force(4 * N);
    HIP_OK(hipMemcpyAsync(h_force.data(), d_force, sizeof(float) * 4 * N, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<double> f_ref(4 * N);
    ref_lamellar_forces(n_wave, lattice, N, ref_pos.data(), mode, &rbox, N, bias_ref, f_ref.data());
    double fmax = 0.0, dmax = 0.0;
    for (unsigned int i = 0; i < N; ++i)
        for (int d = 0; d < 3; ++d)
            {
            fmax = std::fmax(fmax, std::fabs(f_ref[4 * i + d]));
            dmax = std::fmax(dmax, std::fabs((double)h_force[4 * i + d] - f_ref[4 * i + d]));
            }
    for (unsigned int i = 0; i < N; ++i)
        if (h_force[4 * i + 3] != 0.0f) ++bad;                                     // force.w = 0 (.cc:134)
    std::printf("forces: worst deviation %.3g of max |F| %.3g\n", dmax, fmax);
    if (!(dmax <= 1e-5 * fmax)) ++bad;                                             // 1e-5 of max |F|

    // error behaviour: invalid arguments come back as codes, nothing throws across the boundary
    if (mtd_calculate_fourier_modes(0, lattice, N, d_pos, MTD_F32, mode, n_types, d_modes, d_scratch, &box, stream) != MTD_ERR_INVALID_ARGUMENT) ++bad;
    if (mtd_metad_create(&engine, 1, sigma, cv_max, cv_min, num_points, 1.0, 7.0, 1.0, 1, MTD_MODE_WELL_TEMPERED, 1) != MTD_ERR_INVALID_ARGUMENT) ++bad;

    ref_metad_destroy(oracle);
    MTD_OK(mtd_metad_destroy(engine));
    (void)hipFree(d_pos); (void)hipFree(d_force); (void)hipFree(d_modes); (void)hipFree(d_scratch);
    (void)hipStreamDestroy(stream);
    std::printf(bad ? "FAIL (%d)\n" : "PASS\n", bad);
    return bad ? 1 : 0;
    }
