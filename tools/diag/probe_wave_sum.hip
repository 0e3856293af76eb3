// Developer tool (GPU box): wave_sum (DPP + v_readlane, mtd_device.hpp) against the xor butterfly it replaces, bit for bit,
// on random floats and doubles (wide range of magnitudes and signs), and the quad broadcasts chain_wave uses.
// build: hipcc --offload-arch=gfx950 -O3 -I include -I metadynamics-plugin_amd/csrc tools/diag/probe_wave_sum.hip -o tools/bin/probe_wave_sum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "mtd_device.hpp"

using namespace mtd;

__global__ void k_probe(const double *d, const float *f, unsigned long long *bad, int n_waves)
    {
    const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    if (w >= n_waves) return;
    const int lane = threadIdx.x & 63;
    const double x = d[w * 64 + lane];
    const float y = f[w * 64 + lane];
    const double a = wave_sum(x), b = wave_sum_butterfly(x);
    const float p = wave_sum(y), q = wave_sum_butterfly(y);
    if (__double_as_longlong(a) != __double_as_longlong(b)) atomicAdd(bad + 0, 1ull);
    if (__float_as_int(p) != __float_as_int(q)) atomicAdd(bad + 1, 1ull);
    // quad broadcasts: lane l of a quad reads lane b of the same quad
    const double q0 = dpp_move<0x00>(x), q1 = dpp_move<0x55>(x), q2 = dpp_move<0xAA>(x), q3 = dpp_move<0xFF>(x);
    const int base = lane & ~3;
    if (q0 != __shfl(x, base, 64) || q1 != __shfl(x, base + 1, 64) || q2 != __shfl(x, base + 2, 64) || q3 != __shfl(x, base + 3, 64)) atomicAdd(bad + 2, 1ull);
    const double h0 = dpp_move<0xA0>(x), h1 = dpp_move<0xF5>(x);
    const int pb = lane & ~1;
    if (h0 != __shfl(x, pb, 64) || h1 != __shfl(x, pb + 1, 64)) atomicAdd(bad + 3, 1ull);
    if (wave_read(x, 17) != __shfl(x, 17, 64)) atomicAdd(bad + 4, 1ull);
    // the DPP stages one by one against the lanes they are meant to read
    if (dpp_move<MTD_DPP_QUAD_XOR1>(y) != __shfl(y, lane ^ 1, 64)) atomicAdd(bad + 5, 1ull);
    if (dpp_move<MTD_DPP_QUAD_XOR2>(y) != __shfl(y, lane ^ 2, 64)) atomicAdd(bad + 6, 1ull);
    if (dpp_move<MTD_DPP_ROW_HALF_MIRROR>(y) != __shfl(y, (lane & ~7) | (7 - (lane & 7)), 64)) atomicAdd(bad + 7, 1ull);
    if (dpp_move<MTD_DPP_ROW_MIRROR>(y) != __shfl(y, (lane & ~15) | (15 - (lane & 15)), 64)) atomicAdd(bad + 8, 1ull);
    // the float sum stage by stage
    float s1 = y + dpp_move<MTD_DPP_QUAD_XOR1>(y), t1 = y + __shfl_xor(y, 1, 64);
    if (__float_as_int(s1) != __float_as_int(t1)) atomicAdd(bad + 9, 1ull);
    float s2 = s1 + dpp_move<MTD_DPP_QUAD_XOR2>(s1), t2 = t1 + __shfl_xor(t1, 2, 64);
    if (__float_as_int(s2) != __float_as_int(t2)) atomicAdd(bad + 10, 1ull);
    float s3 = s2 + dpp_move<MTD_DPP_ROW_HALF_MIRROR>(s2), t3 = t2 + __shfl_xor(t2, 4, 64);
    if (__float_as_int(s3) != __float_as_int(t3)) atomicAdd(bad + 11, 1ull);
    float s4 = s3 + dpp_move<MTD_DPP_ROW_MIRROR>(s3), t4 = t3 + __shfl_xor(t3, 8, 64);
    if (__float_as_int(s4) != __float_as_int(t4)) atomicAdd(bad + 12, 1ull);
    float t5 = t4 + __shfl_xor(t4, 16, 64);
    float t6 = t5 + __shfl_xor(t5, 32, 64);
    const float r0 = wave_read(s4, 0), r1 = wave_read(s4, 16), r2 = wave_read(s4, 32), r3 = wave_read(s4, 48);
    if (__float_as_int((r0 + r1) + (r2 + r3)) != __float_as_int(t6)) atomicAdd(bad + 13, 1ull);
    if (w == 0 && lane < 4) printf("lane %d: y=%g s4=%g t4=%g sum=%g butterfly=%g t6=%g\n", lane, y, s4, t4, p, q, t6);
    }

int main()
    {
    const int n_waves = 1 << 16;
    std::vector<double> d(n_waves * 64);
    std::vector<float> f(n_waves * 64);
    srand48(7);
    for (size_t i = 0; i < d.size(); ++i)
        {
        const double m = ldexp(drand48() - 0.5, (int)(drand48() * 60) - 30);
        d[i] = m;
        f[i] = (float)ldexp(drand48() - 0.5, (int)(drand48() * 40) - 20);
        }
    double *dd; float *df; unsigned long long *db, bad[16] = {0};
    hipMalloc(&dd, d.size() * 8); hipMalloc(&df, f.size() * 4); hipMalloc(&db, 16 * 8);
    hipMemcpy(dd, d.data(), d.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(df, f.data(), f.size() * 4, hipMemcpyHostToDevice);
    hipMemset(db, 0, 16 * 8);
    k_probe<<<n_waves / 4, 256>>>(dd, df, db, n_waves);
    hipMemcpy(bad, db, 16 * 8, hipMemcpyDeviceToHost);
    unsigned long long tot = 0;
    const char *names[14] = {"double sum", "float sum", "quad bcast", "pair bcast", "readlane", "dpp xor1", "dpp xor2", "dpp half mirror",
                             "dpp mirror", "stage1", "stage2", "stage3", "stage4", "rows"};
    for (int i = 0; i < 14; ++i) { printf("%-16s mismatches %llu\n", names[i], bad[i]); tot += bad[i]; }
    printf("waves %d  mismatches %llu  (%s)\n", n_waves, tot, hipGetErrorString(hipGetLastError()));
    return tot != 0;
    }
