// Developer probe: does v_cos_f32 / v_sin_f32 (angle in turns, domain [-256, 256]) need the v_fract_f32 the lamellar kernels put in
// front of it?  Compares the hardware result with and without the explicit fract over phases in [-18.5, 18.5] turns.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float *t, unsigned long long *cnt, double *out, int n)
    {
    unsigned long long dc = 0, ds = 0;
    double mc0 = 0, mc1 = 0, ms0 = 0, ms1 = 0, e0 = 0, e1 = 0, f0 = 0, f1 = 0, f2 = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        {
        const float c0 = __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(t[i])), c1 = __builtin_amdgcn_cosf(t[i]);
        const float s0 = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(t[i])), s1 = __builtin_amdgcn_sinf(t[i]);
        dc += __float_as_uint(c0) != __float_as_uint(c1);
        ds += __float_as_uint(s0) != __float_as_uint(s1);
        const double x = 2.0 * M_PI * (double)t[i];
        mc0 = fmax(mc0, fabs((double)c0 - cos(x))); mc1 = fmax(mc1, fabs((double)c1 - cos(x)));
        ms0 = fmax(ms0, fabs((double)s0 - sin(x))); ms1 = fmax(ms1, fabs((double)s1 - sin(x)));
        e0 += (double)c0 - cos(x); e1 += (double)c1 - cos(x);
        // the folded second harmonic 2 c^2 - k from the cosine WITHOUT fract, k = 1, 1 - 1 ulp, 1 - 2 ulp
        f0 += (double)(2.0f * c1 * c1 - 1.0f) - cos(2.0 * x); f1 += (double)(2.0f * c1 * c1 - 0.99999994f) - cos(2.0 * x); f2 += (double)(2.0f * c1 * c1 - 0.99999988f) - cos(2.0 * x);
        }
    atomicAdd(cnt + 0, dc); atomicAdd(cnt + 1, ds);
    atomicMax((unsigned long long *)(out + 0), (unsigned long long)__double_as_longlong(mc0));
    atomicMax((unsigned long long *)(out + 1), (unsigned long long)__double_as_longlong(mc1));
    atomicMax((unsigned long long *)(out + 2), (unsigned long long)__double_as_longlong(ms0));
    atomicMax((unsigned long long *)(out + 3), (unsigned long long)__double_as_longlong(ms1));
    atomicAdd(out + 4, e0); atomicAdd(out + 5, e1); atomicAdd(out + 6, f0); atomicAdd(out + 7, f1); atomicAdd(out + 8, f2);
    }
int main()
    {
    const int n = 1 << 24;
    std::vector<float> h(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (float)((s >> 11) * (1.0 / 9007199254740992.0) * 37.0 - 18.5); }
    float *d; double *o; unsigned long long *c; double r[9] = {0}; unsigned long long rc[2] = {0};
    hipMalloc(&d, n * 4); hipMalloc(&o, 72); hipMalloc(&c, 16); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); hipMemset(o, 0, 72); hipMemset(c, 0, 16);
    k<<<1024, 256>>>(d, c, o, n); hipMemcpy(r, o, 72, hipMemcpyDeviceToHost); hipMemcpy(rc, c, 16, hipMemcpyDeviceToHost);
    printf("phases in [-18.5, 18.5] turns, %d samples: results that differ with / without fract: cos %llu  sin %llu\n", n, rc[0], rc[1]);
    printf("max |error| cos: with fract %.3e  without %.3e | sin: with %.3e  without %.3e | mean error cos: with %.3e without %.3e\n",
           r[0], r[1], r[2], r[3], r[4] / n, r[5] / n);
    printf("mean error of the folded harmonic 2 c^2 - k (c without fract): k = 1: %.3e  k = 1 - 1 ulp: %.3e  k = 1 - 2 ulp: %.3e\n", r[6] / n, r[7] / n, r[8] / n);
    return 0;
    }
