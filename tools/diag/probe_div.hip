// Developer probe (GPU box): exact_div (csrc/exact_div.hpp: multiply + four FMAs with a host reciprocal) against the hardware
// double division, bit for bit, over random numerators for divisors of the kinds mesh.hip::locate uses (box lengths, mesh
// dimensions) and adversarial ones.  build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=on -Imetadynamics-plugin_amd/csrc tools/diag/probe_div.hip -o tools/bin/probe_div
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "exact_div.hpp"
__device__ unsigned long long splitmix(unsigned long long &s)
    {
    unsigned long long z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
    }
__global__ void k_probe(const ExactDivisor d, const unsigned long long seed, const unsigned int per_thread, const int mode, unsigned long long *bad, double *first)
    {
    unsigned long long s = seed + 0x1234567ull * (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x);
    for (unsigned int i = 0; i < per_thread; ++i)
        {
        const unsigned long long r = splitmix(s);
        double a;
        if (mode == 0)                       // uniform in (-4 b, 4 b): coordinates relative to a box
            a = ((double)(long long)r * (1.0 / 9223372036854775808.0)) * 4.0 * d.b;
        else if (mode == 1)                  // half-integers (cell centres): (i + 0.5), i < 1024
            a = (double)(r & 1023) + 0.5;
        else                                 // random bit patterns with moderate exponents
            a = __longlong_as_double((long long)((r & 0x800fffffffffffffull) | ((unsigned long long)(1023 - 40 + (r >> 52) % 80) << 52)));
        const double q = exact_div(a, d), t = a / d.b;
        if (__double_as_longlong(q) != __double_as_longlong(t))
            if (atomicAdd(bad, 1ull) == 0) { first[0] = a; first[1] = q; first[2] = t; }
        }
    }
int main()
    {
    unsigned long long *d_bad; double *d_first;
    hipMalloc(&d_bad, 8); hipMalloc(&d_first, 24);
    std::vector<double> divisors = {100.0, 12.3082284, 9.7, 6.6019954, 7.5, 3.0000000000000004, 14.999999999999998, 1.0 / 3.0, 128.0, 17.0, 33.0, 9.0, 1023.0, 5.0, 6.0, 20.0,
                                    1.9999999999999998 /* all-ones significand: must take the slow form */, 0.1, 1e-3, 12345.678, 3.141592653589793};
    srand(7);
    for (int k = 0; k < 40; ++k) divisors.push_back(3.0 + 12.0 * (rand() / (double)RAND_MAX));
    unsigned long long total = 0, total_bad = 0;
    for (double b : divisors)
        for (int mode = 0; mode < 3; ++mode)
            {
            const ExactDivisor d = make_exact_divisor(b);
            hipMemset(d_bad, 0, 8);
            const unsigned int per = 1u << 13;
            k_probe<<<1024, 256>>>(d, 0xabcdefull + (unsigned long long)(b * 1e6) + mode, per, mode, d_bad, d_first);
            unsigned long long bad = 0; double first[3];
            hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost); hipMemcpy(first, d_first, 24, hipMemcpyDeviceToHost);
            total += 1024ull * 256 * per; total_bad += bad;
            if (bad) printf("divisor %.17g (fast %d) mode %d: %llu differences, first a = %.17g: %.17g vs %.17g\n", b, d.fast, mode, bad, first[0], first[1], first[2]);
            }
    printf("probe_div: %llu quotients over %zu divisors: %llu differ from the hardware division\n", total, divisors.size(), total_bad);
    return total_bad ? 1 : 0;
    }
