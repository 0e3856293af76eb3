// Developer micro-benchmark (GPU box): what does the FIRST pass of a wave through straight-line code cost on gfx950?
// A kernel of KB kilobytes of s_nop (4 bytes, one cycle each) run `iters` times in a loop by 256 blocks x 8 waves (one block per CU,
// like the mesh transforms): time(iters = 2) - time(iters = 1) is a warm pass, time(iters = 1) - launch floor a cold one.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/icache tools/diag/icache.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template<int KB>
__global__ __launch_bounds__(512) void k_code(const int iters, unsigned long long *out)
    {
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it)
        {
        asm volatile(".rept %0\n s_nop 0\n .endr" :: "n"(KB * 256));
        }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    }

template<int KB>
void run(unsigned long long *d_out)
    {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int iters = 0; iters <= 3; ++iters)
        {
        std::vector<float> ms;
        for (int rep = 0; rep < 30; ++rep)
            {
            hipEventRecord(a, 0);
            k_code<KB><<<256, 512, 64 * 1024, 0>>>(iters, d_out);
            hipEventRecord(b, 0);
            hipEventSynchronize(b);
            float t;
            hipEventElapsedTime(&t, a, b);
            ms.push_back(t);
            // something else in between, as in a step of several kernels
            k_code<1><<<256, 512, 0, 0>>>(1, d_out);
            hipDeviceSynchronize();
            }
        std::sort(ms.begin(), ms.end());
        printf("code %3d KB  passes %d  median %7.2f us  min %7.2f us\n", KB, iters, 1e3 * ms[ms.size() / 2], 1e3 * ms[0]);
        }
    }

int main()
    {
    unsigned long long *d_out;
    hipMalloc(&d_out, 256 * sizeof(unsigned long long));
    run<4>(d_out);
    run<16>(d_out);
    run<32>(d_out);
    run<64>(d_out);
    return 0;
    }
