// Developer tool (not product): calibrates fixed per-kernel costs on MI355X so the step structure can be
// chosen from measurements: back-to-back launch cost of trivial kernels, cost of dependent global
// round trips, block reductions, double exp chains, kernarg->LDS staging.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/latency_probe tools/diag/latency_probe.hip && /tmp/latency_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                      \
    do                                                                                \
        {                                                                             \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess)                                                          \
            {                                                                         \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            return 1;                                                                 \
            }                                                                         \
        } while (0)

__global__ void k_empty() {}

__global__ void k_touch(double *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0; }

// chain of `hops` dependent loads (pointer chase through idx[])
__global__ void k_chase(const unsigned int *idx, unsigned int *out, int hops)
    {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        {
        unsigned int i = 0;
        for (int h = 0; h < hops; ++h) i = idx[i];
        out[0] = i;
        }
    }

__device__ __forceinline__ double wave_sum(double v)
    {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
    }

// every block: load 2 values per thread from partials, block reduce (2 syncs), thread 0 exp + store
__global__ void k_reduce_exp(const double *partials, double *out, int n_exp)
    {
    __shared__ double s[16];
    double v = partials[threadIdx.x] + partials[threadIdx.x + 256];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = s[0] + s[1] + s[2] + s[3];
    for (int i = 0; i < n_exp; ++i) r = exp(-r * 1e-3);
    if (threadIdx.x == 0) out[blockIdx.x] = r;
    }

struct Big { float4 h[64]; float4 q[64]; float c[128]; };

__global__ void k_kernarg_lds(const Big a, float *out)
    {
    __shared__ float4 s_h[64];
    for (unsigned int k = threadIdx.x; k < 64; k += blockDim.x) s_h[k] = a.h[k];
    __syncthreads();
    float acc = 0.f;
    for (int k = 0; k < 16; ++k) acc += s_h[k].x * threadIdx.x;
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
    }

__global__ void k_kernarg_sgpr(const Big a, float *out)
    {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += a.h[k].x * threadIdx.x;
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
    }

// streaming read of n float4 + trivial compute, grid-stride
__global__ void k_stream_read(const float4 *p, unsigned int n, float *out)
    {
    float acc = 0.f;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        {
        float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
        }
    if (acc == 12345.678f) out[0] = acc;
    }

__global__ void k_stream_rw(const float4 *p, float4 *o1, float4 *o2, unsigned int n)
    {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        {
        float4 v = p[i];
        o1[i] = v;
        o2[i] = make_float4(v.y, v.x, v.w, v.z);
        }
    }

template<typename F> float time_loop(F f, int iters, hipStream_t s)
    {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 50; ++i) f();
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a);
    hipEventDestroy(b);
    return 1e3f * ms / iters;
    }

int main()
    {
    hipStream_t s = 0;
    const int it = 2000;
    double *d;
    CHECK(hipMalloc(&d, 1 << 20));
    CHECK(hipMemset(d, 0, 1 << 20));
    unsigned int *idx, *uo;
    CHECK(hipMalloc(&idx, 1 << 22));
    CHECK(hipMalloc(&uo, 64));
    std::vector<unsigned int> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned int)((i * 7919u + 104729u) % h.size());
    CHECK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    float *fo;
    CHECK(hipMalloc(&fo, 1 << 16));
    Big big = {};

    printf("empty<<<1,64>>>            %.2f us/launch\n", time_loop([&] { k_empty<<<1, 64, 0, s>>>(); }, it, s));
    printf("empty<<<256,256>>>         %.2f us/launch\n", time_loop([&] { k_empty<<<256, 256, 0, s>>>(); }, it, s));
    printf("empty<<<1024,256>>>        %.2f us/launch\n", time_loop([&] { k_empty<<<1024, 256, 0, s>>>(); }, it, s));
    printf("empty<<<2048,256>>>        %.2f us/launch\n", time_loop([&] { k_empty<<<2048, 256, 0, s>>>(); }, it, s));
    printf("touch (1 RMW)<<<1,64>>>    %.2f us/launch\n", time_loop([&] { k_touch<<<1, 64, 0, s>>>(d); }, it, s));
    for (int hops : {1, 2, 4, 8, 16})
        printf("chase %2d hops<<<1,64>>>    %.2f us/launch\n", hops, time_loop([&] { k_chase<<<1, 64, 0, s>>>(idx, uo, hops); }, it, s));
    for (int ne : {0, 1, 4, 16})
        printf("reduce+%2d exp<<<1024,256>>> %.2f us/launch\n", ne, time_loop([&] { k_reduce_exp<<<1024, 256, 0, s>>>(d, d + 4096, ne); }, it, s));
    for (int ne : {0, 4})
        printf("reduce+%2d exp<<<1,256>>>    %.2f us/launch\n", ne, time_loop([&] { k_reduce_exp<<<1, 256, 0, s>>>(d, d + 4096, ne); }, it, s));
    printf("kernarg->LDS<<<1024,256>>> %.2f us/launch\n", time_loop([&] { k_kernarg_lds<<<1024, 256, 0, s>>>(big, fo); }, it, s));
    printf("kernarg sgpr<<<1024,256>>> %.2f us/launch\n", time_loop([&] { k_kernarg_sgpr<<<1024, 256, 0, s>>>(big, fo); }, it, s));

    // producer/consumer pair: kernel A writes partials, kernel B reduces them (dependent boundary)
    printf("pair touch+reduce          %.2f us/pair\n", time_loop([&] { k_touch<<<1, 64, 0, s>>>(d); k_reduce_exp<<<1024, 256, 0, s>>>(d, d + 4096, 1); }, it, s));

    const unsigned int N = 1000000;
    float4 *p, *o1, *o2;
    CHECK(hipMalloc(&p, (size_t)N * 16 * 4));
    CHECK(hipMalloc(&o1, (size_t)N * 16 * 4));
    CHECK(hipMalloc(&o2, (size_t)N * 16 * 4));
    CHECK(hipMemset(p, 0, (size_t)N * 16 * 4));
    for (unsigned int n : {1000u, 250000u, 1000000u, 4000000u})
        for (int blocks : {256, 1024, 4096})
            {
            float t1 = time_loop([&] { k_stream_read<<<blocks, 256, 0, s>>>(p, n, fo); }, 500, s);
            float t2 = time_loop([&] { k_stream_rw<<<blocks, 256, 0, s>>>(p, o1, o2, n); }, 500, s);
            printf("n=%8u blocks=%5d  read16B %.2f us (%.0f GB/s)   r16+w32 %.2f us (%.0f GB/s)\n", n, blocks, t1, n * 16.0 / t1 / 1e3, t2,
                   n * 48.0 / t2 / 1e3);
            }
    CHECK(hipDeviceSynchronize());
    return 0;
    }
