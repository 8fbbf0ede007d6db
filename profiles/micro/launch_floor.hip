// Micro-benchmark: what does a launch of G workgroups x 256 threads cost on MI355X when each wave does
// (a) nothing, (b) one dependent chain of k global loads, with a small or a large by-value kernarg block?
// Build: hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip ; run: ./launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Big { unsigned long long p[110]; };
__global__ void __launch_bounds__(256) k_empty(int *out) { if (out == (int *) 1) *out = 0; }
__global__ void __launch_bounds__(256) k_big(Big b, int *out) { if (b.p[109] == 1) *out = (int) b.p[3]; }
// each lane: chain of `depth` dependent 8-byte loads through a permutation table, then one store
__global__ void __launch_bounds__(256) k_chain(const unsigned *next, unsigned n, int depth, unsigned *out)
{
    unsigned i = (blockIdx.x * 256u + threadIdx.x) % n;
    for (int d = 0; d < depth; d++) i = next[i];
    out[blockIdx.x * 256u + threadIdx.x] = i;
}
// streaming: each lane loads `k` independent 8-byte words (coalesced) and stores one
__global__ void __launch_bounds__(256) k_stream(const double *a, const double *b, const double *c, const double *d, double *o)
{
    const size_t i = blockIdx.x * 256u + threadIdx.x;
    o[i] = a[i] + b[i] + c[i] + d[i];
}
// the same stream with L bytes of static LDS (touched) and B workgroup barriers
template <int L, int B> __global__ void __launch_bounds__(256) k_stream_lds(const double *a, const double *b, const double *c, const double *d, double *o)
{
    __shared__ double s[L / 8 > 0 ? L / 8 : 1];
    const size_t i = blockIdx.x * 256u + threadIdx.x;
    const double x = a[i] + b[i] + c[i] + d[i];
    if (L > 0) s[threadIdx.x] = x;
    if (B > 0) __syncthreads();
    double y = (L > 0) ? s[threadIdx.x ^ 1] : x;
    if (B > 1) { __syncthreads(); if (L > 0) s[threadIdx.x] = y; __syncthreads(); y += (L > 0) ? s[threadIdx.x ^ 2] : 0.0; }
    o[i] = y;
}
// the stream plus `n` dependent fp64 operations per lane
__global__ void __launch_bounds__(256) k_stream_alu(const double *a, const double *b, const double *c, const double *d, double *o, int n)
{
    const size_t i = blockIdx.x * 256u + threadIdx.x;
    double x = a[i] + b[i] + c[i] + d[i];
    for (int k = 0; k < n; k++) x = x * 1.0000001 + 0.5;
    o[i] = x;
}
template <typename F> double time_us(F &&launch, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 50; i++) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3 * ms / reps;
}
int main()
{
    int *out;
    CK(hipMalloc(&out, 4));
    const unsigned n = 1u << 20;
    std::vector<unsigned> perm(n);
    for (unsigned i = 0; i < n; i++) perm[i] = (i * 2654435761u + 12345u) % n;
    unsigned *next, *o2;
    CK(hipMalloc(&next, n * 4));
    CK(hipMalloc(&o2, (size_t) 8192 * 256 * 4));
    CK(hipMemcpy(next, perm.data(), n * 4, hipMemcpyHostToDevice));
    double *a, *b, *c, *d, *o;
    const size_t ns = (size_t) 8192 * 256;
    CK(hipMalloc(&a, ns * 8)); CK(hipMalloc(&b, ns * 8)); CK(hipMalloc(&c, ns * 8)); CK(hipMalloc(&d, ns * 8)); CK(hipMalloc(&o, ns * 8));
    Big big{};
    for (int grid : {256, 1024, 4096, 8192})
    {
        printf("grid %5d: empty %.2f us", grid, time_us([&] { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, 0, out); }, 2000));
        printf("  big-kernarg %.2f us", time_us([&] { hipLaunchKernelGGL(k_big, dim3(grid), dim3(256), 0, 0, big, out); }, 2000));
        for (int depth : {1, 2, 4})
            printf("  chain%d %.2f us", depth, time_us([&] { hipLaunchKernelGGL(k_chain, dim3(grid), dim3(256), 0, 0, next, n, depth, o2); }, 2000));
        printf("  stream(40B/lane) %.2f us\n", time_us([&] { hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, a, b, c, d, o); }, 2000));
        printf("            stream+barrier %.2f", time_us([&] { hipLaunchKernelGGL((k_stream_lds<0, 1>), dim3(grid), dim3(256), 0, 0, a, b, c, d, o); }, 2000));
        printf("  +2KB lds,1 bar %.2f", time_us([&] { hipLaunchKernelGGL((k_stream_lds<2048, 1>), dim3(grid), dim3(256), 0, 0, a, b, c, d, o); }, 2000));
        printf("  +12KB lds,1 bar %.2f", time_us([&] { hipLaunchKernelGGL((k_stream_lds<12288, 1>), dim3(grid), dim3(256), 0, 0, a, b, c, d, o); }, 2000));
        printf("  +12KB lds,3 bar %.2f", time_us([&] { hipLaunchKernelGGL((k_stream_lds<12288, 3>), dim3(grid), dim3(256), 0, 0, a, b, c, d, o); }, 2000));
        for (int nalu : {50, 200, 400})
            printf("  alu%d %.2f", nalu, time_us([&] { hipLaunchKernelGGL(k_stream_alu, dim3(grid), dim3(256), 0, 0, a, b, c, d, o, nalu); }, 2000));
        printf("\n");
    }
    return 0;
}
