// Micro-benchmark: LDS-pipe cost of the operations the delivery kernel issues per synapse word on MI355X.
// Every workgroup (256 threads, 5 waves/SIMD resident like deliver_kernel) runs ITER rounds of one operation with
// pseudo-random per-lane indices and `active` of 64 lanes enabled; reported: cycles per wave-instruction per CU
// (= kernel time x clock / instructions issued per CU).
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o lds_ops lds_ops.hip ; run: ./lds_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 4096;
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// OP: 0 ds_add_f64 random over n, 1 ds_add_u32 random over n, 2 ds_read_u8 random over 256, 3 ds_read_b64 from a 32-entry
// table, 4 ds_read_b32 from a 32-entry table, 5 ds_add_f32, 6 nothing (the index arithmetic alone), 7 ds_add_u64
template <int OP> __global__ void __launch_bounds__(256) k_ops(uint32_t n, uint32_t active_pct, uint32_t *out)
{
    __shared__ double acc[2049];
    __shared__ uint8_t tab[256];
    __shared__ double lut[32];
    for (uint32_t i = threadIdx.x; i < 2049; i += 256) acc[i] = 0.0;
    tab[threadIdx.x] = (uint8_t) (threadIdx.x & 1);
    if (threadIdx.x < 32) lut[threadIdx.x] = threadIdx.x;
    __syncthreads();
    uint32_t r = mix(blockIdx.x * 256u + threadIdx.x + 1u);
    uint32_t sink = 0;
    double dsink = 0;
    uint32_t *acc32 = reinterpret_cast<uint32_t *>(acc);
    float *accf = reinterpret_cast<float *>(acc);
    unsigned long long *acc64 = reinterpret_cast<unsigned long long *>(acc);
    // the set of enabled lanes is fixed per wave (the bank conflicts depend on the indices, not on which lanes add)
    const bool on = (mix(threadIdx.x * 7919u + blockIdx.x) % 100u) < active_pct;
#pragma unroll 2
    for (int it = 0; it < ITER / 4; it++)
    {
        r = r * 1664525u + 1013904223u;
#pragma unroll
        for (int q = 0; q < 4; q++) // four operations per generator step: two VALU instructions per LDS instruction
        {
            const uint32_t idx = (r >> (6 + 5 * q)) & (n - 1u); // n: power of two
            if (OP == 0) { if (on) atomicAdd(&acc[idx], 1.0); }
            else if (OP == 1) { if (on) atomicAdd(&acc32[idx], 1u); }
            else if (OP == 2) sink += tab[idx & 255u];
            else if (OP == 3) dsink += lut[idx & 31u];
            else if (OP == 4) sink += reinterpret_cast<uint32_t *>(lut)[idx & 31u];
            else if (OP == 5) { if (on) atomicAdd(&accf[idx], 1.0f); }
            else if (OP == 7) { if (on) atomicAdd(&acc64[idx], 1ull); }
            else sink += idx + (on ? 1u : 0u);
        }
    }
    __syncthreads();
    if (sink == 0x12345u || dsink == 1.5) out[0] = sink + acc32[threadIdx.x];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = acc32[3];
}
template <int OP> void run(const char *name, uint32_t n, uint32_t pct, uint32_t *out, double base_us = 0)
{
    const int G = 256 * 5; // 5 workgroups of 4 waves per CU = 5 waves/SIMD
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) k_ops<OP><<<G, 256>>>(n, pct, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    const int reps = 10;
    for (int i = 0; i < reps; i++) k_ops<OP><<<G, 256>>>(n, pct, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps;
    const double inst_per_cu = (double) ITER * 4 * 5; // wave-instructions per CU
    printf("%-34s n=%5u active=%3u%%  %8.1f us  %6.2f cycles/wave-instr/CU at 2.4 GHz (minus index arithmetic: %6.2f)\n", name, n, pct, us,
           us * 2400.0 / inst_per_cu, (us - base_us) * 2400.0 / inst_per_cu);
}
int main()
{
    uint32_t *out;
    CK(hipMalloc(&out, 4096));
    run<6>("index arithmetic only", 1024, 34, out);
    run<0>("ds_add_f64", 1024, 100, out);
    run<0>("ds_add_f64", 1024, 34, out);
    run<0>("ds_add_f64", 512, 34, out);
    run<7>("ds_add_u64", 1024, 34, out);
    run<1>("ds_add_u32", 1024, 100, out);
    run<1>("ds_add_u32", 1024, 34, out);
    run<1>("ds_add_u32", 512, 34, out);
    run<5>("ds_add_f32", 1024, 34, out);
    run<2>("ds_read_u8 (256-byte table)", 1024, 100, out);
    run<3>("ds_read_b64 (32-entry table)", 1024, 100, out);
    run<4>("ds_read_b32 (32-entry table)", 1024, 100, out);
    return 0;
}
