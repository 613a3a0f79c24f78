// Issue rate of v_mfma_f32_32x32x16_bf16 by where the accumulators live and how many independent ones rotate:
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_tmp/mfma_rate tools/probe/mfma_rate.hip && gpurun -- ./gpurun_tmp/mfma_rate
// One wave per SIMD (256 threads per workgroup, one workgroup per CU would be 4 waves: here 1 workgroup of 256 threads on one CU) and two
// (512 threads).  Per variant: cycles (s_memtime) per MFMA over a 48-MFMA block repeated 64 times.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool AGPR>
__global__ void rate_kernel(unsigned long long* out, const float* seed) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = seed[r];
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)seed[e]; y[e] = (__bf16)seed[8 + e]; }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < 64; ++rep) {
#pragma unroll
        for (int k = 0; k < 48; ++k) {
            if constexpr (AGPR)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[k % NACC]) : "v"(x), "v"(y));
            else
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[k % NACC]) : "v"(x), "v"(y));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (threadIdx.x == 0) { out[0] = t1 - t0; }
    if (s == 123.456f) out[1] = 1;
}

template <int NACC, bool AGPR>
static void run(const char* name, int threads, unsigned long long* d_out, const float* d_seed) {
    unsigned long long h = 0;
    for (int i = 0; i < 3; ++i) {
        hipLaunchKernelGGL((rate_kernel<NACC, AGPR>), dim3(1), dim3(threads), 0, 0, d_out, d_seed);
        hipDeviceSynchronize();
    }
    hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
    printf("%-44s %4d threads: %7.2f cycles per MFMA per wave-slot (%llu cycles / 3072 MFMAs per wave)\n", name, threads, (double)h / 3072.0, h);
}

int main() {
    unsigned long long* d_out; float* d_seed; float hs[16];
    for (int i = 0; i < 16; ++i) hs[i] = 0.001f * (i + 1);
    hipMalloc(&d_out, 64); hipMalloc(&d_seed, 64); hipMemcpy(d_seed, hs, 64, hipMemcpyHostToDevice);
    for (int threads : {256, 512}) {
        run<1, false>("1 accumulator (dependent chain), VGPR", threads, d_out, d_seed);
        run<2, false>("2 accumulators alternating, VGPR", threads, d_out, d_seed);
        run<4, false>("4 accumulators rotating, VGPR", threads, d_out, d_seed);
        run<8, false>("8 accumulators rotating, VGPR", threads, d_out, d_seed);
        run<1, true>("1 accumulator (dependent chain), AGPR", threads, d_out, d_seed);
        run<2, true>("2 accumulators alternating, AGPR", threads, d_out, d_seed);
        run<4, true>("4 accumulators rotating, AGPR", threads, d_out, d_seed);
        run<8, true>("8 accumulators rotating, AGPR", threads, d_out, d_seed);
    }
    return 0;
}
