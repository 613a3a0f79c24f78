// What does ds_read_b64_tr_b16 deliver?  (gfx950; MI355X guide T10 says: each 16-lane group reads one [4][16] block of 16-bit
// elements, 4 contiguous elements per lane at the lane's own 8-byte-aligned address, and delivers it column-major.)
// Every 16-bit LDS element holds its own index; lane l reads at byte address 8 l (elements 4l .. 4l+3) and prints what it
// got: the printed source lane / element of every result word IS the instruction's lane mapping.
//     hipcc --offload-arch=gfx950 -O2 tools/probe/tr_read_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void probe(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
    for (int e = threadIdx.x; e < 1024; e += 64) lds[e] = (unsigned short)e;
    __syncthreads();
    const int l = threadIdx.x;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + 4 * l));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

int main() {
    unsigned short* d; unsigned short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("lane: result j=0..3 as (source lane, source element)\n");
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        printf("%2d:", l);
        for (int j = 0; j < 4; ++j) {
            printf(" (%2d,%d)", h[l * 4 + j] / 4, h[l * 4 + j] % 4);
            // hypothesis: inside a 16-lane group, result j of lane L = element (L & 3) of lane 4 j + (L >> 2)
            const int L = l & 15, g = l & ~15;
            if (h[l * 4 + j] != 4 * (g + 4 * j + (L >> 2)) + (L & 3)) ok = 0;
        }
        printf("\n");
    }
    printf("hypothesis out[L][j] = in[group + 4 j + (L >> 2)][L & 3]: %s\n", ok ? "CONFIRMED" : "REFUTED");
    return 0;
}
