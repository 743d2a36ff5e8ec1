// Probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read): lane l supplies the address of its own 8-byte piece (4 x 16 bit);
// prints which piece element each lane receives.  lds[i] = i, lane l's piece = elements 4l .. 4l+3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(int* out) {
    __shared__ short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + (l * 4)));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    int* d; (void)hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d (piece of lane %2d, elem %d)", h[l * 4 + j], h[l * 4 + j] / 4, h[l * 4 + j] % 4); printf("\n"); if (l == 17) l = 46; }
    return 0;
}
