// Probe of ds_read_b64_tr_b8 on gfx950: which LDS bytes does lane l receive when lane l supplies address 8*l?
// hipcc --offload-arch=gfx950 -O2 tr8_probe.hip -o tr8_probe && ./tr8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void k(int hi, unsigned *out) {
    __shared__ __attribute__((aligned(16))) unsigned char s[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = hi ? (i >> 8) : (i & 255);
    __syncthreads();
    const int lane = threadIdx.x;
    v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i *)(s + lane * 8));
    out[2 * lane] = (unsigned)r[0];
    out[2 * lane + 1] = (unsigned)r[1];
}
int main() {
    unsigned *d, lo[128], hi[128];
    hipMalloc(&d, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, 0, d); hipMemcpy(lo, d, 512, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, 1, d); hipMemcpy(hi, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int b = 0; b < 8; ++b) {
            unsigned lb = (lo[2 * l + b / 4] >> (8 * (b % 4))) & 255, hb = (hi[2 * l + b / 4] >> (8 * (b % 4))) & 255;
            printf(" %4u", hb * 256 + lb);
        }
        printf("\n");
    }
    return 0;
}
