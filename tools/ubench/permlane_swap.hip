#include <hip/hip_runtime.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *o) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    u2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    u2 s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1]; o[128 + threadIdx.x] = s[0]; o[192 + threadIdx.x] = s[1];
}
int main() {
    unsigned *d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    for (int v = 0; v < 4; ++v) { for (int i = 0; i < 64; i += 8) printf("%u ", h[64 * v + i]); printf("\n"); }
    return 0;
}
