// Probe of the operand / result layout of v_mfma_i32_16x16x32_i8 on gfx950 (the K7 matrix-core kernel relies on it):
// A[m][k] = 1 only at (m0, k0), B[k][n] = 1 only at (k0, n0)  ->  D[m0][n0] = 1; which lane / register holds it, and
// which lane / byte the two ones have to be put in, is what the loops below search for.
// hipcc --offload-arch=gfx950 -O2 mfma_i8_probe.hip -o mfma_i8_probe && ./mfma_i8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(int la, int ba, int lb, int bb, int *out)
{
    const int lane = threadIdx.x;
    long a = 0, b = 0;
    if (lane == la) a = 1L << (8 * ba);
    if (lane == lb) b = 1L << (8 * bb);
    v4i c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[4 * lane + r] = c[r];
}
int main()
{
    int *d, h[256];
    hipMalloc(&d, sizeof(h));
    // operand lane l, byte j  <->  row (or column) l % 16, k = 8 (l / 16) + j : put A at (row 3, k 13), B at (k 13, col 5)
    const int la = 3 + 16 * (13 / 8), ba = 13 % 8, lb = 5 + 16 * (13 / 8), bb = 13 % 8;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, la, ba, lb, bb, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r)
            if (h[4 * l + r]) printf("D = %d at lane %d reg %d  (expected: column 5 = lane %% 16, row 3 = 4 (lane / 16) + reg)\n", h[4 * l + r], l, r);
    return 0;
}
