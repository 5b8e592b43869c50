// Micro-benchmark: issue rate of a few gfx950 VALU instructions (cycles per wave instruction per SIMD).
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define ITER 4096

#define KERNEL(name, decl, body)                                                        \
    __global__ __launch_bounds__(256) void name(double *out, unsigned seed)             \
    {                                                                                   \
        decl;                                                                           \
        for (int i = 0; i < ITER; ++i) { REP16(body) }                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3; \
    }

#define DECL_D double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3; double c = 1.0000001
#define DECL_U unsigned u0 = seed, u1 = seed + 1, u2 = seed + 2, u3 = seed + 3; double a0 = 0, a1 = 0, a2 = 0, a3 = 0

KERNEL(k_add_f64, DECL_D,
       asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
KERNEL(k_fma_f64, DECL_D,
       asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
KERNEL(k_cvt_f64_u32, DECL_U,
       asm volatile("v_cvt_f64_u32 %0, %4\n v_cvt_f64_u32 %1, %5\n v_cvt_f64_u32 %2, %6\n v_cvt_f64_u32 %3, %7"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));)
KERNEL(k_cvt_f64_f32, DECL_U,
       asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));)
KERNEL(k_cvt_f32_ubyte, DECL_U,
       asm volatile("v_cvt_f32_ubyte0 %0, %0\n v_cvt_f32_ubyte1 %1, %1\n v_cvt_f32_ubyte2 %2, %2\n v_cvt_f32_ubyte3 %3, %3"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)); a0 = u0;)
KERNEL(k_and_b32, DECL_U,
       asm volatile("v_and_b32 %0, 0xffff, %0\n v_and_b32 %1, 0xffff, %1\n v_and_b32 %2, 0xffff, %2\n v_and_b32 %3, 0xffff, %3"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)); a0 = u0;)
#define DECL_W DECL_U; unsigned long long w0 = seed, w1 = seed, w2 = seed, w3 = seed
KERNEL(k_mad_u64_u32, DECL_W,
       asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3"
                    : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(u0), "v"(u1) : "vcc"); a0 = (double)w0; a1 = (double)w1; a2 = (double)w2; a3 = (double)w3;)
KERNEL(k_mul_lo_u32, DECL_U,
       asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)); a0 = u0;)
KERNEL(k_rcp_f64, DECL_D,
       asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
KERNEL(k_rsq_f64, DECL_D,
       asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
KERNEL(k_cndmask, DECL_U,
       asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)); a0 = u0;)
KERNEL(k_cmp_f64, DECL_D,
       asm volatile("v_cmp_lt_f64 vcc, %0, %4\n v_cmp_lt_f64 vcc, %1, %4\n v_cmp_lt_f64 vcc, %2, %4\n v_cmp_lt_f64 vcc, %3, %4"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "vcc");)
KERNEL(k_mov_dpp, DECL_U,
       asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)); a0 = u0;)
KERNEL(k_pk_fma_f32, DECL_D,
       asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)

KERNEL(k_cndmask_sgpr, DECL_U,
       asm volatile("v_cndmask_b32 %0, %0, %1, %4\n v_cndmask_b32 %1, %1, %2, %4\n v_cndmask_b32 %2, %2, %3, %4\n v_cndmask_b32 %3, %3, %0, %4"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "s"(0x5555555555555555ull)); a0 = u0;)
KERNEL(k_cndmask_vccset, DECL_U,
       asm volatile("s_mov_b64 vcc, 0x55\n s_nop 4\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : : "vcc"); a0 = u0;)
KERNEL(k_cmp_cndmask, DECL_U,
       asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_u32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : : "vcc"); a0 = u0;)
KERNEL(k_add_u32, DECL_U,
       asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)); a0 = u0;)
KERNEL(k_mul_f64, DECL_D,
       asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
KERNEL(k_fma_f32, DECL_U,
       asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %1, %1, %2, %2\n v_fma_f32 %2, %2, %3, %3\n v_fma_f32 %3, %3, %0, %0"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)); a0 = u0;)
KERNEL(k_max_f64, DECL_D,
       asm volatile("v_max_f64 %0, %0, %4\n v_max_f64 %1, %1, %4\n v_max_f64 %2, %2, %4\n v_max_f64 %3, %3, %4"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
KERNEL(k_bfe_u32, DECL_U,
       asm volatile("v_bfe_u32 %0, %0, 3, 9\n v_bfe_u32 %1, %1, 3, 9\n v_bfe_u32 %2, %2, 3, 9\n v_bfe_u32 %3, %3, 3, 9"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)); a0 = u0;)
KERNEL(k_lshl_add, DECL_U,
       asm volatile("v_lshl_add_u32 %0, %0, 3, %1\n v_lshl_add_u32 %1, %1, 3, %2\n v_lshl_add_u32 %2, %2, 3, %3\n v_lshl_add_u32 %3, %3, 3, %0"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)); a0 = u0;)
KERNEL(k_mov_b32, DECL_U,
       asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)); a0 = u0;)

template <typename K>
static void run(const char *name, K kern, double *d)
{
    const int blocks = 256 * 8, threads = 256; // 8 blocks (32 waves) per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kern<<<blocks, threads>>>(d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, threads>>>(d, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    // wave-instructions per SIMD: waves per SIMD = blocks*4/(256 CUs*4 SIMDs) = 8 ; each issues ITER*64
    const double insts_per_simd = (double)blocks * 4 / 1024.0 * ITER * 64.0;
    const double cycles = ms * 1e-3 * clk_khz * 1e3;
    printf("%-18s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (clock %d MHz)\n", name, ms, cycles / insts_per_simd, clk_khz / 1000);
}

int main()
{
    double *d;
    hipMalloc(&d, 256 * 8 * 256 * sizeof(double));
    run("v_add_f64", k_add_f64, d);
    run("v_fma_f64", k_fma_f64, d);
    run("v_cvt_f64_u32", k_cvt_f64_u32, d);
    run("v_cvt_f64_f32", k_cvt_f64_f32, d);
    run("v_cvt_f32_ubyteN", k_cvt_f32_ubyte, d);
    run("v_and_b32", k_and_b32, d);
    run("v_mad_u64_u32", k_mad_u64_u32, d);
    run("v_mul_lo_u32", k_mul_lo_u32, d);
    run("v_rcp_f64", k_rcp_f64, d);
    run("v_rsq_f64", k_rsq_f64, d);
    run("v_cndmask_b32", k_cndmask, d);
    run("v_cmp_lt_f64", k_cmp_f64, d);
    run("v_mov_b32_dpp", k_mov_dpp, d);
    run("v_pk_fma_f32", k_pk_fma_f32, d);
    run("v_cndmask sgpr", k_cndmask_sgpr, d);
    run("v_cndmask vccset", k_cndmask_vccset, d);
    run("cmp+cndmask (x2)", k_cmp_cndmask, d);
    run("v_add_u32", k_add_u32, d);
    run("v_mul_f64", k_mul_f64, d);
    run("v_fma_f32", k_fma_f32, d);
    run("v_max_f64", k_max_f64, d);
    run("v_bfe_u32", k_bfe_u32, d);
    run("v_lshl_add_u32", k_lshl_add, d);
    run("v_mov_b32", k_mov_b32, d);
    return 0;
}
