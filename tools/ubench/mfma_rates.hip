// mfma_rates.hip -- what the matrix cores of this MI355X sustain on RANDOM operands, shape by shape (the chip lowers its clock
// under matrix load: MI355X_MICROARCH.md 'DVFS give-back'), to decide the operand type of K8's pre-filter:
//   v_mfma_f32_32x32x16_f16, v_mfma_f32_16x16x32_f16, v_mfma_i32_32x32x32_i8, v_mfma_i32_16x16x64_i8
// Operands in registers (no LDS, no memory), NACC independent accumulators per wave, WPS waves per SIMD, every CU busy,
// launches repeated for >= 1.5 s before the timed ones.  Prints MAC/s, the in-kernel clock (s_memtime / s_memrealtime) and
// cycles per MFMA.   hipcc -O3 --offload-arch=gfx950 -o mfma_rates mfma_rates.hip && ./mfma_rates
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));
typedef int i4v __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

constexpr int NACC = 4;

template <int SHAPE>
__global__ __launch_bounds__(512) void k_rate(const uint32_t *__restrict__ seed, int iters, float *__restrict__ sink,
                                              unsigned long long *__restrict__ stamps)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s[8];
    for (int i = 0; i < 8; ++i) s[i] = seed[(tid * 8 + i) & 0xffff];
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float out = 0.0f;
    if constexpr (SHAPE == 0 || SHAPE == 1) {
        h8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)((int)(s[i] & 1023) - 512) * (_Float16)0.001f; b[i] = (_Float16)((int)((s[i] >> 10) & 1023) - 512) * (_Float16)0.001f; }
        if constexpr (SHAPE == 0) {
            f16v acc[NACC];
            for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
            }
            for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) out += acc[n][r];
        } else {
            f4v acc[NACC];
            for (int n = 0; n < NACC; ++n) for (int r = 0; r < 4; ++r) acc[n][r] = 0.0f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[n], 0, 0, 0);
            }
            for (int n = 0; n < NACC; ++n) for (int r = 0; r < 4; ++r) out += acc[n][r];
        }
    } else {
        i4v a, b;
        for (int i = 0; i < 4; ++i) { a[i] = (int)(s[i] * 2654435761u); b[i] = (int)(s[i + 4] * 40503u + s[i]); }
        if constexpr (SHAPE == 2) {
            i16v acc[NACC];
            for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[n], 0, 0, 0);
            }
            for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) out += (float)acc[n][r];
        } else {
            i4v acc[NACC];
            for (int n = 0; n < NACC; ++n) for (int r = 0; r < 4; ++r) acc[n][r] = 0;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[n], 0, 0, 0);
            }
            for (int n = 0; n < NACC; ++n) for (int r = 0; r < 4; ++r) out += (float)acc[n][r];
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (out == 12345.678f) sink[tid] = out;
    if ((threadIdx.x & 63) == 0 && stamps) {
        const int w = tid >> 6;
        stamps[2 * w] = c1 - c0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

template <int SHAPE>
static void run(const char *name, double macs_per_mfma, int wps, const uint32_t *dseed, float *sink, unsigned long long *dst, int cus)
{
    const int threads = 64 * 4 * wps, blocks = cus, iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // warm: >= 1.5 s of back-to-back launches
    float ms = 0.0f;
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<SHAPE>, dim3(blocks), dim3(threads), 0, 0, dseed, iters, sink, dst);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const int warm = (int)(1500.0f / (ms > 0.01f ? ms : 0.01f)) + 1;
    for (int i = 0; i < warm; ++i) hipLaunchKernelGGL(k_rate<SHAPE>, dim3(blocks), dim3(threads), 0, 0, dseed, iters, sink, dst);
    const int reps = 20;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_rate<SHAPE>, dim3(blocks), dim3(threads), 0, 0, dseed, iters, sink, dst);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const int waves = blocks * threads / 64;
    std::vector<unsigned long long> st(2 * (size_t)waves);
    CHECK(hipMemcpy(st.data(), dst, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> clk;
    double cyc = 0.0;
    for (int w = 0; w < waves; ++w) { clk.push_back((double)st[2 * w] / (double)st[2 * w + 1] * 100.0); cyc += (double)st[2 * w]; }
    std::sort(clk.begin(), clk.end());
    const double mfmas = (double)waves * iters * NACC, per_launch_s = ms * 1e-3 / reps;
    printf("%-28s %d wave(s)/SIMD: %7.1f T MAC/s (%6.1f T op/s)  in-kernel clock %4.0f MHz  %5.1f cycles per MFMA per SIMD\n", name, wps,
           mfmas * macs_per_mfma / per_launch_s / 1e12, 2.0 * mfmas * macs_per_mfma / per_launch_s / 1e12, clk[clk.size() / 2],
           cyc / waves / ((double)iters * NACC) / wps);
}

#include <algorithm>

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs\n", prop.name, cus);
    std::vector<uint32_t> seed(65536);
    uint32_t x = 12345;
    for (auto &v : seed) { x = x * 1664525u + 1013904223u; v = x; }
    uint32_t *dseed;
    float *sink;
    unsigned long long *dst;
    CHECK(hipMalloc(&dseed, seed.size() * 4));
    CHECK(hipMemcpy(dseed, seed.data(), seed.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&sink, (size_t)cus * 512 * 4));
    CHECK(hipMalloc(&dst, (size_t)cus * 8 * 2 * sizeof(unsigned long long)));
    for (int wps = 1; wps <= 2; ++wps) {
        run<0>("v_mfma_f32_32x32x16_f16", 32.0 * 32 * 16, wps, dseed, sink, dst, cus);
        run<1>("v_mfma_f32_16x16x32_f16", 16.0 * 16 * 32, wps, dseed, sink, dst, cus);
        run<2>("v_mfma_i32_32x32x32_i8", 32.0 * 32 * 32, wps, dseed, sink, dst, cus);
        run<3>("v_mfma_i32_16x16x64_i8", 16.0 * 16 * 64, wps, dseed, sink, dst, cus);
    }
    return 0;
}
