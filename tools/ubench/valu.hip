// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2_ __attribute__((ext_vector_type(2)));
#define N_ITER 4096
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, float seed, long long* cyc) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float m = 1.0000001f;
    const float2_ pm = {m, m};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N_ITER; ++i) {
        if constexpr (KIND == 0) {  // 8 independent v_mul_f32
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if constexpr (KIND == 1) {  // 4 independent v_pk_mul_f32 (= 8 multiplies)
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));
        } else if constexpr (KIND == 2) {  // 8 v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                         "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if constexpr (KIND == 3) {  // 8 v_rcp_f32
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                         "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == 4) {  // 8 IEEE divisions (compiler sequence)
            a0 = a0 / m; a1 = a1 / m; a2 = a2 / m; a3 = a3 / m; a4 = a4 / m; a5 = a5 / m; a6 = a6 / m; a7 = a7 / m;
        } else if constexpr (KIND == 5) {  // 8 v_div_fixup_f32
            asm volatile("v_div_fixup_f32 %0, %0, %8, %8\n v_div_fixup_f32 %1, %1, %8, %8\n v_div_fixup_f32 %2, %2, %8, %8\n v_div_fixup_f32 %3, %3, %8, %8\n"
                         "v_div_fixup_f32 %4, %4, %8, %8\n v_div_fixup_f32 %5, %5, %8, %8\n v_div_fixup_f32 %6, %6, %8, %8\n v_div_fixup_f32 %7, %7, %8, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if constexpr (KIND == 6) {  // 8 v_cmp + v_cndmask pairs
            asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc\n"
                         "v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_lt_f32 vcc, %6, %7\n v_cndmask_b32 %6, %6, %7, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");
        } else if constexpr (KIND == 7) {  // 4 v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));
        } else if constexpr (KIND == 8) {  // 8 v_sqrt_f32
            asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                         "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int KIND>
void run(const char* name, int ops_per_iter) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 8 * 4); hipMalloc(&cyc, 8);
    for (int waves_per_simd : {1, 2, 4, 8}) {
        // 256 CUs, blocks of 256 threads = 1 wave per SIMD per block; waves_per_simd blocks per CU
        hipLaunchKernelGGL(k<KIND>, dim3(256 * waves_per_simd), dim3(256), 0, 0, out, 1.5f, cyc);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256 * waves_per_simd), dim3(256), 0, 0, out, 1.5f, cyc);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        // SIMD-cycles per wave-instruction, assuming 2.4 GHz nominal for the wall-clock figure
        double wave_instrs_per_simd = double(N_ITER) * ops_per_iter * waves_per_simd;
        printf("%-14s waves/SIMD=%d  in-kernel cycles/instr (one wave) = %6.2f   wall: %.3f ms -> %.2f ns per wave-instr per SIMD\n", name, waves_per_simd,
               double(c) / (double(N_ITER) * ops_per_iter), ms, ms * 1e6 / wave_instrs_per_simd);
    }
}
int main() {
    run<0>("v_mul_f32", 8); run<1>("v_pk_mul_f32", 4); run<2>("v_fma_f32", 8); run<7>("v_pk_fma_f32", 4); run<3>("v_rcp_f32", 8);
    run<8>("v_sqrt_f32", 8); run<5>("v_div_fixup", 8); run<6>("cmp+cndmask", 8); run<4>("ieee fdiv", 8);
    return 0;
}
