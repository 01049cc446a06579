// Microbenchmark: sustained issue rate of v_fma_f32, v_pk_fma_f32, v_rcp_f32 and the IEEE
// division sequence on gfx950, per CU, as a function of waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_microbench.hip -o /tmp/valu_microbench && /tmp/valu_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2_ __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    float a[8];
    float2_ p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = float2_{a[i], a[i] + 1.0f}; }
    const float m = seed * 0.999f, c = seed * 0.001f;
    const float2_ pm{m, m}, pc{c, c};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
                if (MODE == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (MODE == 3) a[i] = m / a[i];
                if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
                if (MODE == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 6) a[i] = __builtin_sqrtf(a[i] + m);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, double ops_per_inst) {
    float* out;
    hipMalloc(&out, 256 * 256 * 32 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int blocks_per_cu : {1, 2, 4, 8}) {
        const int blocks = 256 * blocks_per_cu;
        k<MODE><<<blocks, 256>>>(out, 10, 1.5f);
        hipEventRecord(e0);
        k<MODE><<<blocks, 256>>>(out, iters, 1.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double insts = double(blocks) * 4 /*waves*/ * iters * 32;   // wave-instructions
        const double lane_ops = insts * 64 * ops_per_inst;
        printf("%-14s waves/SIMD=%d  %.3f ms  %.1f Ginst/s(wave)  %.2f Tlaneop/s  cycles/inst/SIMD@2.4GHz=%.2f\n", name, blocks_per_cu, ms,
               insts / ms / 1e6, lane_ops / ms / 1e9, (ms * 1e-3 * 2.4e9) / (insts / (256.0 * 4)));
    }
    hipFree(out);
}

int main() {
    run<0>("v_fma_f32", 1);
    run<1>("v_pk_fma_f32", 2);
    run<5>("v_mul_f32", 1);
    run<4>("v_pk_mul_f32", 2);
    run<2>("v_rcp_f32", 1);
    run<3>("ieee div", 1);
    run<6>("ieee sqrt", 1);
    return 0;
}
