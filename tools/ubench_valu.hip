// Micro-benchmark (gfx950): issue rate of v_add_f32 / v_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 /
// v_pk_fma_f32 with an SGPR operand, N independent chains per wave. Answers one design question of
// the scan kernel: do packed-f32 VALU ops cost one issue slot (2x elements per slot) or two?
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float s) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f2{(float)threadIdx.x * 1e-3f + i, 1.0f + i};
    f2 sv = {s, s * 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i].x) : "s"(s));
                if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i].x) : "s"(s));
                if (MODE == 2) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sv));
                if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sv));
                if (MODE == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "s"(sv));
                if (MODE == 5) asm volatile("v_pk_add_f32 %0, %1, %0 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a[i]) : "s"(sv));
                if (MODE == 6) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i].x) : "s"(s));
            }
        }
    }
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    int dev = 0;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, dev);
    const int blocks = p.multiProcessorCount * 8, iters = 4096;
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[] = {"v_add_f32", "v_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_pk_add_f32(opsel,neg)", "v_mul_f32"};
    void (*fn[])(float*, int, float) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>};
    printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    for (int wpb = 1; wpb <= 2; ++wpb)
    for (int m = 0; m < 7; ++m) {
        const int nb = blocks / (wpb == 1 ? 2 : 1);   // 4 or 8 blocks of 256 threads per CU => 4 / 8 waves per SIMD
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(fn[m], dim3(nb), dim3(256), 0, 0, out, iters, 1.0001f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double winstr = (double)nb * 4 * iters * 32;   // wave-instructions
        const double per_simd = winstr / (p.multiProcessorCount * 4);
        printf("%-26s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)\n", names[m], nb * 4 / (p.multiProcessorCount * 4),
               ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    }
    return 0;
}
