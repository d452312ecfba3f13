// valu_peak.hip -- measures the VALU / transcendental issue ceilings of the device the raymarch kernels are
// priced against (DESIGN.md "roofline").  Standalone: hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o valu_peak
// Each kernel issues ITER x 16 independent ops per lane on 8 accumulators-pairs, 256 CUs x 8 waves/SIMD resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITER = 4096;
constexpr int UNROLL = 16;

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, float seed) {
    float a[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) a[i] = seed + (float)(threadIdx.x + i) * 1e-3f;
    const float m = 1.0000001f, c = 1e-7f;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if (OP == 5) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
            if (OP == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 8) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 9) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a[i]));
            if (OP == 10) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
            if (OP == 11) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c));
        }
        if (OP == 7) {
#pragma unroll
            for (int i = 0; i < UNROLL; i += 2) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 v = {a[i], a[i + 1]}, mm = {m, m}, cc = {c, c};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(mm), "v"(cc));
                a[i] = v.x; a[i + 1] = v.y;
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
int run(const char *name, double ops_per_inst, float *d_out, int blocks) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double insts_per_wave = (OP == 7 ? (double)ITER * UNROLL / 2 : (double)ITER * UNROLL);
    const double waves = (double)blocks * 4;
    const double winst_per_s = insts_per_wave * waves / (best * 1e-3);
    // per SIMD: 1024 SIMDs on the chip
    printf("{\"op\": \"%s\", \"ms\": %.4f, \"wave_inst_per_s\": %.4e, \"lane_ops_per_s\": %.4e, \"cycles_per_wave_inst_per_simd_at_2.4GHz\": %.3f}\n",
           name, best, winst_per_s, winst_per_s * 64 * ops_per_inst, 2.4e9 * 1024.0 / winst_per_s);
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    const int blocks = p.multiProcessorCount * 8;  // 8 blocks x 4 waves = 32 waves/CU = 8 waves/SIMD
    float *d_out;
    CHECK(hipMalloc(&d_out, (size_t)blocks * 256 * sizeof(float)));
    if (run<0>("v_fma_f32", 1, d_out, blocks)) return 1;
    if (run<1>("v_mul_f32", 1, d_out, blocks)) return 1;
    if (run<2>("v_add_f32", 1, d_out, blocks)) return 1;
    if (run<8>("v_max_f32", 1, d_out, blocks)) return 1;
    if (run<7>("v_pk_fma_f32", 2, d_out, blocks)) return 1;
    if (run<3>("v_exp_f32", 1, d_out, blocks)) return 1;
    if (run<4>("v_sqrt_f32", 1, d_out, blocks)) return 1;
    if (run<5>("v_rsq_f32", 1, d_out, blocks)) return 1;
    if (run<6>("v_rcp_f32", 1, d_out, blocks)) return 1;
    if (run<9>("v_cvt_f32_ubyte0", 1, d_out, blocks)) return 1;
    if (run<10>("v_floor_f32", 1, d_out, blocks)) return 1;
    if (run<11>("v_cndmask_b32", 1, d_out, blocks)) return 1;
    CHECK(hipFree(d_out));
    return 0;
}
