// Probe (round 4): does s_wqm_b64 + v_mov_b32_dpp inside one inline-asm block let an ACTIVE lane read a value computed, inside the
// block, by an INACTIVE quad mate (a "helper lane", as the graphics pipeline's whole-quad mode provides for derivatives)?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wqm_probe tools/probes/wqm_probe.hip && /tmp/wqm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
struct QC { float a, ax, ay; };
__device__ __forceinline__ QC ex(float x, float r0) {
    QC o; unsigned long long save;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "s_wqm_b64 exec, exec\n\t"
        "v_mul_f32_e32 %[a], %[r0], %[x]\n\t"
        "s_nop 1\n\t"
        "v_mov_b32_dpp %[ax], %[a] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[ay], %[a] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 exec, %[save]"
        : [save] "=&s"(save), [a] "=&v"(o.a), [ax] "=&v"(o.ax), [ay] "=&v"(o.ay)
        : [x] "v"(x), [r0] "s"(r0)
        : "scc");
    return o;
}
__global__ void k(const float *p, float *out, float r0) {
    const float x = p[threadIdx.x];
    float rx = -1.f, ry = -1.f, keep = x * 3.0f;   // `keep` is live across the branch for the inactive lanes
    if (x > 0.5f) { QC q = ex(x, r0); rx = q.ax; ry = q.ay; }
    out[3 * threadIdx.x] = rx; out[3 * threadIdx.x + 1] = ry; out[3 * threadIdx.x + 2] = keep;
}
int main() {
    float h[64], *d, *o, ho[192];
    for (int i = 0; i < 64; ++i) h[i] = (i % 4 == 0) ? float(i + 1) : 0.001f * float(i + 1);   // one active lane per quad
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 2.0f);
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        const bool act = i % 4 == 0;
        const float wx = act ? 2.0f * h[i ^ 1] : -1.f, wy = act ? 2.0f * h[i ^ 2] : -1.f;
        if (ho[3 * i] != wx || ho[3 * i + 1] != wy || ho[3 * i + 2] != h[i] * 3.0f) { if (bad < 6) printf("lane %d: got %g %g %g want %g %g %g\n", i, ho[3*i], ho[3*i+1], ho[3*i+2], wx, wy, h[i]*3.0f); ++bad; }
    }
    printf("wqm probe: %d bad lanes of 64\n", bad);
    return bad != 0;
}
