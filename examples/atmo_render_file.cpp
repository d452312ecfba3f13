// atmo_render_file.cpp -- a host that uses ONLY the C ABI (include/atmo.h) and the HIP runtime: no Python, no torch.
// It is what a GDExtension or any other native host does: create a context for a shader variant,
// set uniforms by the reference's names, bake the optical-depth LUT on the device, draw, read the frame back.
//
//   hipcc -O2 -I include examples/atmo_render_file.cpp -L godot_atmosphere_shader_amd -latmo_hip \
//         -Wl,-rpath,$PWD/godot_atmosphere_shader_amd -o atmo_render_file
//   ./atmo_render_file <frame.bin> <depth.bin> <out.bin> <planet_radius> <atmosphere_height> <u_density> <view_steps>
//
// frame.bin = one AtmoFrame struct; depth.bin = viewport_h*viewport_w floats; out.bin = rect RGBA float4.
// tests/test_gpu_parity.py::test_native_host_matches_python_binding checks the bytes against the Python path.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "atmo.h"

#define CHECK_ATMO(call)                                                                       \
    do {                                                                                       \
        int rc_ = (call);                                                                      \
        if (rc_ != ATMO_OK) {                                                                  \
            std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, atmo_last_error_string(ctx)); \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)
#define CHECK_HIP(call)                                                              \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) {                                                      \
            std::fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_));        \
            return 1;                                                                \
        }                                                                            \
    } while (0)

static bool read_file(const char *path, void *dst, size_t bytes) {
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    const size_t n = std::fread(dst, 1, bytes, f);
    std::fclose(f);
    return n == bytes;
}

int main(int argc, char **argv) {
    if (argc != 8) {
        std::fprintf(stderr, "usage: %s frame.bin depth.bin out.bin planet_radius atmosphere_height u_density view_steps\n", argv[0]);
        return 2;
    }
    AtmoContext *ctx = nullptr;
    AtmoFrame frame;
    if (!read_file(argv[1], &frame, sizeof(frame))) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    const size_t npix = (size_t)frame.viewport_w * frame.viewport_h;
    std::vector<float> depth(npix);
    if (!read_file(argv[2], depth.data(), npix * sizeof(float))) { std::fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
    const float radius = (float)std::atof(argv[4]), height = (float)std::atof(argv[5]), density = (float)std::atof(argv[6]);
    const int view_steps = std::atoi(argv[7]);

    if (atmo_abi_version() != ATMO_ABI_VERSION) { std::fprintf(stderr, "ABI mismatch\n"); return 1; }
    int rc = atmo_create(0, ATMO_VARIANT_NO_CLOUDS, view_steps, 0, ATMO_LIGHT_LUT, 0, &ctx);
    if (rc != ATMO_OK) { std::fprintf(stderr, "atmo_create -> %d: %s\n", rc, atmo_last_error_string(nullptr)); return 1; }
    // planet_atmosphere.gd:114-115 and the demo's shader_params (planet_atmosphere_test.tscn:97-104)
    CHECK_ATMO(atmo_set_param_f32(ctx, "u_planet_radius", &radius, 1));
    CHECK_ATMO(atmo_set_param_f32(ctx, "u_atmosphere_height", &height, 1));
    CHECK_ATMO(atmo_set_param_f32(ctx, "u_density", &density, 1));
    const float strength = 1.0f;
    CHECK_ATMO(atmo_set_param_f32(ctx, "u_scattering_strength", &strength, 1));
    if (atmo_set_param_f32(ctx, "u_not_a_uniform", &strength, 1) != ATMO_E_NAME) { std::fprintf(stderr, "expected ATMO_E_NAME\n"); return 1; }
    CHECK_ATMO(atmo_bake_optical_depth(ctx, nullptr));  // replaces OpticalDepthBaker's SubViewport round trip

    const size_t rect_pix = (size_t)(frame.x1 - frame.x0) * (frame.y1 - frame.y0);
    float *d_depth = nullptr, *d_rgba = nullptr;
    CHECK_HIP(hipMalloc((void **)&d_depth, npix * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_rgba, rect_pix * 4 * sizeof(float)));
    CHECK_HIP(hipMemcpy(d_depth, depth.data(), npix * sizeof(float), hipMemcpyHostToDevice));
    CHECK_ATMO(atmo_render(ctx, &frame, d_depth, d_rgba, nullptr));
    CHECK_HIP(hipDeviceSynchronize());
    std::vector<float> rgba(rect_pix * 4);
    CHECK_HIP(hipMemcpy(rgba.data(), d_rgba, rgba.size() * sizeof(float), hipMemcpyDeviceToHost));
    FILE *f = std::fopen(argv[3], "wb");
    if (!f || std::fwrite(rgba.data(), sizeof(float), rgba.size(), f) != rgba.size()) { std::fprintf(stderr, "cannot write %s\n", argv[3]); return 1; }
    std::fclose(f);
    std::printf("atmo_render: %zu pixels shaded\n", rect_pix);
    (void)hipFree(d_depth);
    (void)hipFree(d_rgba);
    CHECK_ATMO(atmo_destroy(ctx));
    return 0;
}
