/* abi_smoke.c -- the C-ABI boundary from plain C99 (no Python, no C++): gcc -std=c99 -pedantic -Wall -Werror abi_smoke.c -llinevis_hip.
 *
 * What a LineVis-side binding does through include/linevis_hip.h, on the small fixture
 * tests/golden/abi_smoke.bin (inputs + the CPU checker's frames, written by tests/golden/make_golden.py):
 *   lv_create -> lv_set_lines -> lv_set_transfer_function -> lv_set_camera -> lv_set_option ... -> lv_render (mode 11: ray tracer with
 *   RTAO and depth cues; mode 2: per-pixel linked lists with the transparent transfer function) -> lv_get_stats -> lv_destroy,
 * and compares both frames with the fixture: +- 2 LSB per RGBA8 channel (the contract of BASELINE.json's north_star).
 * Built and run by tests/test_abi.py::test_compiled_c_program_renders_the_fixture (-m gpu).  Exit code 0 = both frames inside the bar. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "linevis_hip.h"

typedef struct fixture {
    uint32_t magic, version, num_points, num_segments, tf_n, width, height, pad;
    float line_width, fov_y, near_dist, far_dist;
    float view[16], proj[16];
} fixture;

static void* read_block(FILE* f, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, f) != bytes) {
        fprintf(stderr, "abi_smoke: short read (%lu bytes)\n", (unsigned long)bytes);
        exit(2);
    }
    return p;
}

static int max_diff(const uint8_t* a, const uint8_t* b, size_t n) {
    int m = 0;
    size_t i;
    for (i = 0; i < n; i++) {
        const int d = a[i] > b[i] ? a[i] - b[i] : b[i] - a[i];
        if (d > m) m = d;
    }
    return m;
}

#define CK(call)                                                                              \
    do {                                                                                      \
        const int rc_ = (call);                                                               \
        if (rc_ != LV_OK) {                                                                   \
            fprintf(stderr, "abi_smoke: %s -> %d: %s\n", #call, rc_, lv_last_error(ctx));     \
            return 3;                                                                         \
        }                                                                                     \
    } while (0)

int main(int argc, char** argv) {
    FILE* f;
    fixture fx;
    lv_line_point* points;
    uint32_t* seg;
    float *tf, *tf_transparent;
    uint8_t *want_rt, *want_ppll, *got;
    size_t frame_bytes;
    lv_ctx* ctx;
    lv_stats st;
    int err = 0, d_rt, d_ppll, covered = 0;
    size_t i;
    const float white[4] = {1.0f, 1.0f, 1.0f, 1.0f};

    if (argc < 2) {
        fprintf(stderr, "usage: abi_smoke tests/golden/abi_smoke.bin\n");
        return 2;
    }
    f = fopen(argv[1], "rb");
    if (!f || fread(&fx, sizeof fx, 1, f) != 1 || fx.magic != 0x4D53564Cu /* "LVSM" */ || fx.version != 1u) {
        fprintf(stderr, "abi_smoke: cannot read the fixture %s\n", argv[1]);
        return 2;
    }
    points = (lv_line_point*)read_block(f, (size_t)fx.num_points * sizeof(lv_line_point));
    seg = (uint32_t*)read_block(f, (size_t)fx.num_segments * 8u);
    tf = (float*)read_block(f, (size_t)fx.tf_n * 16u);
    tf_transparent = (float*)read_block(f, (size_t)fx.tf_n * 16u);
    frame_bytes = (size_t)fx.width * fx.height * 4u;
    want_rt = (uint8_t*)read_block(f, frame_bytes);
    want_ppll = (uint8_t*)read_block(f, frame_bytes);
    fclose(f);
    got = (uint8_t*)malloc(frame_bytes);

    printf("%s\n", lv_version());
    ctx = lv_create(0, &err);
    if (!ctx) {
        fprintf(stderr, "abi_smoke: lv_create(0) failed with %d (no HIP device? there is no CPU fallback)\n", err);
        return 4;
    }
    CK(lv_set_lines(ctx, points, fx.num_points, seg, fx.num_segments));
    CK(lv_set_transfer_function(ctx, tf, fx.tf_n, 0.0f, 1.0f));
    CK(lv_set_camera(ctx, fx.view, fx.proj, fx.fov_y, fx.near_dist, fx.far_dist, fx.width, fx.height));
    CK(lv_set_background(ctx, white));
    {
        char buf[32];
        sprintf(buf, "%.9g", (double)fx.line_width);
        CK(lv_set_option(ctx, "line_width", buf));
    }
    CK(lv_set_option(ctx, "ambient_occlusion_mode", "RTAO (Screen Space)"));
    CK(lv_set_option(ctx, "ambient_occlusion_strength", "1"));
    CK(lv_set_option(ctx, "ambient_occlusion_iterations", "1"));
    CK(lv_set_option(ctx, "ambient_occlusion_samples_per_frame", "8"));
    CK(lv_set_option(ctx, "depth_cue_strength", "0.8"));
    if (lv_set_option(ctx, "no_such_key", "1") != LV_E_INVALID) {
        fprintf(stderr, "abi_smoke: an unknown option must fail with LV_E_INVALID\n");
        return 5;
    }
    CK(lv_build_accel(ctx));
    CK(lv_render(ctx, LV_RENDERING_MODE_VULKAN_RAY_TRACER, 0, 0, fx.width, fx.height, got));
    d_rt = max_diff(got, want_rt, frame_bytes);
    for (i = 0; i < frame_bytes; i += 4) covered += got[i] != 255 || got[i + 1] != 255 || got[i + 2] != 255;
    CK(lv_get_stats(ctx, &st));
    printf("mode 11: max difference %d LSB, %d covered pixels, %u segments, %u LBVH nodes, %.3f ms\n", d_rt, covered,
           (unsigned)st.num_segments, (unsigned)st.num_nodes, (double)st.ms_total);

    CK(lv_set_transfer_function(ctx, tf_transparent, fx.tf_n, 0.0f, 1.0f));
    CK(lv_set_option(ctx, "ambient_occlusion_mode", "None"));
    CK(lv_set_option(ctx, "depth_cue_strength", "0"));
    CK(lv_render(ctx, LV_RENDERING_MODE_PER_PIXEL_LINKED_LIST, 0, 0, fx.width, fx.height, got));
    d_ppll = max_diff(got, want_ppll, frame_bytes);
    CK(lv_get_stats(ctx, &st));
    printf("mode 2: max difference %d LSB, %lu fragments, longest list %u\n", d_ppll, (unsigned long)st.fragments,
           (unsigned)st.max_depth_complexity);
    lv_destroy(ctx);
    free(points); free(seg); free(tf); free(tf_transparent); free(want_rt); free(want_ppll); free(got);
    if (d_rt > 2 || d_ppll > 2 || covered < 200) {
        fprintf(stderr, "abi_smoke: outside the +-2 LSB contract (or an empty frame)\n");
        return 1;
    }
    printf("abi_smoke ok\n");
    return 0;
}
