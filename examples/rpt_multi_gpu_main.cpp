// rpt_multi_gpu_main.cpp — the multi-GPU frame path of INTEGRATION.md §4 as a native C++ host: one process drives
// every GPU of the node, the frame is sharded by interleaved 8-row tiles (tile k -> GPU k mod N), each GPU renders
// its tiles into a 4 B/pixel colour plane, drops the constant alpha byte (3 B/pixel on the wire), ONE ncclGather per
// frame brings the planes to GPU 0 and rpt_scatter_colour_plane3_on expands them into the reference's 16 B/pixel
// framebuffer.  Three frames are in
// flight: per GPU three contexts on one resident scene (rpt_share_scene), each with its own stream, and everything
// a frame slot does — refresh, render, gather, reassembly — is ordered by that one stream.
//
//   hipcc -O2 -std=c++17 -Iinclude examples/rpt_multi_gpu_main.cpp -o rpt_multi_gpu \
//       -Lrelativitypathtracer_amd -lrpt_hip -lrpt_scene -lrccl -Wl,-rpath,$PWD/relativitypathtracer_amd
//   ./rpt_multi_gpu 3840 2160 out.ppm frames [n_gpus [root_run]] < assets/reference/Scenes/shadows.txt
//
// root_run (a power of two) selects the WEIGHTED split: per period of root_run + N - 1 tiles GPU 0 renders root_run
// tiles straight into the frame's framebuffer and GPU j the single tile root_run + j - 1 into its plane — pixels
// rendered on GPU 0 cross no link (rpt_set_tile_pattern, rpt_scatter_helper_planes3_on).
//
// (bench.py does the same with one process per GPU over torch.distributed; this file is the drop-in shape for the
// reference's single-process C++ host.)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <iterator>
#include <string>
#include <vector>

#include "rpt.h"
#include "rpt_scene.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        const int rc_ = (int)(call);                                                 \
        if (rc_ != 0) {                                                              \
            std::fprintf(stderr, "%s failed with %d (line %d)\n", #call, rc_, __LINE__); \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: %s width height out.ppm frames [n_gpus [root_run]] < scene.txt\n", argv[0]);
        return 2;
    }
    const int width = std::atoi(argv[1]), height = std::atoi(argv[2]), frames = std::atoi(argv[4]);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        std::fprintf(stderr, "no usable gfx950 device (the render path has no CPU fallback)\n");
        return 1;
    }
    if (argc >= 6 && std::atoi(argv[5]) >= 1 && std::atoi(argv[5]) < n) n = std::atoi(argv[5]);
    const int kSlots = 3;
    const int root_run = argc >= 7 ? std::atoi(argv[6]) : 0;      // 0: equal split, tile k -> GPU k mod N
    const int period = root_run ? root_run + n - 1 : n;

    const std::string text((std::istreambuf_iterator<char>(std::cin)), std::istreambuf_iterator<char>());
    rpt_scene *scene = rpt_scene_create();
    rpt_scene_set_asset_root(scene, std::getenv("RPT_ASSETS") ? std::getenv("RPT_ASSETS") : ".");
    if (rpt_scene_input(scene, text.c_str()) != 0) {
        std::fprintf(stderr, "scene: %s\n", rpt_scene_last_error(scene));
        return 1;
    }
    if (std::getenv("RPT_T0")) {
        const float v[3] = {0, 0, 0}, p[4] = {(float)std::atof(std::getenv("RPT_T0")), 0, 0, 0};
        rpt_scene_set_camera(scene, v, p);
    }
    rpt_scene_set_paused(scene, 0);
    rpt_scene_update_objects(scene);
    rpt_scene_desc desc;
    rpt_scene_get_desc(scene, &desc);
    float wp[3], ambient;
    int interval;
    rpt_scene_get_params(scene, wp, &ambient, &interval);

    // per GPU: three frame slots = three contexts on one resident scene, each with a stream and a colour plane
    const size_t tiles = (size_t)(height + RPT_TILE_ROWS - 1) / RPT_TILE_ROWS;
    const size_t words = ((tiles + period - 1) / period) * RPT_TILE_ROWS * (size_t)width;   // padded: every GPU sends the same count
    std::vector<std::vector<rpt_ctx *>> ctx(n, std::vector<rpt_ctx *>(kSlots, nullptr));
    std::vector<std::vector<hipStream_t>> stream(n, std::vector<hipStream_t>(kSlots));
    std::vector<std::vector<void *>> plane(n, std::vector<void *>(kSlots, nullptr)), plane3 = plane;
    std::vector<int> devs(n);
    for (int d = 0; d < n; d++) {
        devs[d] = d;
        CHECK(hipSetDevice(d));
        const std::vector<int> slot_devices(kSlots, d);                 // the frame slots of GPU d: one context each
        CHECK(rpt_create_multi(ctx[d].data(), slot_devices.data(), kSlots));
        for (int k = 0; k < kSlots; k++) {
            CHECK(k == 0 ? rpt_upload_scene(ctx[d][0], &desc) : rpt_share_scene(ctx[d][k], ctx[d][0]));
            CHECK(rpt_set_params(ctx[d][k], wp, ambient, width, height, interval));
            if (!root_run) CHECK(rpt_set_rows(ctx[d][k], d, n, /*colour_plane=*/1));
            else if (d == 0) CHECK(rpt_set_tile_pattern(ctx[d][k], 0, period, root_run, /*colour_plane=*/0));
            else CHECK(rpt_set_tile_pattern(ctx[d][k], root_run + d - 1, period, 1, /*colour_plane=*/1));
            CHECK(hipStreamCreateWithFlags(&stream[d][k], hipStreamNonBlocking));
            CHECK(rpt_set_stream(ctx[d][k], stream[d][k]));
            CHECK(hipMalloc(&plane[d][k], words * 4));
            CHECK(hipMemset(plane[d][k], 0, words * 4));
            if (!(root_run && d == 0)) CHECK(rpt_set_plane_output(ctx[d][k], plane[d][k]));
            CHECK(hipMalloc(&plane3[d][k], words * 3));
            CHECK(hipMemset(plane3[d][k], 0, words * 3));
        }
    }
    std::vector<ncclComm_t> comm(n);
    CHECK(ncclCommInitAll(comm.data(), n, devs.data()));
    CHECK(hipSetDevice(0));
    std::vector<void *> gathered(kSlots, nullptr);
    for (int k = 0; k < kSlots; k++) CHECK(hipMalloc(&gathered[k], (size_t)n * words * 3));
    std::vector<void *> framebuffer(kSlots, nullptr);             // one per frame slot: a frame is complete in ITS buffer
    for (int k = 0; k < kSlots; k++) {
        CHECK(hipMalloc(&framebuffer[k], (size_t)width * height * 16));
        if (root_run) CHECK(rpt_set_output(ctx[0][k], framebuffer[k]));     // GPU 0's own tiles are rendered in place
    }

    const auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < frames; f++) {
        const int k = f % kSlots;
        rpt_scene_advance_time(scene, 16);                       // render(): cameraPos.x += dt    Render.cpp:177
        rpt_scene_update_objects(scene);                         // Lorentz block of render()      Render.cpp:179-200
        rpt_scene_get_desc(scene, &desc);
        for (int d = 0; d < n; d++) {
            CHECK(rpt_set_objects(ctx[d][k], desc.objects, (int)desc.object_count));      //      Render.cpp:202
            CHECK(rpt_render_async(ctx[d][k]));                                          // runKernel()
            if (!(root_run && d == 0)) CHECK(rpt_pack_colour_plane3_on(ctx[d][k], stream[d][k], plane[d][k], plane3[d][k], words));
        }
        CHECK(ncclGroupStart());                                 // the frame's one exchange step
        for (int d = 0; d < n; d++)
            CHECK(ncclGather(plane3[d][k], gathered[k], words * 3, ncclUint8, 0, comm[d], stream[d][k]));
        CHECK(ncclGroupEnd());
        if (root_run) CHECK(rpt_scatter_helper_planes3_on(ctx[0][k], stream[0][k], gathered[k], framebuffer[k], width, height, n, root_run, words * 3));
        else CHECK(rpt_scatter_colour_plane3_on(ctx[0][k], stream[0][k], gathered[k], framebuffer[k], width, height, n, words * 3));
    }
    for (int d = 0; d < n; d++)
        for (int k = 0; k < kSlots; k++) CHECK(rpt_sync(ctx[d][k]));
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "%d GPU(s), %d frames of %dx%d, %d in flight: %.4f ms/frame, %.0f Mrays/s\n", n, frames, width, height, kSlots,
                 sec / frames * 1e3, (double)width * height * frames / sec / 1e6);

    std::vector<unsigned char> fb((size_t)width * height * 16);
    CHECK(hipSetDevice(0));
    CHECK(hipMemcpy(fb.data(), framebuffer[(frames - 1) % kSlots], fb.size(), hipMemcpyDeviceToHost));
    const int rc = rpt_write_ppm(argv[3], fb.data(), width, height);                     // drawGL()   gl_interop.cpp:51
    for (int d = 0; d < n; d++) {
        ncclCommDestroy(comm[d]);
        for (int k = 0; k < kSlots; k++) rpt_destroy(ctx[d][k]);
    }
    rpt_scene_destroy(scene);
    return rc;
}
