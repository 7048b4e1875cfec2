// rpt_render_main.cpp — a headless C++ host in the shape of the reference's main.cpp:14-74 / render():
// scene DSL on stdin -> upload -> per-frame Lorentz refresh + rpt_set_objects + rpt_render -> PPM.
//
//   g++ -O2 -std=c++17 -Iinclude examples/rpt_render_main.cpp -o rpt_render \
//       -Lrelativitypathtracer_amd -lrpt_hip -lrpt_scene -Wl,-rpath,$PWD/relativitypathtracer_amd
//   ./rpt_render 1920 1080 out.ppm [vx vy vz t [frames in_flight]] < assets/reference/Scenes/shadows.txt
//
// With `frames` > 1 the clock runs (16 ms per frame, as the reference's timer does) and the frames are rendered with
// `in_flight` of them overlapping on the GPU: rpt::FrameRing (include/rpt_frames.hpp) — one context per frame slot
// sharing one resident scene (rpt_share_scene), frame f in slot f mod in_flight.  The PPM is the last frame.
//
// Textures are read with the library's built-in binary PPM reader (convert the JPEGs first, e.g. with
// Pillow) — decoding JPEG is the job of CImg/libjpeg in the reference and is outside the render path.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <iterator>
#include <memory>
#include <string>
#include <vector>

#include "rpt.h"
#include "rpt_frames.hpp"
#include "rpt_scene.h"

int main(int argc, char **argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s width height out.ppm [vx vy vz t [frames in_flight]] < scene.txt\n", argv[0]);
        return 2;
    }
    const int width = std::atoi(argv[1]), height = std::atoi(argv[2]);
    const std::string text((std::istreambuf_iterator<char>(std::cin)), std::istreambuf_iterator<char>());

    rpt_scene *scene = rpt_scene_create();                       // inputScene()            main.cpp:31
    rpt_scene_set_asset_root(scene, std::getenv("RPT_ASSETS") ? std::getenv("RPT_ASSETS") : ".");
    if (rpt_scene_input(scene, text.c_str()) != 0) {
        std::fprintf(stderr, "scene: %s\n", rpt_scene_last_error(scene));
        return 1;
    }
    if (argc >= 8) {
        const float v[3] = {(float)std::atof(argv[4]), (float)std::atof(argv[5]), (float)std::atof(argv[6])};
        const float p[4] = {(float)std::atof(argv[7]), 0, 0, 0};
        rpt_scene_set_camera(scene, v, p);
    }

    rpt_ctx *ctx = nullptr;                                      // initOpenCL()            main.cpp:22
    if (rpt_create(&ctx, 0) != RPT_OK) {
        std::fprintf(stderr, "no usable gfx950 device (the render path has no CPU fallback)\n");
        return 1;
    }
    rpt_scene_desc desc;
    rpt_scene_update_objects(scene);                             // Lorentz block of render() Render.cpp:179-200
    rpt_scene_get_desc(scene, &desc);
    float wp[3], ambient;
    int interval;
    rpt_scene_get_params(scene, wp, &ambient, &interval);
    int rc = rpt_upload_scene(ctx, &desc);                       // 8x cl::Buffer + write    main.cpp:33-55
    if (!rc) rc = rpt_set_params(ctx, wp, ambient, width, height, interval);   // initCLKernel()  main.cpp:62
    if (!rc) rc = rpt_set_output(ctx, nullptr);                  // BufferGL(vbo)           main.cpp:58
    if (!rc) rc = rpt_set_objects(ctx, desc.objects, (int)desc.object_count);   //          Render.cpp:202
    if (!rc) rc = rpt_render(ctx);                               // runKernel()             Render.cpp:205
    if (rc) {
        std::fprintf(stderr, "render: %s\n", rpt_last_error(ctx));
        return 1;
    }
    float ms = 0;
    rpt_last_frame_ms(ctx, &ms);
    rpt_ctx *last = ctx;
    const int frames = argc >= 10 ? std::atoi(argv[8]) : 1, in_flight = argc >= 10 ? std::atoi(argv[9]) : 1;
    std::unique_ptr<rpt::FrameRing> ring_owner;                  // include/rpt_frames.hpp: one context per frame slot, ONE resident scene
    if (frames > 1 && in_flight >= 1) {
        ring_owner.reset(new rpt::FrameRing(0, in_flight));
        rpt::FrameRing &ring = *ring_owner;
        rc = ring.status();
        if (!rc) rc = ring.upload(desc);
        if (!rc) rc = ring.set_params(wp, ambient, width, height, interval);
        rpt_scene_set_paused(scene, 0);
        const auto t0 = std::chrono::steady_clock::now();
        int presented = 0;
        // RPT_DUMP_FRAME=k RPT_DUMP_PATH=file.ppm: also write frame k (0-based) as acquire() handed it out
        const char *dump_path = std::getenv("RPT_DUMP_PATH");
        const int dump_frame = std::getenv("RPT_DUMP_FRAME") ? std::atoi(std::getenv("RPT_DUMP_FRAME")) : -1;
        std::vector<unsigned char> dumped;
        for (int f = 0; f < frames && !rc; f++) {
            rpt_scene_advance_time(scene, 16);                   // render(): cameraPos.x += dt  Render.cpp:177
            rpt_scene_update_objects(scene);
            rpt_scene_get_desc(scene, &desc);
            if (void *finished = ring.acquire()) {               // frame f - in_flight, complete: drawGL() goes here,
                presented++;                                     // BEFORE the slot is resubmitted
                if (dump_path && f - ring.frames_in_flight() == dump_frame) {      // test hook: keep that frame's pixels
                    dumped.resize((size_t)width * height * 16);
                    rc = rpt_read_framebuffer(ring.slot(f % ring.frames_in_flight()), dumped.data(), dumped.size());
                    (void)finished;
                }
            }
            if (!rc) rc = ring.enqueue(desc.objects, (int)desc.object_count);
        }
        if (!rc && ring.drain() == nullptr) rc = ring.status() ? ring.status() : 1;
        if (!rc && dump_path && !dumped.empty()) rc = rpt_write_ppm(dump_path, dumped.data(), width, height);
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rc) {
            std::fprintf(stderr, "render: %s\n", ring.last_error());
            return 1;
        }
        last = ring.newest();
        std::fprintf(stderr, "%d frames, %d in flight: %.4f ms/frame, %.0f Mrays/s (%d presented while rendering)\n", frames,
                     ring.frames_in_flight(), sec / frames * 1e3, (double)width * height * frames / sec / 1e6, presented);
    }
    std::vector<unsigned char> fb((size_t)width * height * 16);
    rpt_read_framebuffer(last, fb.data(), fb.size());
    rc = rpt_write_ppm(argv[3], fb.data(), width, height);       // drawGL()                gl_interop.cpp:51
    std::fprintf(stderr, "%dx%d frame in %.3f ms -> %s\n", width, height, ms, argv[3]);
    rpt_destroy(ctx);
    rpt_scene_destroy(scene);
    return rc;
}
