// crt_driver.cpp — the small C++ host program that takes the place of the reference's main()
// (kernel.cu:41-111) for the paths this library covers: parameters (kernel.cu:44-51), curand
// states (:69-71), camera (:74-76), mesh (an OBJ file instead of the FBX SDK's data; :83-90), then
// either renderListAnimation (core/render.h:159-189, the call commented out at kernel.cu:93) or
// renderBVH (kernel.cu:24-39,97) with its timing rows written by writeCSV (kernel.cu:101).
// Everything goes through the C ABI of include/crt.h; no HIP code here except the colour buffer.
//
//   crt_driver --obj mesh.obj [--scale 100] [--width 1024 --height 512] [--spp 4] [--frames 1]
//              [--world list|bvh] [--shader lambert|shade|normal] [--depth 8] [--gpus N]
//              [--out images/moveTest] [--csv output.csv]
#include "../../include/crt.h"

#include <hip/hip_runtime.h>
#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {
double seconds_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
int die(const char* what) {
    fprintf(stderr, "crt_driver: %s: %s\n", what, crt_last_error());
    return 1;
}
}  // namespace

int main(int argc, char** argv) {
    // kernel.cu:44-51
    int nx = 1024, ny = 512, samples = 4, max_depth = 8, beginFrame = 0, n_frames = 1, gpus = 1;
    int world_kind = CRT_WORLD_BVH, shader = CRT_SHADER_LAMBERT;
    float scale = 100.0f;   // Loader/obj_loader.h:164
    std::string obj, out = "images/moveTest", csv = "output.csv";
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { return (i + 1 < argc) ? argv[++i] : ""; };
        if (a == "--obj") obj = next();
        else if (a == "--scale") scale = (float)atof(next());
        else if (a == "--width") nx = atoi(next());
        else if (a == "--height") ny = atoi(next());
        else if (a == "--spp") samples = atoi(next());
        else if (a == "--depth") max_depth = atoi(next());
        else if (a == "--frames") n_frames = atoi(next());
        else if (a == "--gpus") gpus = atoi(next());
        else if (a == "--out") out = next();
        else if (a == "--csv") csv = next();
        else if (a == "--world") world_kind = strcmp(next(), "list") == 0 ? CRT_WORLD_LIST : CRT_WORLD_BVH;
        else if (a == "--shader") { const char* s = next(); shader = !strcmp(s, "shade") ? CRT_SHADER_SHADE : (!strcmp(s, "normal") ? CRT_SHADER_NORMAL : CRT_SHADER_LAMBERT); }
        else { fprintf(stderr, "crt_driver: unknown option %s\n", a.c_str()); return 2; }
    }
    if (obj.empty()) { fprintf(stderr, "crt_driver: --obj is required\n"); return 2; }
    if (crt_device_count() < 1) { fprintf(stderr, "crt_driver: no HIP device (there is no CPU fallback)\n"); return 1; }
    mkdir("images", 0777);
    mkdir(out.c_str(), 0777);

    const float albedo[3] = {0.65f, 0.05f, 0.05f};                       // createScene.h:180
    crt_camera camera;                                                   // init_camera, createScene.h:160
    const float lookfrom[3] = {0, 100, 1000}, lookat[3] = {0, 150, 0}, vup[3] = {0, 1, 0};
    crt_camera_make(&camera, lookfrom, lookat, vup, 40.0f, (float)nx / (float)ny, 0.0f, 10.0f);

    std::vector<float> colorBuffer((size_t)nx * ny * 3);
    std::vector<int> frames;
    std::vector<double> rendering, update;
    double build = 0.0;

    if (gpus > 1) {   // one process, several devices: strips dealt round-robin, one grouped RCCL send/recv per frame
        int np = 0, nt = 0;
        if (crt_parse_obj(obj.c_str(), nullptr, nullptr, &np, &nt) != CRT_OK) return die("crt_parse_obj");
        std::vector<float> pts((size_t)np * 3), idx((size_t)nt * 3);
        if (crt_parse_obj(obj.c_str(), pts.data(), idx.data(), &np, &nt) != CRT_OK) return die("crt_parse_obj");
        for (float& p : pts) p *= scale;
        crt_multi* m = crt_multi_create(pts.data(), np, idx.data(), nt, nullptr, 1, albedo, nx, ny, 0, gpus);
        if (!m) return die("crt_multi_create");
        for (int frame = beginFrame; frame < beginFrame + n_frames; ++frame) {
            const auto t0 = std::chrono::steady_clock::now();
            if (crt_multi_render(m, &camera, samples, max_depth, frame, colorBuffer.data()) != CRT_OK) return die("crt_multi_render");
            frames.push_back(frame); rendering.push_back(seconds_since(t0)); update.push_back(0.0);
            const std::string path = out + "/picture_" + std::to_string(frame) + ".png";     // core/render.h:148-152
            if (crt_write_png(path.c_str(), nx, ny, colorBuffer.data()) != CRT_OK) return die("crt_write_png");
        }
        crt_multi_free(m);
    } else {
        crt_scene* world = crt_scene_from_obj(obj.c_str(), scale, albedo);
        if (!world) return die("crt_scene_from_obj");
        crt_rng* curand_state = crt_rng_create(0, (int64_t)nx * ny, 0);                      // kernel.cu:69-71
        if (!curand_state) return die("crt_rng_create");
        if (world_kind == CRT_WORLD_BVH) {                                                    // renderBVH, kernel.cu:24-39
            const auto t0 = std::chrono::steady_clock::now();
            if (crt_scene_build_bvh(world, curand_state, 0) != CRT_OK) return die("crt_scene_build_bvh");
            build = seconds_since(t0);
        }
        float* d_colorBuffer = nullptr;
        if (hipMalloc(&d_colorBuffer, colorBuffer.size() * sizeof(float)) != hipSuccess) { fprintf(stderr, "crt_driver: hipMalloc failed\n"); return 1; }
        // the frame's host buffer (the reference's malloc'ed colorBuffer, render.h:165): page-locked, so the copy back runs at the
        // link's rate right behind the frame's kernels and the frame costs ONE synchronisation
        float* h_colorBuffer = nullptr;
        if (hipHostMalloc(&h_colorBuffer, colorBuffer.size() * sizeof(float), hipHostMallocDefault) != hipSuccess) { fprintf(stderr, "crt_driver: hipHostMalloc failed\n"); return 1; }
        for (int frame = beginFrame; frame < beginFrame + n_frames; ++frame) {                // core/render.h:174 / :207
            auto t0 = std::chrono::steady_clock::now();
            if (world_kind == CRT_WORLD_BVH) {
                if (crt_scene_refit_bvh(world, nullptr) != CRT_OK) return die("crt_scene_refit_bvh");   // Update_BVH, render.h:215
                (void)hipDeviceSynchronize();
            }
            update.push_back(seconds_since(t0));
            t0 = std::chrono::steady_clock::now();
            if (crt_render_rows_ex(world, &camera, curand_state, nx, ny, samples, max_depth, frame, 0, ny, world_kind, shader,
                                   d_colorBuffer, nullptr) != CRT_OK) return die("crt_render_rows_ex");
            if (hipMemcpyAsync(h_colorBuffer, d_colorBuffer, colorBuffer.size() * sizeof(float), hipMemcpyDeviceToHost, nullptr) != hipSuccess ||
                hipStreamSynchronize(nullptr) != hipSuccess) { fprintf(stderr, "crt_driver: copying the frame back failed\n"); return 1; }   // render.h:180-183
            rendering.push_back(seconds_since(t0));
            frames.push_back(frame);
            const std::string path = out + "/picture_" + std::to_string(frame) + ".png";
            if (crt_write_png(path.c_str(), nx, ny, h_colorBuffer) != CRT_OK) return die("crt_write_png");
            printf("frame %d: %d triangles, %dx%dx%d, render %.3f ms, update %.3f ms -> %s\n", frame, crt_scene_triangle_count(world), nx, ny,
                   samples, rendering.back() * 1e3, update.back() * 1e3, path.c_str());
        }
        (void)hipFree(d_colorBuffer);
        (void)hipHostFree(h_colorBuffer);
        crt_rng_free(curand_state);
        crt_scene_free(world);
    }
    if (crt_write_timing_csv(csv.c_str(), (int)frames.size(), frames.data(), rendering.data(), update.data(), build) != CRT_OK)   // kernel.cu:101
        return die("crt_write_timing_csv");
    return 0;
}
