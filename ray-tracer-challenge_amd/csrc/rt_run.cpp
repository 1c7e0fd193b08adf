// rt_run -- the reference's CLI (rt_run/src/main.rs:10-44) on top of include/rtc.h:
//   rt_run <scene.yaml> [--width W --height H] [--device D] [--gpus N] [--depth N]
// reads the YAML scene, renders it on one MI355X (or, with --gpus N, on devices D .. D+N-1 of this node: interleaved
// 8-row tiles, canvas assembled on device D) and writes "<stem>.ppm" into the current directory, where
// <stem> is the text before the first '.' of the path, after its last '/' (main.rs:32-39).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtc.h"

static int die(const char *what) {
    fprintf(stderr, "rt_run: %s: %s\n", what, rtc_last_error());
    return 1;
}

int main(int argc, char **argv) {
    const char *path = nullptr;
    uint32_t width = 0, height = 0, depth = 5;
    int device = 0, gpus = 1;
    for (int i = 1; i < argc; i++) {
        auto next = [&](const char *flag) -> const char * {
            if (i + 1 >= argc) {
                fprintf(stderr, "rt_run: %s needs a value\n", flag);
                exit(2);
            }
            return argv[++i];
        };
        if (!strcmp(argv[i], "--width")) width = (uint32_t)atoi(next("--width"));
        else if (!strcmp(argv[i], "--height")) height = (uint32_t)atoi(next("--height"));
        else if (!strcmp(argv[i], "--device")) device = atoi(next("--device"));
        else if (!strcmp(argv[i], "--depth")) depth = (uint32_t)atoi(next("--depth"));
        else if (!strcmp(argv[i], "--gpus")) gpus = atoi(next("--gpus"));
        else if (!strcmp(argv[i], "--strict-fp")) {}  // (the only arithmetic there is: the reference's, uncontracted)
        else if (!path) path = argv[i];
    }
    if (!path) {
        fprintf(stderr, "Must provide a path to a YAML file with a scene description.\n");
        return 2;
    }
    rtc_scene *scene = nullptr;
    if (rtc_scene_load_yaml_file(path, width, height, &scene) != RTC_OK) return die("Should be able to parse scene YAML");
    if (gpus < 1 || gpus > 64) {
        fprintf(stderr, "rt_run: --gpus must be 1..64\n");
        return 2;
    }
    std::vector<int> devices;
    for (int k = 0; k < gpus; k++) devices.push_back(device + k);
    if (rtc_scene_commit_devices(scene, devices.data(), (uint32_t)devices.size()) != RTC_OK)
        return die("Should be able to render the scene");
    uint32_t hsize = 0, vsize = 0;
    rtc_scene_get_camera(scene, &hsize, &vsize, nullptr, nullptr);
    std::vector<double> rgb((size_t)hsize * vsize * 3);
    rtc_render_opts opts;
    rtc_render_opts_default(&opts);
    opts.depth = depth;
    rtc_stats st;
    const auto t0 = std::chrono::steady_clock::now();
    if (rtc_render(scene, &opts, rgb.data(), &st) != RTC_OK) return die("Should be able to render the scene");
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    rtc_scene_destroy(scene);

    size_t need = 0;
    rtc_canvas_to_ppm(rgb.data(), hsize, vsize, nullptr, 0, &need);
    std::string ppm(need, '\0');
    if (rtc_canvas_to_ppm(rgb.data(), hsize, vsize, &ppm[0], ppm.size(), nullptr) != RTC_OK) return die("to_ppm");

    std::string stem(path);
    stem = stem.substr(0, stem.find('.'));
    const size_t slash = stem.find_last_of('/');
    if (slash != std::string::npos) stem = stem.substr(slash + 1);
    const std::string out = stem + ".ppm";
    FILE *f = fopen(out.c_str(), "wb");
    if (!f || fwrite(ppm.data(), 1, ppm.size(), f) != ppm.size()) {
        fprintf(stderr, "rt_run: cannot write %s\n", out.c_str());
        return 1;
    }
    fclose(f);
    printf("Wrote output to %s\n", out.c_str());
    fprintf(stderr, "rt_run: %ux%u on %d GPU(s), kernel %.3f ms, render call %.1f ms\n", hsize, vsize, gpus,
            st.kernel_ms, ms);
    return 0;
}
