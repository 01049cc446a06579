// rt_amd_render: the reference's command line flow (Source/main.cpp:25-98) on the GPU path:
// parse the scene file, render it, write the PNG named by the scene's `output` command.
#include <cstdio>
#include <vector>

#include "rt_amd.h"

int main(int argc, char** argv) {
    if (argc < 2) {
        std::puts("Specify an input file.");
        return -1;
    }
    rt_file_info* info = nullptr;
    if (rt_parse_input_file(argv[1], &info) != 0) {
        std::printf("Failed to parse input file. %s\n", rt_last_error());
        return -1;
    }
    if (rt_init(0) != 0) {
        std::printf("No usable GPU: %s\n", rt_last_error());
        return -2;
    }
    const uint32_t w = rt_file_info_width(info), h = rt_file_info_height(info);
    std::vector<uint8_t> rgb(static_cast<size_t>(w) * h * 3);
    rt_stats stats;
    if (rt_render_file_info(info, rgb.data(), &stats) != 0) {
        std::printf("Render failed: %s\n", rt_last_error());
        return -3;
    }
    if (rt_write_png(rt_file_info_output_filename(info), w, h, rgb.data()) != 0) {
        std::printf("Could not write image: %s\n", rt_last_error());
        return -4;
    }
    std::printf("total time: %gms (device), %llu rays, %llu shadow rays\n", stats.device_ms,
                static_cast<unsigned long long>(stats.closest_rays), static_cast<unsigned long long>(stats.shadow_rays));
    rt_file_info_destroy(info);
    return 0;
}
