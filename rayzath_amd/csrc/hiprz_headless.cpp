// hiprz_headless — command line of the headless runner (Application/main.cpp:41-77):
//   hiprz_headless --headless <tasks.json> [report_dir] [-r] [--device N | --devices N,M,...] [--shard-mode samples|tiles] [--quiet]
//     several devices divide a frame by samples (default: whole frames per device on its own seed stream, accumulators summed at the
//     readback — a device's step stays a whole-frame step, rays per second scale with the devices) or by interleaved tiles (the one-device
//     frame bit for bit)
//   hiprz_headless --format <integer>        prints scientificWithPrefix(integer) (used by the tests)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "headless.hpp"

int main(int argc, char** argv) {
    std::string task_file, report_dir;
    bool save_images = false, quiet = false, headless = false, sample_sharding = true;
    std::vector<int> devices{0};
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-h" || a == "--help") {
            std::printf("usage: %s --headless <task_path> [report_path] [-r|--render] [--device N | --devices N,M,...] [--shard-mode samples|tiles] [--quiet]\n", argv[0]);
            return 0;
        } else if (a == "--format" && i + 1 < argc) {
            std::printf("%s\n", RayZath::Hip::Headless::scientificWithPrefix(std::strtoull(argv[++i], nullptr, 10)).c_str());
            return 0;
        } else if (a == "--headless") {
            headless = true;
            if (i + 1 < argc && argv[i + 1][0] != '-') task_file = argv[++i];
            if (i + 1 < argc && argv[i + 1][0] != '-') report_dir = argv[++i];
        } else if (a == "-r" || a == "--render") {
            save_images = true;
        } else if (a == "--device" && i + 1 < argc) {
            devices.assign(1, std::atoi(argv[++i]));
        } else if (a == "--devices" && i + 1 < argc) {  // one context over several GPUs of the node
            devices.clear();
            for (const char* p = argv[++i]; *p;) {
                devices.push_back(int(std::strtol(p, const_cast<char**>(&p), 10)));
                if (*p == ',') ++p;
            }
            if (devices.empty()) devices.assign(1, 0);
        } else if (a == "--shard-mode" && i + 1 < argc) {
            const std::string m = argv[++i];
            if (m != "samples" && m != "tiles") {
                std::fprintf(stderr, "--shard-mode samples|tiles\n");
                return 2;
            }
            sample_sharding = m == "samples";
        } else if (a == "--quiet") {
            quiet = true;
        } else {
            std::fprintf(stderr, "unknown argument %s\n", a.c_str());
            return 2;
        }
    }
    if (!headless || task_file.empty()) {
        std::fprintf(stderr, "usage: %s --headless <task_path> [report_path] [-r] (this host side has no UI)\n", argv[0]);
        return 2;
    }
    return RayZath::Hip::Headless::run(task_file, report_dir, save_images, devices, quiet, sample_sharding);
}
