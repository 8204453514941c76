// headless.cpp — see headless.hpp.  Host-only C++17 above hip_engine.hpp / scene_io.hpp.
#include "headless.hpp"

#include "image_io.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <filesystem>
#include <fstream>
#include <sstream>

#include "hip_engine.hpp"
#include "mini_json.hpp"
#include "scene_io.hpp"

namespace RayZath::Hip::Headless {

namespace {
std::string parent_dir(const std::string& path) {
    const size_t p = path.find_last_of("/\\");
    return p == std::string::npos ? std::string() : path.substr(0, p + 1);
}
std::string file_name(const std::string& path) {
    const size_t p = path.find_last_of("/\\");
    return p == std::string::npos ? path : path.substr(p + 1);
}
std::string fixed3(double v) {
    char b[64];
    std::snprintf(b, sizeof b, "%.3f", v);
    return b;
}
using clock_t_ = std::chrono::steady_clock;
float seconds_since(clock_t_::time_point t0) { return std::chrono::duration<float>(clock_t_::now() - t0).count(); }
}  // namespace

std::string scientificWithPrefix(size_t value) {
    // the decimal digits of `value`, at least four of them ("0000" padding behind short numbers)
    std::string digits = std::to_string(value);
    const size_t n = digits.size();
    digits.resize(std::max<size_t>(n, 4), '0');
    static const char prefixes[] = "####KKKMMMGGGTTTPPPEEE";
    const size_t before_point = (n - 1) % 3 + 1;
    std::string out = digits.substr(0, before_point) + "." + digits.substr(before_point, 4 - before_point);
    if (value >= 1000) out.push_back(prefixes[n]);
    return out;
}

std::vector<RenderTask> prepareTasks(const std::string& task_file) {
    std::ifstream file(task_file, std::ios::binary);
    if (!file.is_open()) throw Exception(HIPRZ_ERR_INVALID, "Failed to read file: " + task_file + ": Failed to open the file.");
    std::stringstream text;
    text << file.rdbuf();
    try {
        const IO::Json json = IO::parseJson(text.str());
        const IO::Json* tasks_json = json.find("tasks");
        if (!tasks_json) throw std::runtime_error("File must contain \"tasks\" key.");
        auto create = [&](const IO::Json& e) {
            if (!e.is_object()) throw std::runtime_error("Benchmark entry must be an object.");
            const IO::Json* scene = e.find("scene path");
            if (!scene) throw std::runtime_error("Benchmark entry must contain a scene path key.");
            if (!scene->is_string()) throw std::runtime_error("scene path key must be a string");
            RenderTask t;
            t.scene_path = scene->str;
            const bool absolute = !t.scene_path.empty() && (t.scene_path[0] == '/' || (t.scene_path.size() > 1 && t.scene_path[1] == ':'));
            if (!absolute) t.scene_path = parent_dir(task_file) + t.scene_path;
            if (const IO::Json* engine = e.find("engine")) {
                auto add = [&](const IO::Json& name) {
                    if (!name.is_string()) throw std::runtime_error("Specified engine must be a string.");
                    if (name.str != "CPU" && name.str != "CUDAGPU" && name.str != "HIPGPU") throw std::runtime_error("Unknown engine type \"" + name.str + "\"");
                    t.engines.push_back(name.str);
                };
                if (engine->is_string()) add(*engine);
                else if (engine->is_array())
                    for (const auto& n : engine->items) add(n);
                else throw std::runtime_error("Engine value must be either a string or an array.");
            } else {
                t.engines = {"HIPGPU"};  // the reference defaults to its GPU engine (headless.cpp:121-124)
            }
            if (const IO::Json* v = e.find("rpp"); v && v->is_number()) t.rpp = unsigned(v->num);
            if (const IO::Json* v = e.find("timeout"); v && v->is_number()) t.timeout = float(v->num);
            if (const IO::Json* v = e.find("max depth"); v && v->is_number()) t.max_depth = unsigned(std::min(std::max(v->num, 1.0), 255.0));
            return t;
        };
        std::vector<RenderTask> tasks;
        if (tasks_json->is_object()) tasks.push_back(create(*tasks_json));
        else if (tasks_json->is_array())
            for (const auto& e : tasks_json->items) tasks.push_back(create(e));
        else throw std::runtime_error("tasks's value have to be either an array or an object.");
        return tasks;
    } catch (const std::runtime_error& e) {
        throw Exception(HIPRZ_ERR_INVALID, "Failed to read file: " + task_file + ": " + e.what());
    }
}

std::vector<TaskResult> executeTask(const RenderTask& task, const std::string& report_dir, bool save_images, const std::vector<int>& devices, bool quiet, bool sample_sharding) {
    World world;
    {
        if (!quiet) std::printf("Loading \"%s\"\n", file_name(task.scene_path).c_str());
        const auto t0 = clock_t_::now();
        IO::LoadLog log;
        IO::loadScene(task.scene_path, world, log);
        if (!quiet) std::printf("%sLoaded in: %ss\n\n", log.str().c_str(), fixed3(seconds_since(t0)).c_str());
    }
    std::vector<TaskResult> results;
    for (const std::string& engine_name : task.engines) {
        if (engine_name != "HIPGPU") {
            std::printf("Engine %s is not part of this host library: skipped.\n", engine_name.c_str());
            continue;
        }
        // several ids: one context over those GPUs (tiles interleaved, gathered inside the readback); one id: the plain engine, which
        // picks its stream count by the world like every other host (Hip::Engine::defaultStreams)
        std::unique_ptr<Engine> engine_owner = devices.size() == 1 ? std::make_unique<Engine>(devices[0]) : std::make_unique<Engine>(devices);
        Engine& engine = *engine_owner;
        // the runner converges a frame and reports rays per second: several GPUs each render whole frames on their own seed streams
        // (their accumulators summed at the readback) — a device's step stays a whole-frame step whatever their number
        if (devices.size() > 1 && sample_sharding) engine.shardMode(Engine::ShardMode::Samples);
        const size_t rays_per_pass_scale = devices.size() > 1 && sample_sharding ? devices.size() : 1u;
        RenderConfig config;
        config.tracing.max_depth = uint8_t(task.max_depth);
        config.tracing.rpp = 1;
        TaskResult result;
        result.scene_path = task.scene_path, result.engine = engine_name, result.max_depth = task.max_depth;

        // Headless::render (headless.cpp:277-296): one pipelined renderWorld, then steer the passes per call so that a call
        // takes `load_time` (the square root damps the correction, the running mean damps it again)
        const float load_time = 0.1f;
        float floaty_rpp = 1.0f;
        auto render = [&]() {
            const auto t0 = clock_t_::now();
            engine.renderWorld(world, config, true, false);
            const float duration = seconds_since(t0);
            if (std::fabs((duration - load_time) / load_time) > 0.05f) {
                const float new_rpp = floaty_rpp * std::pow(load_time / duration, 0.5f);
                floaty_rpp = (floaty_rpp + new_rpp) * 0.5f;
                config.tracing.rpp = std::min(std::max<uint32_t>(uint32_t(floaty_rpp), 1u), 1024u);
            }
        };
        unsigned traced = 0;
        if (task.rpp - traced < config.tracing.rpp) config.tracing.rpp = task.rpp - traced;
        render();  // warm-up (headless.cpp:203)
        const auto start = clock_t_::now();
        auto last_stop = start;
        for (traced = 0; traced < task.rpp;) {
            if (task.rpp - traced < config.tracing.rpp) config.tracing.rpp = task.rpp - traced;
            const uint32_t this_call = config.tracing.rpp;
            render();
            const auto stop = clock_t_::now();
            const float task_duration = std::chrono::duration<float>(stop - start).count();
            const float pass_duration = std::chrono::duration<float>(stop - last_stop).count();
            last_stop = stop;
            traced += this_call;
            // the pipelined call enqueued `this_call` passes of W*H rays each (cpu_engine_renderer.cpp:173)
            const size_t diff = size_t(this_call) * world.camera.width * world.camera.height * rays_per_pass_scale;
            result.total_traced_rays += diff;
            if (!quiet)
                std::printf("\rRendering... %u/%u +%u [rpp] (%.2f%%) | %s rps | %.3fs (timeout: %.3fs)   ", traced, task.rpp, config.tracing.rpp,
                            traced / float(task.rpp) * 100.0f, scientificWithPrefix(size_t(diff / pass_duration)).c_str(), task_duration, task.timeout);
            if (task_duration >= task.timeout) break;
        }
        // the calls above are pipelined (sync = false): one more pass with sync = true puts the final frame into the camera
        // buffers, and the clock stops when it is there
        config.tracing.rpp = 1;
        engine.renderWorld(world, config, true, true);
        result.total_traced_rays += size_t(world.camera.width) * world.camera.height * rays_per_pass_scale;
        result.duration = seconds_since(start);
        if (!quiet) std::printf("\nRendered in: %ss\n\n", fixed3(result.duration).c_str());
        if (save_images) {
            // saveMap<Texture> (headless.cpp:255-275, saver.cpp:16-37): the camera's RGBA8 frame as a PNG
            const std::string name = report_dir + file_name(task.scene_path) + "_camera_" + scientificWithPrefix(result.total_traced_rays) + "_" + engine_name + ".png";
            std::string why;
            if (!IO::writePNG(name, world.camera.image_buffer.data(), world.camera.width, world.camera.height, 4, why)) throw Exception(HIPRZ_ERR_INVALID, why);
            if (!quiet) std::printf("Saved %s\n", name.c_str());
        }
        if (!quiet) std::printf("%s\n", engine.timingsString().c_str());
        results.push_back(result);
    }
    return results;
}

std::string reportText(const std::vector<TaskResult>& results) {
    std::string out;
    for (const auto& r : results) {
        out += "Scene: " + file_name(r.scene_path) + "\n";
        out += "\tengine: " + r.engine + " | max depth: " + std::to_string(r.max_depth) + "\n";
        out += "\tduration: " + fixed3(r.duration) + "s | traced " + scientificWithPrefix(r.total_traced_rays) + " rays (" +
               scientificWithPrefix(r.duration > 0 ? size_t(r.total_traced_rays / r.duration) : 0) + " rps)\n";
    }
    return out;
}

int run(const std::string& task_file, std::string report_dir, bool save_images, const std::vector<int>& devices, bool quiet, bool sample_sharding) {
    try {
        if (report_dir.empty()) report_dir = parent_dir(task_file);
        if (!report_dir.empty() && report_dir.back() != '/') report_dir.push_back('/');
        if (!report_dir.empty()) {  // the reference creates benchmark_<date>_<time>/ inside an existing directory (headless.cpp:35-41); here the
            std::error_code ec;     // directory named on the command line is the report directory itself and is created when missing
            std::filesystem::create_directories(report_dir, ec);
            if (ec) throw Exception(HIPRZ_ERR_INVALID, "cannot create the report directory " + report_dir);
        }
        if (!quiet) std::printf("Reading config file: \"%s\"\n", task_file.c_str());
        const auto tasks = prepareTasks(task_file);
        std::vector<TaskResult> results;
        for (const auto& task : tasks) {
            auto r = executeTask(task, report_dir, save_images, devices, quiet, sample_sharding);
            results.insert(results.end(), r.begin(), r.end());
        }
        const std::string path = report_dir + "report.txt";
        if (!quiet) std::printf("Generating report in \"%s\"\n", path.c_str());
        std::ofstream report(path);
        if (!report.is_open()) throw Exception(HIPRZ_ERR_INVALID, "cannot write " + path);
        report << reportText(results);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
}

}  // namespace RayZath::Hip::Headless
