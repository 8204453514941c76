// headless.hpp — the benchmark runner of the HIPGPU host side, after RayZath's own harness
// (Application/headless.{hpp,cpp}, CLI Application/main.cpp:41-77): a task file names scenes, an rpp budget and a timeout;
// every task loads its scene, renders through Engine::renderWorld until the budget or the timeout is reached, adapting the
// passes per call to a target call time, and a `report.txt` lists rays and rays per second per task.
#pragma once

#include <cstddef>
#include <string>
#include <vector>

namespace RayZath::Hip::Headless {

// RayZath::Utils::scientificWithPrefix (RayZath/text_utils.h:10-38; pinned by Tests/text_utils.cpp): four significant digits
// with a K/M/G/T/P/E prefix, e.g. 0 -> "0.000", 999 -> "999.0", 1001 -> "1.001K", 10000000 -> "10.00M".
std::string scientificWithPrefix(size_t value);

struct RenderTask {  // headless.hpp:10-17
    std::string scene_path;
    unsigned rpp = 1000;
    float timeout = 60.0f;
    std::vector<std::string> engines;  // names as in Engine::engine_name + "HIPGPU"
    unsigned max_depth = 16;
};
struct TaskResult {  // headless.hpp:18-33
    std::string scene_path, engine;
    float duration = 0.0f;
    size_t total_traced_rays = 0;
    unsigned max_depth = 16;
};

// Headless::prepareTasks (headless.cpp:56-160): {"tasks": {...} | [{"scene path", "engine": name | [names], "rpp", "timeout"}]};
// relative scene paths are relative to the task file.  "max depth" is an extension (the reference renders at 16).
std::vector<RenderTask> prepareTasks(const std::string& task_file);
// Headless::executeTask (headless.cpp:163-276) for the engines this host side has ("HIPGPU"; others are reported and skipped)
std::vector<TaskResult> executeTask(const RenderTask& task, const std::string& report_dir, bool save_images, const std::vector<int>& devices, bool quiet, bool sample_sharding = true);
// Headless::generateReport (headless.cpp:297-330): the same three lines per result
std::string reportText(const std::vector<TaskResult>& results);
// Headless::run (headless.cpp:17-55)
// devices: one context over all of them; sample_sharding (several devices): Engine::ShardMode::Samples — whole frames per device on its own
// seed stream, summed — instead of interleaved tiles (`--shard-mode tiles`)
int run(const std::string& task_file, std::string report_dir, bool save_images, const std::vector<int>& devices, bool quiet, bool sample_sharding = true);

}  // namespace RayZath::Hip::Headless
