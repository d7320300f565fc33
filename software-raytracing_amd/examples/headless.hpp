// headless.hpp - what replaces the Win32 shell of the reference examples: command-line size instead of the window's
// client rectangle (784x561 is the client area of the reference's 800x600 window) and a binary PPM file instead of
// PresentFrame (Sources/01_Basic/Main.cpp:272-310).
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include "../host/Common/Present.hpp"

struct Options {
    unsigned width = 784, height = 561, samples = 0, seed = 1;
    std::string out, dump, checkpoint;
    std::string present;                 // frame sink in place of the window: file.bmp (rewritten per frame), a raw B,G,R,x stream file / FIFO, or - (stdout)
    unsigned chunk = 0, stopAfter = 0;   // progressive rendering: samples per pass, stop early (to be resumed)
    bool bvh = false;                    // 03/04: query the scene through the optional BVH (same image, fewer primitive tests)
    bool reference_arithmetic = false;   // 03/04: RT_ARITH_REFERENCE - rays and spheres in the reference's own floating-point operation sequence (slower)
};

inline Options ParseOptions(int argc, char* argv[], unsigned defaultSamples) {
    Options o;
    o.samples = defaultSamples;
    for (int i = 1; i < argc; ++i) {
        auto value = [&](const char* name) -> const char* {
            if (std::strcmp(argv[i], name) != 0) return nullptr;
            if (i + 1 >= argc) throw std::runtime_error(std::string("ERROR: missing value after ") + name);
            return argv[++i];
        };
        if (std::strcmp(argv[i], "--bvh") == 0) o.bvh = true;
        else if (std::strcmp(argv[i], "--reference-arithmetic") == 0) o.reference_arithmetic = true;
        else if (const char* v = value("--gpus")) {          // 03/04: a device group over GPUs 0..N-1 (rt_create_multi)
            std::vector<int32_t> ids; for (int k = 0, n = std::atoi(v); k < n; ++k) ids.push_back(k);
            rtbackend::use_devices(ids);
        } else if (const char* v = value("--devices")) {     // explicit list, e.g. 0,1 (an id may repeat: one-GPU rehearsal)
            std::vector<int32_t> ids;
            for (const char* p = v; *p;) { char* end = nullptr; long id = std::strtol(p, &end, 10); if (end == p) break; ids.push_back(static_cast<int32_t>(id)); p = (*end == ',') ? end + 1 : end; }
            rtbackend::use_devices(ids);
        }
        else if (const char* v = value("--width")) o.width = static_cast<unsigned>(std::atoi(v));
        else if (const char* v = value("--height")) o.height = static_cast<unsigned>(std::atoi(v));
        else if (const char* v = value("--samples")) o.samples = static_cast<unsigned>(std::atoi(v));
        else if (const char* v = value("--seed")) o.seed = static_cast<unsigned>(std::strtoul(v, nullptr, 10));
        else if (const char* v = value("--out")) o.out = v;
        else if (const char* v = value("--present")) o.present = v;
        else if (const char* v = value("--dump-scene")) o.dump = v;
        else if (const char* v = value("--checkpoint")) o.checkpoint = v;
        else if (const char* v = value("--chunk")) o.chunk = static_cast<unsigned>(std::atoi(v));
        else if (const char* v = value("--stop-after")) o.stopAfter = static_cast<unsigned>(std::atoi(v));
        else throw std::runtime_error(std::string("ERROR: unknown option ") + argv[i] + " (use --width --height --samples --seed --out --dump-scene --chunk --checkpoint --stop-after --bvh --reference-arithmetic --gpus --devices --present)");
    }
    return o;
}

// The sink --present names: *.bmp -> BmpFileSink, anything else -> RawStreamSink
inline std::unique_ptr<FrameSink> MakeSink(const std::string& path) {
    if (path.size() > 4 && path.compare(path.size() - 4, 4, ".bmp") == 0) return std::unique_ptr<FrameSink>(new BmpFileSink(path));
    return std::unique_ptr<FrameSink>(new RawStreamSink(path));
}
// What the reference's main() does after Render (01 Main.cpp:214-215): show the finished frame once.
template <class FrameBuffer> inline void PresentOnce(FrameBuffer& fb, const Options& o) {
    if (o.present.empty()) return;
    auto sink = MakeSink(o.present);
    PresentFrame(fb.getData(), static_cast<int>(fb.getWidth()), static_cast<int>(fb.getHeight()), *sink);
}

// P6 PPM, top row first, R,G,B from the B,G,R,reserved quads
template <class FrameBuffer> inline void WriteFrame(FrameBuffer& fb, const std::string& path) {
    if (path.empty()) return;
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("ERROR: can't open " + path);
    std::fprintf(f, "P6\n%u %u\n255\n", fb.getWidth(), fb.getHeight());
    std::string row(static_cast<size_t>(fb.getWidth()) * 3, '\0');
    for (unsigned y = 0; y < fb.getHeight(); ++y) {
        const auto* px = fb[static_cast<int>(y)];
        for (unsigned x = 0; x < fb.getWidth(); ++x) {
            row[3 * x + 0] = static_cast<char>(px[x].rgbRed); row[3 * x + 1] = static_cast<char>(px[x].rgbGreen); row[3 * x + 2] = static_cast<char>(px[x].rgbBlue);
        }
        std::fwrite(row.data(), 1, row.size(), f);
    }
    std::fclose(f);
}

// Raw flat records (counts, then rt_prim[], rt_material[], rt_light[]) for tests/test_host_mirror.py; needs no GPU.
inline void DumpFlat(const rtbackend::Flat& flat, const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("ERROR: can't open " + path);
    const int32_t n[3] = {static_cast<int32_t>(flat.prims.size()), static_cast<int32_t>(flat.mats.size()), static_cast<int32_t>(flat.lights.size())};
    std::fwrite(n, sizeof n, 1, f);
    std::fwrite(flat.prims.data(), sizeof(rt_prim), flat.prims.size(), f);
    std::fwrite(flat.mats.data(), sizeof(rt_material), flat.mats.size(), f);
    std::fwrite(flat.lights.data(), sizeof(rt_light), flat.lights.size(), f);
    std::fclose(f);
}

inline void PrintBackendStats() {
    rt_stats st{};
    if (rt_get_stats(rtbackend::context(), &st) == RT_OK)
        std::cout << "INFO: rays " << st.rays << ", shadow rays " << st.shadow_rays << ", samples " << st.samples
                  << ", kernel " << st.kernel_ms << " ms, call " << st.total_ms << " ms" << (st.allreduce_ms > 0 ? ", all-reduce " + std::to_string(st.allreduce_ms) + " ms" : std::string()) << std::endl;
}
