// Progressive.hpp - progressive / checkpointed accumulation for the path-traced examples (the reference
// renders its samples in one go and has no counterpart).  Every pixel-sample is a pure function of (seed, pixel, sample
// index), so a frame can be rendered in sample ranges, in any order, in any number of processes, and the float sums
// added: rt_render_path with [sample_begin, sample_end) returns sum(range)/spp per pixel, which is what is kept here.
// A checkpoint is the running sums plus the ranges already done; resuming continues with the samples still missing.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "RtBackend.hpp"

class ProgressiveFrame {
    struct Header { char magic[8]; uint32_t width, height, spp, done, seed, variant; uint64_t scene_hash; };
    uint32_t width_, height_, spp_, done_ = 0, seed_, variant_;
    uint64_t scene_hash_ = 0;   // FNV-1a of the scene records and the camera the sums belong to (0 = nothing accumulated yet)
    static uint64_t fnv(uint64_t h, const void* p, size_t n) {
        const unsigned char* b = static_cast<const unsigned char*>(p);
        for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ULL; }
        return h;
    }
    // The sums of one frame belong to one scene and one camera: mixing in samples of another is refused.
    void bind(const rt_scene& scene, const rt_camera& cam) {
        uint64_t h = 1469598103934665603ULL;
        h = fnv(h, scene.prims, static_cast<size_t>(scene.n_prims) * sizeof(rt_prim));
        h = fnv(h, scene.materials, static_cast<size_t>(scene.n_materials) * sizeof(rt_material));
        h = fnv(h, &cam, sizeof cam);
        if (h == 0) h = 1;
        if (scene_hash_ != 0 && scene_hash_ != h) throw std::runtime_error("ERROR: ProgressiveFrame holds samples of another scene or camera");
        scene_hash_ = h;
    }
    std::vector<float> mean_;   // width*height*3: sum over the samples done so far / spp
public:
    ProgressiveFrame(unsigned width, unsigned height, unsigned spp, uint32_t seed, int variant)
        : width_(width), height_(height), spp_(spp), seed_(seed), variant_(static_cast<uint32_t>(variant)), mean_(static_cast<size_t>(width) * height * 3, 0.0f) {}
    unsigned samplesDone() const { return done_; }
    bool complete() const { return done_ >= spp_; }

    // Renders the next 'count' samples of every pixel and adds them.
    void accumulate(const rt_scene& scene, const rt_camera& cam, unsigned count) {
        if (complete() || count == 0) return;      // an empty range is not (0, 0): rt_path_params reads that as "the whole frame"
        bind(scene, cam);
        rt_path_params prm;
        rtbackend::check(rt_path_params_default(static_cast<int32_t>(variant_), &prm));
        prm.spp = static_cast<int32_t>(spp_); prm.seed = seed_;
        prm.sample_begin = static_cast<int32_t>(done_);
        prm.sample_end = static_cast<int32_t>(done_ + count < spp_ ? done_ + count : spp_);
        std::vector<float> part(mean_.size());
        rt_image img{nullptr, part.data(), static_cast<int32_t>(width_), static_cast<int32_t>(height_), 0};
        rtbackend::check(rt_render_path(rtbackend::context(), &scene, &cam, &prm, &img));
        for (size_t i = 0; i < mean_.size(); ++i) mean_[i] += part[i];
        done_ = static_cast<uint32_t>(prm.sample_end);
    }
    // The tail of the reference's Render (03 Main.cpp:352-367) on what has been accumulated so far, scaled to the samples
    // done (a preview is correctly exposed): sqrt gamma, Clamp, truncation, B,G,R,255.
    void resolve(ImageBuffer<RGBQUAD>* fb) const {
        const float scale = done_ ? static_cast<float>(spp_) / static_cast<float>(done_) : 0.0f;
        RGBQUAD* px = fb->getData();
        auto q = [&](float v) { v = std::sqrt(v * scale); v = v < 0.0f ? 0.0f : (1.0f < v ? 1.0f : v); return static_cast<uint8_t>(v * 255.0f); };
        for (size_t i = 0; i < static_cast<size_t>(width_) * height_; ++i)
            px[i] = RGBQUAD{q(mean_[3 * i + 2]), q(mean_[3 * i + 1]), q(mean_[3 * i]), 255};
    }
    void save(const std::string& path) const {
        FILE* f = std::fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("ERROR: can't open " + path);
        Header h{{'R', 'T', 'P', 'R', 'O', 'G', '2', 0}, width_, height_, spp_, done_, seed_, variant_, scene_hash_};
        std::fwrite(&h, sizeof h, 1, f); std::fwrite(mean_.data(), sizeof(float), mean_.size(), f);
        std::fclose(f);
    }
    // false when there is no checkpoint; throws when the checkpoint belongs to another frame (size, spp, seed, variant)
    bool load(const std::string& path) {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) return false;
        Header h{};
        bool ok = std::fread(&h, sizeof h, 1, f) == 1 && std::memcmp(h.magic, "RTPROG2", 8) == 0 && h.width == width_ && h.height == height_ &&
                  h.spp == spp_ && h.seed == seed_ && h.variant == variant_ && h.done <= spp_ &&
                  std::fread(mean_.data(), sizeof(float), mean_.size(), f) == mean_.size();
        std::fclose(f);
        if (!ok) throw std::runtime_error("ERROR: " + path + " is not a checkpoint of this frame");
        done_ = h.done; scene_hash_ = h.scene_hash;     // the next accumulate() checks its scene and camera against it
        return true;
    }
};
