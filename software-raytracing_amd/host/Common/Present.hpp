// Present.hpp - the present step of the reference's main(): PresentFrame(pixels, width, height, hWnd) copies the B,G,R,x
// quads of the frame buffer to the window (Sources/01_Basic/Main.cpp:272-310, the same function in 02-04).  A headless host
// has no window; what it has is a sink for finished frames that a viewer watches.  PresentFrame keeps its argument order
// with the sink in place of the window handle.  Two sinks:
//   BmpFileSink    - rewrites one 32-bit top-down BMP (its pixel format IS the frame buffer's B,G,R,x quads, so the pixels
//                    are written as they are) through a temporary file + rename, so a viewer that reloads the file never
//                    sees a torn frame;
//   RawStreamSink  - appends raw frames to a file, FIFO or stdout ("-"), e.g. for
//                    ffplay -f rawvideo -pixel_format bgra -video_size 784x561 <fifo>.
// InteractiveRender drives the library's progressive frame (rt_progressive_*): the running sums stay in HBM, every pass
// adds samples and presents a preview exposed for the samples done so far - only 4 bytes per pixel cross PCIe per
// presented frame.  It is what an interactive main() calls in place of Render + PresentFrame.
#pragma once
#include <cstdio>
#include <cstdint>
#include <stdexcept>
#include <string>
#include "RtBackend.hpp"

class FrameSink {
public:
    virtual ~FrameSink() = default;
    virtual void present(const void* pixels, int width, int height) = 0;
    unsigned framesPresented() const { return frames_; }
protected:
    unsigned frames_ = 0;
};

class BmpFileSink : public FrameSink {
    std::string path_;
    static void put32(unsigned char* p, uint32_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; p[2] = (v >> 16) & 255; p[3] = (v >> 24) & 255; }
public:
    explicit BmpFileSink(std::string path) : path_(std::move(path)) {}
    void present(const void* pixels, int width, int height) override {
        if (!pixels || width <= 0 || height <= 0) throw std::runtime_error("ERROR: BmpFileSink: empty frame");
        const uint32_t bytes = static_cast<uint32_t>(width) * static_cast<uint32_t>(height) * 4u;
        unsigned char h[54] = {'B', 'M'};
        put32(h + 2, 54u + bytes); put32(h + 10, 54u);                     // file size, pixel offset
        put32(h + 14, 40u); put32(h + 18, static_cast<uint32_t>(width));    // BITMAPINFOHEADER
        put32(h + 22, static_cast<uint32_t>(-height));                      // negative height: top row first, as the frame buffer
        h[26] = 1; h[28] = 32;                                              // planes, bits per pixel (BI_RGB: B,G,R,x)
        put32(h + 34, bytes); put32(h + 38, 2835u); put32(h + 42, 2835u);
        const std::string tmp = path_ + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f) throw std::runtime_error("ERROR: can't open " + tmp);
        const bool ok = std::fwrite(h, 1, sizeof h, f) == sizeof h && std::fwrite(pixels, 1, bytes, f) == bytes;
        if (std::fclose(f) != 0 || !ok || std::rename(tmp.c_str(), path_.c_str()) != 0) throw std::runtime_error("ERROR: can't write " + path_);
        ++frames_;
    }
};

class RawStreamSink : public FrameSink {
    FILE* f_ = nullptr; bool own_ = false;
public:
    explicit RawStreamSink(const std::string& path) {
        if (path == "-") f_ = stdout;
        else { f_ = std::fopen(path.c_str(), "wb"); own_ = true; }
        if (!f_) throw std::runtime_error("ERROR: can't open " + path);
    }
    ~RawStreamSink() override { if (own_ && f_) std::fclose(f_); }
    RawStreamSink(const RawStreamSink&) = delete; RawStreamSink& operator=(const RawStreamSink&) = delete;
    void present(const void* pixels, int width, int height) override {
        const size_t bytes = static_cast<size_t>(width) * static_cast<size_t>(height) * 4u;
        if (std::fwrite(pixels, 1, bytes, f_) != bytes || std::fflush(f_) != 0) throw std::runtime_error("ERROR: frame stream closed");
        ++frames_;
    }
};

// Sources/01_Basic/Main.cpp:272 - same arguments, the sink where the window handle was
inline void PresentFrame(void* pixels, int width, int height, FrameSink& sink) { sink.present(pixels, width, height); }

// The interactive counterpart of Render(03/04) + PresentFrame: samples are added 'chunk' per pass until 'samples' are done
// (or keepGoing() says stop), every pass ends with a preview in *fb and a PresentFrame to the sink.  Returns the samples
// per pixel accumulated; fb holds the last preview - after the last pass the finished frame.
template <class KeepGoing>
inline unsigned InteractiveRender(ImageBuffer<RGBQUAD>* fb, const rt_scene& scene, const rt_camera& cam, int variant, unsigned samples,
                                  unsigned chunk, uint32_t seed, FrameSink& sink, KeepGoing keepGoing) {
    rt_ctx* ctx = rtbackend::context();
    rt_path_params prm;
    rtbackend::check(rt_path_params_default(variant, &prm));
    prm.spp = static_cast<int32_t>(samples); prm.seed = seed;
    rtbackend::check(rt_progressive_begin(ctx, &scene, &cam, &prm, static_cast<int32_t>(fb->getWidth()), static_cast<int32_t>(fb->getHeight())));
    const rt_image img = rtbackend::image_of(fb);
    int32_t done = 0;
    try {
        while (static_cast<unsigned>(done) < samples && keepGoing(static_cast<unsigned>(done))) {
            rtbackend::check(rt_progressive_pass(ctx, static_cast<int32_t>(chunk ? chunk : 1), &img, &done));
            PresentFrame(fb->getData(), static_cast<int>(fb->getWidth()), static_cast<int>(fb->getHeight()), sink);
        }
    } catch (...) { rt_progressive_end(ctx); throw; }
    rtbackend::check(rt_progressive_end(ctx));
    return static_cast<unsigned>(done);
}
