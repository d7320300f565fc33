// 04 - Path tracing with lights, headless: scene literals of Sources/04_PathTracingLights/Main.cpp:181-199,
// Render(&fb, scene, 90, SAMPLES_PER_PIXEL = 32, {0,0,10}, {0,0,0}) :203.
#include "../host/04_PathTracingLights/Render.hpp"
#include "../host/Common/Progressive.hpp"
#include "headless.hpp"

#define SAMPLES_PER_PIXEL 32

int main(int argc, char* argv[]) {
    try {
        const Options opt = ParseOptions(argc, argv, SAMPLES_PER_PIXEL);
        RenderSeed() = opt.seed;
        auto frameBuffer = ImageBuffer<RGBQUAD>(opt.width, opt.height, {0, 0, 0, 0});
        std::cout << "INFO: Frame-buffer initialized  (resolution : " << frameBuffer.getWidth() << "x" << frameBuffer.getHeight() << ", size : " << frameBuffer.getSize() << " bytes)" << std::endl;

        auto red = std::make_shared<materials::Diffuse>(math::Vec3<float>(0.65f, 0.05f, 0.05f));
        auto white = std::make_shared<materials::Diffuse>(math::Vec3<float>(0.73f, 0.73f, 0.73f));
        auto green = std::make_shared<materials::Diffuse>(math::Vec3<float>(0.12f, 0.45f, 0.15f));
        auto light = std::make_shared<materials::Light>(math::Vec3<float>(15.0f, 15.0f, 15.0f));
        auto metal = std::make_shared<materials::Metal>(math::Vec3<float>(0.8f, 0.8f, 0.8f), 0.2f);
        auto ball = std::make_shared<materials::Diffuse>(math::Vec3<float>(0.1f, 0.2f, 0.5f));
        auto glass = std::make_shared<materials::Refractive>(0.6f);

        scene::List scene{};
        scene.addElement(std::make_shared<scene::Plane>(white, math::Vec3<float>(0.0f, 5.0f, 0.0f), math::Vec3<float>(0.0f, -1.0f, 0.0f)));
        scene.addElement(std::make_shared<scene::Plane>(white, math::Vec3<float>(0.0f, -5.0f, 0.0f), math::Vec3<float>(0.0f, 1.0f, 0.0f)));
        scene.addElement(std::make_shared<scene::Plane>(white, math::Vec3<float>(0.0f, 0.0f, -5.0f), math::Vec3<float>(0.0f, 0.0f, 1.0f)));
        scene.addElement(std::make_shared<scene::Plane>(red, math::Vec3<float>(-5.0f, 0.0f, 0.0f), math::Vec3<float>(1.0f, 0.0f, 0.0f)));
        scene.addElement(std::make_shared<scene::Plane>(green, math::Vec3<float>(5.0f, 0.0f, 0.0f), math::Vec3<float>(-1.0f, 0.0f, 0.0f)));
        scene.addElement(std::make_shared<scene::Sphere>(metal, math::Vec3<float>(0.0f, -3.0f, -1.0f), 2.0f));
        scene.addElement(std::make_shared<scene::Sphere>(ball, math::Vec3<float>(-2.0f, -4.0f, 2.5f), 1.0f));
        scene.addElement(std::make_shared<scene::Sphere>(glass, math::Vec3<float>(2.5f, -3.5f, 3.0f), 1.5f));
        scene.addElement(std::make_shared<scene::Rectangle>(light, math::Vec3<float>(0.0f, 4.95f, 0.0f), math::Vec2<float>(3.0f, 3.0f), math::Vec3<float>(90.0f, 0.0f, 0.0f)));

        if (!opt.dump.empty()) { rtbackend::Flat flat; path_detail::flatten(scene.getElements(), &flat); DumpFlat(flat, opt.dump); return 0; }

        rtbackend::context();   // device / library initialisation is not part of the render time reported below
        if (opt.bvh) rtbackend::check(rt_set_option(rtbackend::context(), RT_OPT_ACCEL, RT_ACCEL_BVH));
        if (opt.reference_arithmetic) rtbackend::check(rt_set_option(rtbackend::context(), RT_OPT_ARITHMETIC, RT_ARITH_REFERENCE));
        auto renderBeginTime = std::chrono::system_clock::now();
        if (opt.chunk && !opt.present.empty()) {
            // interactive: --chunk samples per pass, every pass presented to --present (the sums stay on the GPU)
            rtbackend::Flat flat; path_detail::flatten(scene.getElements(), &flat);
            auto sink = MakeSink(opt.present);
            const unsigned done = InteractiveRender(&frameBuffer, flat.scene(), rt_camera{90.0f, {0.0f, 0.0f, 10.0f}, {0.0f, 0.0f, 0.0f}}, 4, opt.samples, opt.chunk, RenderSeed(),
                                                    *sink, [&](unsigned d) { return !(opt.stopAfter && d >= opt.stopAfter); });
            std::cout << "INFO: " << done << " of " << opt.samples << " samples per pixel done, " << sink->framesPresented() << " frames presented" << std::endl;
        } else if (opt.chunk) {
            // progressive: --chunk samples per pass, the running sums kept in --checkpoint so that an interrupted run
            // (--stop-after N simulates one) continues where it stopped; the finished frame equals the one-shot frame
            rtbackend::Flat flat; path_detail::flatten(scene.getElements(), &flat);
            const rt_scene sc = flat.scene();
            const rt_camera cam{90.0f, {0.0f, 0.0f, 10.0f}, {0.0f, 0.0f, 0.0f}};
            ProgressiveFrame frame(opt.width, opt.height, opt.samples, RenderSeed(), 4);
            if (!opt.checkpoint.empty() && frame.load(opt.checkpoint)) std::cout << "INFO: Resumed at sample " << frame.samplesDone() << std::endl;
            while (!frame.complete() && !(opt.stopAfter && frame.samplesDone() >= opt.stopAfter)) {
                frame.accumulate(sc, cam, opt.chunk);
                if (!opt.checkpoint.empty()) frame.save(opt.checkpoint);
            }
            std::cout << "INFO: " << frame.samplesDone() << " of " << opt.samples << " samples per pixel done" << std::endl;
            frame.resolve(&frameBuffer);
        } else {
        Render(&frameBuffer, scene, 90.0f, opt.samples, {0.0f, 0.0f, 10.0f}, {0.0f, 0.0f, 0.0f});
        PresentOnce(frameBuffer, opt);
        }
        std::cout << "INFO: Scene rendered in " << std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now() - renderBeginTime).count() << " ms." << std::endl;
        PrintBackendStats();
        WriteFrame(frameBuffer, opt.out);
    } catch (std::exception& ex) {
        std::cout << ex.what() << std::endl;
        return 1;
    }
    return 0;
}
