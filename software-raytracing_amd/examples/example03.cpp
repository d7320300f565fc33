// 03 - Path tracing basics, headless: scene literals of Sources/03_PathTracingBasics/Main.cpp:172-185,
// Render(&fb, scene, 50, MULTISAMPLING_LEVEL = 16, {-2,1.5,1}, {-30,-42,0}) :189.
#include "../host/03_PathTracingBasics/Render.hpp"
#include "headless.hpp"

#define MULTISAMPLING_LEVEL 16

int main(int argc, char* argv[]) {
    try {
        const Options opt = ParseOptions(argc, argv, MULTISAMPLING_LEVEL);
        RenderSeed() = opt.seed;
        auto frameBuffer = ImageBuffer<RGBQUAD>(opt.width, opt.height, {0, 0, 0, 0});
        std::cout << "INFO: Frame-buffer initialized  (resolution : " << frameBuffer.getWidth() << "x" << frameBuffer.getHeight() << ", size : " << frameBuffer.getSize() << " bytes)" << std::endl;

        auto materialGround = std::make_shared<MaterialDiffuse>(math::Vec3<float>(0.8f, 0.8f, 0.0f));
        auto materialCenter = std::make_shared<MaterialDiffuse>(math::Vec3<float>(0.1f, 0.2f, 0.5f));
        auto materialLeft = std::make_shared<MaterialMetal>(math::Vec3<float>(0.8f, 0.8f, 0.8f), 0.3f);
        auto materialRight = std::make_shared<MaterialMetal>(math::Vec3<float>(0.8f, 0.6f, 0.2f), 0.4f);

        Scene scene{};
        scene.addElement(std::make_shared<Sphere>(math::Vec3<float>(0.0f, -100.5f, -1.0f), 100.0f, materialGround));
        scene.addElement(std::make_shared<Sphere>(math::Vec3<float>(0.0f, 0.0f, -1.0f), 0.5f, materialCenter));
        scene.addElement(std::make_shared<Sphere>(math::Vec3<float>(-1.0f, 0.0f, -1.0f), 0.5f, materialLeft));
        scene.addElement(std::make_shared<Sphere>(math::Vec3<float>(1.0f, 0.0f, -1.0f), 0.5f, materialRight));

        if (!opt.dump.empty()) { rtbackend::Flat flat; path_detail::flatten(scene.getElements(), &flat); DumpFlat(flat, opt.dump); return 0; }

        rtbackend::context();   // device / library initialisation is not part of the render time reported below
        if (opt.bvh) rtbackend::check(rt_set_option(rtbackend::context(), RT_OPT_ACCEL, RT_ACCEL_BVH));
        if (opt.reference_arithmetic) rtbackend::check(rt_set_option(rtbackend::context(), RT_OPT_ARITHMETIC, RT_ARITH_REFERENCE));
        auto renderBeginTime = std::chrono::system_clock::now();
        if (opt.chunk && !opt.present.empty()) {
            // interactive: --chunk samples per pass, every pass presented to --present (the sums stay on the GPU)
            rtbackend::Flat flat; path_detail::flatten(scene.getElements(), &flat);
            auto sink = MakeSink(opt.present);
            const unsigned done = InteractiveRender(&frameBuffer, flat.scene(), rt_camera{50.0f, {-2.0f, 1.5f, 1.0f}, {-30.0f, -42.0f, 0.0f}}, 3, opt.samples, opt.chunk, RenderSeed(),
                                                    *sink, [&](unsigned d) { return !(opt.stopAfter && d >= opt.stopAfter); });
            std::cout << "INFO: " << done << " of " << opt.samples << " samples per pixel done, " << sink->framesPresented() << " frames presented" << std::endl;
        } else {
        Render(&frameBuffer, scene, 50.0f, opt.samples, {-2.0f, 1.5f, 1.0f}, {-30.0f, -42.0f, 0.0f});
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
