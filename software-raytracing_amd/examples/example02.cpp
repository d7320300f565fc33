// 02 - Soft shadows, headless: scene literals of Sources/02_SoftShadows/Main.cpp:201-218, Render(&fb, 90, ...) :221.
#include "../host/02_SoftShadows/Render.hpp"
#include "headless.hpp"

int main(int argc, char* argv[]) {
    try {
        const Options opt = ParseOptions(argc, argv, 1);
        RenderSeed() = opt.seed;
        auto frameBuffer = ImageBuffer<RGBQUAD>(opt.width, opt.height, {0, 0, 0, 0});
        std::cout << "INFO: Frame-buffer initialized  (resolution : " << frameBuffer.getWidth() << "x" << frameBuffer.getHeight() << ", size : " << frameBuffer.getSize() << " bytes)" << std::endl;

        Material redMatte = {math::Vec3<float>(1.0f, 0.0f, 0.0f), 0.0f};
        Material greenMatte = {math::Vec3<float>(0.0f, 1.0f, 0.0f), 0.0f};
        Material whiteMatte = {math::Vec3<float>(1.0f, 1.0f, 1.0f), 0.0f};
        Material redRubber = {math::Vec3<float>(0.6f, 0.2f, 0.2f), 0.1f, 16.0f};

        std::vector<std::shared_ptr<SceneElement>> scene{};
        scene.push_back(std::make_shared<Plane>(whiteMatte, math::Vec3<float>(0.0f, -10.0f, 0.0f), math::Vec3<float>(0.0f, 1.0f, 0.0f)));
        scene.push_back(std::make_shared<Plane>(whiteMatte, math::Vec3<float>(0.0f, 10.0f, 0.0f), math::Vec3<float>(0.0f, -1.0f, 0.0f)));
        scene.push_back(std::make_shared<Plane>(whiteMatte, math::Vec3<float>(0.0f, 0.0f, -20.0f), math::Vec3<float>(0.0f, 0.0f, 1.0f)));
        scene.push_back(std::make_shared<Plane>(whiteMatte, math::Vec3<float>(0.0f, 0.0f, 0.001f), math::Vec3<float>(0.0f, 0.0f, -1.0f)));
        scene.push_back(std::make_shared<Plane>(greenMatte, math::Vec3<float>(10.0f, 0.0f, 0.0f), math::Vec3<float>(-1.0f, 0.0f, 0.0f)));
        scene.push_back(std::make_shared<Plane>(redMatte, math::Vec3<float>(-10.0f, 0.0f, 0.0f), math::Vec3<float>(1.0f, 0.0f, 0.0f)));
        scene.push_back(std::make_shared<Sphere>(redRubber, math::Vec3<float>(0.0f, -6.5f, -15.0f), 3.0f));

        std::vector<LightSource> lightSources{};
        lightSources.push_back({math::Vec3<float>(0.0f, 6.5f, -10.0f), math::Vec3<float>(0.9f, 0.9f, 0.9f), 3.0f});

        if (!opt.dump.empty()) { DumpFlat(whitted_detail::flatten(scene, lightSources), opt.dump); return 0; }

        rtbackend::context();   // device / library initialisation is not part of the render time reported below
        auto renderBeginTime = std::chrono::system_clock::now();
        Render(&frameBuffer, 90.0f, scene, lightSources);
        std::cout << "INFO: Scene rendered in " << std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now() - renderBeginTime).count() << " ms." << std::endl;
        PrintBackendStats();
        PresentOnce(frameBuffer, opt);
        WriteFrame(frameBuffer, opt.out);
    } catch (std::exception& ex) {
        std::cout << ex.what() << std::endl;
        return 1;
    }
    return 0;
}
