// exp/decbench.cc -- PNG decode / encode scaling over host threads (no GPU): decbench file.png iters
#include "../stereo_vision_amd/host/png_io.hpp"
#include <chrono>
#include <cstdio>
#include <thread>
int main(int argc, char **argv)
{
    if (argc < 3) return 1;
    const int iters = std::atoi(argv[2]);
    using clk = std::chrono::steady_clock;
    sgmx::Mat img = sgmx::png::read_gray8(argv[1]);
    sgmx::Mat d16(img.rows, img.cols, sgmx::MAT_16U);
    for (int y = 0; y < img.rows; y++)
        for (int x = 0; x < img.cols; x++)
            d16.at<uint16_t>(y, x) = (uint16_t)(img.at<uint8_t>(y, x) * 97);
    for (int mode = 0; mode < 2; mode++)
        for (int nt : {1, 2, 4, 8, 16, 32}) {
            auto t0 = clk::now();
            std::vector<std::thread> th;
            for (int t = 0; t < nt; t++)
                th.emplace_back([&, t]() {
                    for (int i = 0; i < iters; i++) {
                        if (mode == 0) {
                            sgmx::Mat m = sgmx::png::read_gray8(argv[1]);
                            if (m.data[5] == 255 && m.data[6] == 254) std::puts("");
                        } else {
                            sgmx::png::write_gray("/tmp/decbench_" + std::to_string(t) + ".png", d16);
                        }
                    }
                });
            for (auto &x : th) x.join();
            const double dt = std::chrono::duration<double>(clk::now() - t0).count();
            std::printf("%s threads %2d: %.2f ms per image per thread, %.0f images/s\n", mode ? "write16" : "read8", nt, dt / iters * 1e3, nt * iters / dt);
        }
    return 0;
}
