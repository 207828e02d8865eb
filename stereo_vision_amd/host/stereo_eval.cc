// stereo_eval -- counterpart of reconstruction/main/stereo_eval/main.cc:14-60: the reference's only
// timing harness for the path.  One pair (Gaussian-blurred, :36-38), D = 128, window 5, the grid
// P1 in {1,3,5,7,10,15,20} x P2 in {20,...,250} (:23-24); prints "My SGM time spent = %fs" around
// compute (:46-52) and writes disp_P1_<p1>_P2_<p2>.png (8-bit, equalised) per grid point.
//   stereo_eval left right [out_dir] [--disp-range 128] [--window-sz 5] [--device 0] [--no-write]
// (the reference hard-codes the two image paths and writes into the working directory).
#include <sys/time.h>

#include "cli_common.hpp"
#include "host_util.hpp"
#include "png_io.hpp"

int main(int argc, char **argv)
{
    return sgmx::cli::guarded("stereo_eval", [&]() {
        const sgmx::cli::Args a = sgmx::cli::parse(argc, argv, {"no-write"});
        if (a.pos.size() < 2 || a.pos.size() > 3) {
            std::cerr << "usage:\n" << argv[0] << " left right [out_dir]" << std::endl;
            return 1;
        }
        const std::vector<int> P1 = {1, 3, 5, 7, 10, 15, 20};
        const std::vector<int> P2 = {20, 30, 35, 40, 50, 60, 70, 80, 100, 150, 250};
        const std::string out = a.pos.size() == 3 ? a.pos[2] + "/" : std::string();
        if (!out.empty())
            sgmx::host::mkdirs(out);
        recon::StereoSGMParams p;
        p.disp_range = a.num("disp-range", 128);
        p.window_sz = a.num("window-sz", 5);
        sgmx::Mat img_left = sgmx::png::read_gray8(a.pos[0]), img_right = sgmx::png::read_gray8(a.pos[1]);
        bool blurred = false;
        for (size_t i = 0; i < P1.size(); i++)
            for (size_t j = 0; j < P2.size(); j++) {
                p.penalty1 = P1[i];
                p.penalty2 = P2[j];
                recon::StereoSGM sgm(p, a.num("device", 0));
                if (!blurred) {
                    sgm.gaussian_blur3(img_left, 0.7);
                    sgm.gaussian_blur3(img_right, 0.7);
                    blurred = true;
                }
                sgmx::Mat mat_disp;
                struct timeval start, end;
                gettimeofday(&start, NULL);
                sgm.compute(img_left, img_right, mat_disp);
                gettimeofday(&end, NULL);
                const float delta = ((end.tv_sec - start.tv_sec) * 1000000u + end.tv_usec - start.tv_usec) / 1.e6;
                std::printf("\nMy SGM time spent = %fs\n\n", delta);
                if (a.has("no-write"))
                    continue;
                char name[64];
                std::snprintf(name, sizeof(name), "disp_P1_%05d_P2_%05d.png", P1[i], P2[j]);
                sgmx::png::write_gray(out + name, sgmx::host::equalize_hist(sgmx::host::convert_to_u8(mat_disp, 1.0)));
            }
        return 0;
    });
}
