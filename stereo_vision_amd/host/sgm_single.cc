// sgm_single -- counterpart of reconstruction/main/sgm_single/main.cc:11-79.
//   sgm_single left right out_folder P1 P2 [--disp-range 256] [--window-sz 3] [--paths 8] [--device 0]
// Same positional arguments and output files: out_folder/disparities/<stem>.png (16-bit, 256 * d)
// and out_folder/norm_hist/<stem>.png (8-bit preview).  disp_range = 256 and window_sz = 3 are
// hard-coded in the reference (:29-32) and are options here.  All SGM work runs in libsgmx.so.
#include "cli_common.hpp"
#include "host_util.hpp"
#include "png_io.hpp"

int main(int argc, char **argv)
{
    return sgmx::cli::guarded("sgm_single", [&]() {
        const sgmx::cli::Args a = sgmx::cli::parse(argc, argv);
        if (a.pos.size() != 5) {
            std::cerr << "usage:\n" << argv[0] << " left right out_folder P1 P2" << std::endl;
            return 1;
        }
        const std::string left = a.pos[0], right = a.pos[1], out = a.pos[2];
        recon::StereoSGMParams p;
        p.disp_range = a.num("disp-range", 256);
        p.window_sz = a.num("window-sz", 3);
        p.penalty1 = sgmx::cli::to_int(a.pos[3], "P1");
        p.penalty2 = sgmx::cli::to_int(a.pos[4], "P2");
        sgmx::Mat img_left = sgmx::png::read_gray8(left), img_right = sgmx::png::read_gray8(right), img_disp;
        recon::StereoSGM sgm(p, a.num("device", 0), a.num("paths", 8));
        sgm.compute(img_left, img_right, img_disp);
        const sgmx::Mat img8 = sgmx::host::convert_to_u8(img_disp, 1.0 / 256.0);
        const std::string prefix = sgmx::host::stem(left);
        sgmx::png::write_gray(out + "/disparities/" + prefix + ".png", img_disp);
        sgmx::png::write_gray(out + "/norm_hist/" + prefix + ".png", sgmx::host::normalize_minmax_u8(img8));
        return 0;
    });
}
