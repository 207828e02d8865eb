// sgm_cnn -- counterpart of reconstruction/main/sgm_cnn/main.cc:11-74.
//   sgm_cnn left.bin right.bin out_folder prefix P1 P2 [--disp-range 255] [--paths 8] [--device 0]
// left/right are descriptor files in the format of sgm.h:378-410; outputs as sgm_single.
#include "cli_common.hpp"
#include "host_util.hpp"
#include "png_io.hpp"

int main(int argc, char **argv)
{
    return sgmx::cli::guarded("sgm_cnn", [&]() {
        const sgmx::cli::Args a = sgmx::cli::parse(argc, argv);
        if (a.pos.size() != 6) {
            std::cerr << "usage:\n" << argv[0] << " left right out_folder prefix P1 P2" << std::endl;
            return 1;
        }
        const std::string out = a.pos[2], prefix = a.pos[3];
        recon::SGMParams p;
        p.disp_range = a.num("disp-range", 255); // sgm_cnn/main.cc:24
        p.window_sz = 0;
        p.penalty1 = sgmx::cli::to_int(a.pos[4], "P1");
        p.penalty2 = sgmx::cli::to_int(a.pos[5], "P2");
        recon::SGM sgm(p, a.num("device", 0), a.num("paths", 8));
        const sgmx::Mat img_disp = sgm.Compute<sgmx::Mat>(a.pos[0], a.pos[1]);
        const sgmx::Mat img8 = sgmx::host::normalize_minmax_u8(sgmx::host::convert_to_u8(img_disp, 1.0 / 256.0));
        sgmx::png::write_gray(out + "/disparities/" + prefix + ".png", img_disp);
        sgmx::png::write_gray(out + "/norm_hist/" + prefix + ".png", img8);
        return 0;
    });
}
