// sgm -- counterpart of reconstruction/base/new_sgm/sgm_stereo_main.cc:46-162 (the `sgm` binary of
// new_sgm/CMakeLists.txt:19, the alternate binary_path of scripts/run_sgm_batch_cnn.py:7).
//   sgm left.bin right.bin out_folder P1 P2 consistency_threshold      (descriptor files)
//   sgm data_cost.bin out_folder P1 P2 consistency_threshold           (precomputed [H][W][D] float cost)
// Outputs out_folder/disparities/<base>.png (16-bit) and out_folder/norm_hist/<base>.png, where
// <base> is the input's file name without its last five characters and extension
// ("000007_10_left.bin" -> "000007_10"; :77-81).  Like the reference's main, a wrong argument count
// prints the usage and returns 0.  Option: --device N.
#include <cmath>

#include "cli_common.hpp"
#include "host_util.hpp"
#include "png_io.hpp"

namespace {

// ConvertMat16ToMat8 (sgm_stereo_main.cc:26-31): min(255, round(v / 256.0)), round half away from zero
sgmx::Mat convert_mat16_to_mat8(const sgmx::Mat &m16)
{
    sgmx::Mat m8(m16.rows, m16.cols, sgmx::MAT_8U);
    for (int i = 0; i < m8.rows; i++)
        for (int j = 0; j < m8.cols; j++)
            m8.at<uint8_t>(i, j) = (uint8_t)std::min(255.0, std::round(m16.at<uint16_t>(i, j) / 256.0));
    return m8;
}

std::string base_name(std::string p)
{
    const size_t slash = p.rfind('/');
    if (slash != std::string::npos)
        p.erase(0, slash + 1);
    const size_t dot = p.rfind('.');
    if (dot != std::string::npos && dot >= 5)
        p.erase(dot - 5);
    return p;
}

} // namespace

int main(int argc, char **argv)
{
    return sgmx::cli::guarded("sgm", [&]() {
        const sgmx::cli::Args a = sgmx::cli::parse(argc, argv);
        if (a.pos.size() != 6 && a.pos.size() != 5) {
            std::cerr << "usage: ./sgm left right out_folder P1 P2 consistency_threshold" << std::endl;
            std::cerr << "usage: ./sgm data_cost_path out_folder P1 P2 consistency_threshold" << std::endl;
            return 0;
        }
        const bool desc = a.pos.size() == 6;
        const size_t k = desc ? 2 : 1;
        const std::string out_folder = a.pos[k];
        recon::SGMStereo sgm(a.num("device", 0));
        sgm.SetSmoothnessCostParameters(sgmx::cli::to_int(a.pos[k + 1], "P1"), sgmx::cli::to_int(a.pos[k + 2], "P2"));
        sgm.SetConsistencyThreshold(sgmx::cli::to_int(a.pos[k + 3], "consistency_threshold"));
        sgmx::Mat img16;
        if (desc)
            sgm.Compute(a.pos[0], a.pos[1], &img16);
        else
            sgm.Compute(a.pos[0], &img16);
        const std::string save_name = base_name(a.pos[0]) + ".png";
        const sgmx::Mat img8 = sgmx::host::normalize_minmax_u8(convert_mat16_to_mat8(img16));
        sgmx::png::write_gray(out_folder + "/disparities/" + save_name, img16);
        sgmx::png::write_gray(out_folder + "/norm_hist/" + save_name, img8);
        return 0;
    });
}
