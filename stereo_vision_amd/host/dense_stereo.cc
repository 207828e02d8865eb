// dense_stereo -- counterpart of reconstruction/main/dense_stereo/main.cc:42-210.
//   dense_stereo -c config.txt   (keys source_folder, disp_folder, imglist; also -s/-d/-l on argv)
// The image list is an OpenCV-XML string sequence with left and right names alternating
// (config_files/kitti_00_lst.xml); every pair is Gaussian-blurred (3x3, sigma 0.7, :83-85), run
// through StereoSGM (window 5, P1 = 7, P2 = 60, disp_range = 270, :87-97) and saved as
// disp_folder/img/disp_<n>.png (8-bit, saturated, histogram-equalised) and
// disp_folder/mat/disp_<n>.yml (the CV_16U matrix).  What the reference hard-codes are options:
// --disp-range --window-sz --p1 --p2 --paths --device --sigma (0 < sigma, or --no-blur) --no-yml.
#include "cli_common.hpp"
#include "host_util.hpp"
#include "png_io.hpp"

int main(int argc, char **argv)
{
    return sgmx::cli::guarded("dense_stereo", [&]() {
        const sgmx::cli::Args a = sgmx::cli::parse(
            argc, argv, {"help", "no-blur", "no-yml"},
            {{'c', "config"}, {'s', "source_folder"}, {'d', "disp_folder"}, {'l', "imglist"}});
        const char *usage = "  --help\n  -c [ --config ] arg (=config.txt)  config filename\n"
                            "  -s [ --source_folder ] arg  folder with source\n  -d [ --disp_folder ] arg  folder for output\n"
                            "  -l [ --imglist ] arg  file with image list\n";
        if (a.has("help")) {
            std::cout << usage;
            return 0;
        }
        const std::string config_file = a.str("config", "config.txt");
        std::map<std::string, std::string> kv;
        if (!sgmx::host::read_config_file(config_file, kv)) {
            std::cout << "can not open config file: " << config_file << "\n" << usage;
            return 0; // dense_stereo/main.cc:155-160 returns 0 here
        }
        // command line first, then the config file (boost stores the first value it sees)
        const std::string source_folder = a.str("source_folder", kv.count("source_folder") ? kv["source_folder"] : "");
        const std::string output_folder = a.str("disp_folder", kv.count("disp_folder") ? kv["disp_folder"] : "");
        const std::string imagelistfn = a.str("imglist", kv.count("imglist") ? kv["imglist"] : "");
        std::cout << "Configuring done, using:" << std::endl;
        std::cout << "Source folder: " << source_folder << "\nDisparity folder: " << output_folder
                  << "\nImage list file: " << imagelistfn << std::endl;
        if (imagelistfn.empty()) {
            std::cout << "error: no xml image list given." << std::endl;
            return -1;
        }
        if (output_folder.empty() || source_folder.empty()) {
            std::cout << "error: no output or source folder given." << std::endl;
            return -1;
        }
        std::vector<std::string> imagelist;
        if (!sgmx::host::read_string_list(imagelistfn, imagelist) || imagelist.empty()) {
            std::cout << "can not open " << imagelistfn << " or the string list is empty" << std::endl;
            return -1;
        }
        recon::StereoSGMParams p;
        p.window_sz = a.num("window-sz", 5);
        p.penalty2 = a.num("p2", 60);
        p.disp_range = a.num("disp-range", 270);
        p.penalty1 = a.num("p1", 7);
        const double sigma = a.has("no-blur") ? 0.0 : std::atof(a.str("sigma", "0.7").c_str());
        recon::StereoSGM sgm(p, a.num("device", 0), a.num("paths", 8));
        sgmx::host::mkdirs(output_folder + "img");
        sgmx::host::mkdirs(output_folder + "mat");
        for (size_t i = 0; i + 1 < imagelist.size(); i += 2) {
            std::cout << source_folder + imagelist[i] << "\n";
            sgmx::Mat img_left = sgmx::png::read_gray8(source_folder + imagelist[i]);
            sgmx::Mat img_right = sgmx::png::read_gray8(source_folder + imagelist[i + 1]);
            if (!a.has("no-blur")) {
                sgm.gaussian_blur3(img_left, sigma);
                sgm.gaussian_blur3(img_right, sigma);
            }
            sgmx::Mat mat_disp;
            sgm.compute(img_left, img_right, mat_disp);
            // dense_stereo/main.cc:106-107: substr(9, size - 13) of "/image_0/000000.png" is "000000"
            const std::string &nm = imagelist[i];
            const std::string tag = nm.size() >= 14 ? nm.substr(9, nm.size() - 13) : sgmx::host::stem(nm);
            const std::string img_fname = output_folder + "img/disp_" + tag + ".png";
            const std::string mat_fname = output_folder + "mat/disp_" + tag + ".yml";
            std::cout << "Saving: " << img_fname << "\n" << mat_fname << "\n----------------------\n";
            sgmx::png::write_gray(img_fname, sgmx::host::equalize_hist(sgmx::host::convert_to_u8(mat_disp, 1.0)), 9);
            if (!a.has("no-yml") && !sgmx::host::write_yml_u16(mat_fname, "disparity_matrix", mat_disp))
                throw std::runtime_error("cannot write " + mat_fname);
        }
        return 0;
    });
}
