// dense_stereo -- counterpart of reconstruction/main/dense_stereo/main.cc:42-210.
//   dense_stereo -c config.txt   (keys source_folder, disp_folder, imglist; also -s/-d/-l on argv)
// The image list is an OpenCV-XML string sequence with left and right names alternating
// (config_files/kitti_00_lst.xml); every pair is Gaussian-blurred (3x3, sigma 0.7, :83-85), run
// through StereoSGM (window 5, P1 = 7, P2 = 60, disp_range = 270, :87-97) and saved as
// disp_folder/img/disp_<n>.png (8-bit, saturated, histogram-equalised) and
// disp_folder/mat/disp_<n>.yml (the CV_16U matrix).  What the reference hard-codes are options:
// --disp-range --window-sz --p1 --p2 --paths --device --sigma (0 < sigma, or --no-blur) --no-yml.
// The reference decodes, blurs, computes and writes one pair after the other; here the list runs through the batch
// pipeline of sgm_batch (batch_pipeline.hpp): launch groups of --batch pairs (default: sized from the first image)
// through sgmx_compute_census_u8_batch with the Gaussian pre-filter fused in front of the census transform on the
// device (sgmx_set_preblur3), decoders and writers on --threads host threads beside the engine.  Same files, same
// bytes.  --one-at-a-time keeps the reference's loop (one compute() per pair, blur through its own calls).
#include "batch_pipeline.hpp"

int main(int argc, char **argv)
{
    return sgmx::cli::guarded("dense_stereo", [&]() {
        const sgmx::cli::Args a = sgmx::cli::parse(
            argc, argv, {"help", "no-blur", "no-yml", "one-at-a-time"},
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
        sgmx::host::mkdirs(output_folder + "img");
        sgmx::host::mkdirs(output_folder + "mat");
        const bool no_yml = a.has("no-yml");
        // dense_stereo/main.cc:106-107: substr(9, size - 13) of "/image_0/000000.png" is "000000"
        auto tag_of = [](const std::string &nm) { return nm.size() >= 14 ? nm.substr(9, nm.size() - 13) : sgmx::host::stem(nm); };
        auto save = [&](const std::string &tag, const sgmx::Mat &mat_disp) {
            const std::string img_fname = output_folder + "img/disp_" + tag + ".png";
            const std::string mat_fname = output_folder + "mat/disp_" + tag + ".yml";
            sgmx::png::write_gray(img_fname, sgmx::host::equalize_hist(sgmx::host::convert_to_u8(mat_disp, 1.0)), 9);
            if (!no_yml && !sgmx::host::write_yml_u16(mat_fname, "disparity_matrix", mat_disp))
                throw std::runtime_error("cannot write " + mat_fname);
        };
        auto say = [&](const std::string &name) {
            const std::string tag = tag_of(name);
            std::cout << source_folder + name << "\n"
                      << "Saving: " << output_folder + "img/disp_" + tag + ".png" << "\n"
                      << output_folder + "mat/disp_" + tag + ".yml" << "\n----------------------\n";
        };
        const auto t0 = std::chrono::steady_clock::now();
        long npairs = 0;
        if (a.has("one-at-a-time")) {
            recon::StereoSGM sgm(p, a.num("device", 0), a.num("paths", 8));
            for (size_t i = 0; i + 1 < imagelist.size(); i += 2, npairs++) {
                sgmx::Mat img_left = sgmx::png::read_gray8(source_folder + imagelist[i]);
                sgmx::Mat img_right = sgmx::png::read_gray8(source_folder + imagelist[i + 1]);
                if (!a.has("no-blur")) {
                    sgm.gaussian_blur3(img_left, sigma);
                    sgm.gaussian_blur3(img_right, sigma);
                }
                sgmx::Mat mat_disp;
                sgm.compute(img_left, img_right, mat_disp);
                say(imagelist[i]);
                save(tag_of(imagelist[i]), mat_disp);
            }
        } else {
            std::vector<sgmx::host::BatchItem> items;
            for (size_t i = 0; i + 1 < imagelist.size(); i += 2)
                items.push_back({source_folder + imagelist[i], source_folder + imagelist[i + 1], imagelist[i]});
            int batch = a.num("batch", 0);
            if (batch <= 0 && !items.empty())
                batch = sgmx::host::default_batch(items[0].left, p.disp_range, a.num("paths", 8));
            batch = (int)std::max<long>(1, std::min<long>(batch, (long)items.size()));
            recon::StereoSGM sgm(p, a.num("device", 0), a.num("paths", 8), batch);
            if (!a.has("no-blur"))
                sgm.set_preblur(sigma);
            const sgmx::host::BatchCounters r = sgmx::host::run_census_pipeline(
                sgm, items, batch, std::max(1, a.num("threads", 16)),
                [&](const sgmx::host::BatchItem &it) { say(it.prefix); },
                [&](const sgmx::host::BatchItem &it, const sgmx::Mat &disp) { save(tag_of(it.prefix), disp); });
            npairs = r.frames;
        }
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::fprintf(stderr, "dense_stereo: %ld pairs in %.3f s = %.1f pairs/s\n", npairs, secs, npairs / std::max(secs, 1e-9));
        return 0;
    });
}
