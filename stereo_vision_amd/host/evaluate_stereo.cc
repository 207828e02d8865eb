// evaluate_stereo -- counterpart of reconstruction/evaluation/evaluate_stereo.cc:121-327 (eval_stereo).
//   evaluate_stereo result_dir [--gt-folder DIR] [--count 200] [--device 0]
// For i < count: result_dir/%06d_10.png against gt-folder/disp_noc_0 and gt-folder/disp_occ_0 (the
// reference hard-codes the ground-truth folder, :124).  Writes result_dir/../stats_{noc,occ}_{out,avg}.txt
// (mean / min / max rows) and appends "out_noc out_all avg_noc avg_all density  result_dir" to
// result_dir/../stats.txt.  The per-image work runs on the GPU (sgmx_kitti_eval_u16).
#include <iomanip>

#include "cli_common.hpp"
#include "host_util.hpp"
#include "png_io.hpp"

namespace {

typedef std::vector<std::vector<float>> Rows;

// utils.h:22-41
float stat_mean(const Rows &e, size_t idx)
{
    float m = 0;
    for (auto &r : e)
        m += r[idx];
    return m / (float)e.size();
}
float stat_min(const Rows &e, size_t idx)
{
    float m = 1;
    for (auto &r : e)
        if (r[idx] < m)
            m = r[idx];
    return m;
}
float stat_max(const Rows &e, size_t idx)
{
    float m = 0;
    for (auto &r : e)
        if (r[idx] > m)
            m = r[idx];
    return m;
}

struct Errors {
    double out_noc = 0, out_all = 0, avg_noc = 0, avg_all = 0, density = 0;
};

bool eval(const std::string &result_dir, const std::string &gt_folder, int count, int device, Errors &ke)
{
    Rows noc_out, occ_out, noc_avg, occ_avg;
    sgmx_params q;
    std::memset(&q, 0, sizeof(q));
    q.disp_range = 16;
    q.window_sz = 1;
    q.penalty1 = 3;
    q.penalty2 = 60;
    q.device = device;
    sgmx::Context ctx(q);
    for (int i = 0; i < count; i++) {
        char prefix[64];
        std::snprintf(prefix, sizeof(prefix), "%06d_10", i);
        try {
            const sgmx::Mat noc = sgmx::png::read_gray16(gt_folder + "/disp_noc_0/" + prefix + ".png");
            const sgmx::Mat occ = sgmx::png::read_gray16(gt_folder + "/disp_occ_0/" + prefix + ".png");
            const std::string image_file = result_dir + "/" + prefix + ".png";
            std::cout << image_file << "\n";
            const sgmx::png::Raw hdr = sgmx::png::decode(image_file);
            if (hdr.ctype != 0 || hdr.depth != 16 || hdr.W != noc.cols || hdr.H != noc.rows) {
                std::printf("ERROR: Input must be png, 1 channel, 16 bit, %d x %d px", noc.cols, noc.rows);
                return false;
            }
            const sgmx::Mat disp = sgmx::png::read_gray16(image_file);
            if (occ.cols != noc.cols || occ.rows != noc.rows)
                return false;
            float e[26];
            ctx.check(sgmx_kitti_eval_u16(ctx.get(), 1, reinterpret_cast<const uint16_t *>(disp.data),
                                          reinterpret_cast<const uint16_t *>(noc.data),
                                          reinterpret_cast<const uint16_t *>(occ.data), noc.cols, noc.rows, e, nullptr),
                      "sgmx_kitti_eval_u16");
            noc_out.emplace_back(e, e + 11);
            occ_out.emplace_back(e + 11, e + 22);
            noc_avg.emplace_back(e + 22, e + 24);
            occ_avg.emplace_back(e + 24, e + 26);
        } catch (const std::exception &) {
            return false; // evaluate_stereo.cc:231-234
        }
    }
    if (noc_out.empty()) {
        std::printf("ERROR: Couldn't generate/store output statistics!");
        return false;
    }
    const struct {
        const char *name;
        const Rows *rows;
    } files[] = {{"stats_noc_out.txt", &noc_out}, {"stats_occ_out.txt", &occ_out}, {"stats_noc_avg.txt", &noc_avg},
                 {"stats_occ_avg.txt", &occ_avg}};
    for (auto &f : files) {
        FILE *fp = std::fopen((result_dir + "/../" + f.name).c_str(), "w");
        if (!fp) {
            std::printf("ERROR: Couldn't generate/store output statistics!");
            return false;
        }
        float (*stats[3])(const Rows &, size_t) = {stat_mean, stat_min, stat_max};
        for (auto st : stats) {
            for (size_t k = 0; k < (*f.rows)[0].size(); k++)
                std::fprintf(fp, "%f ", st(*f.rows, k));
            std::fprintf(fp, "\n");
        }
        std::fclose(fp);
    }
    ke.out_noc = 100.0 * stat_mean(noc_out, 4);
    ke.out_all = 100.0 * stat_mean(occ_out, 4);
    ke.avg_noc = stat_mean(noc_avg, 0);
    ke.avg_all = stat_mean(occ_avg, 0);
    ke.density = 100.0 * stat_mean(noc_out, 10);
    return true;
}

} // namespace

int main(int argc, char **argv)
{
    return sgmx::cli::guarded("evaluate_stereo", [&]() {
        const sgmx::cli::Args a = sgmx::cli::parse(argc, argv);
        // The reference compiles its ground-truth folder in (evaluate_stereo.cc:129, a path under the author's
        // home directory); here it is a required option.
        if (a.pos.size() != 1 || !a.has("gt-folder")) {
            std::cout << "Usage: ./eval_stereo result_dir --gt-folder <KITTI stereo training folder holding disp_noc/ and disp_occ/> "
                         "[--count 200] [--device 0]" << std::endl;
            return 1;
        }
        const std::string result_dir = a.pos[0];
        Errors ke;
        eval(result_dir, a.str("gt-folder", ""), a.num("count", 200), a.num("device", 0), ke);
        // the reference appends the line whether or not eval succeeded (:318-326)
        std::ofstream ofile(result_dir + "/../stats.txt", std::ofstream::app);
        ofile << std::fixed << std::setprecision(2) << ke.out_noc << "\t" << ke.out_all << "\t" << ke.avg_noc << "\t";
        ofile << ke.avg_all << "\t" << ke.density << "\t\t" << result_dir << "\n";
        return 0;
    });
}
