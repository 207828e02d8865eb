// host_tool -- exposes the host-side helpers (PNG codec, image list / config readers, 8-bit
// previews, descriptor loader) one by one, so that tests can check each against an independent
// implementation without a GPU.  Not a counterpart of anything in the reference.
#include "cli_common.hpp"
#include "host_util.hpp"
#include "png_io.hpp"

int main(int argc, char **argv)
{
    return sgmx::cli::guarded("host_tool", [&]() {
        const sgmx::cli::Args a = sgmx::cli::parse(argc, argv);
        if (a.pos.empty()) {
            std::cerr << "usage: host_tool gray8|gray16|div256|norm|equalize|yml|list|config|desc|shard ..." << std::endl;
            return 1;
        }
        const std::string cmd = a.pos[0];
        if (cmd == "gray8" && a.pos.size() == 3) { // any PNG -> 8-bit gray PNG (imread GRAYSCALE + imwrite)
            sgmx::png::write_gray(a.pos[2], sgmx::png::read_gray8(a.pos[1]));
        } else if (cmd == "gray16" && a.pos.size() == 3) { // 16-bit gray PNG -> 16-bit gray PNG
            sgmx::png::write_gray(a.pos[2], sgmx::png::read_gray16(a.pos[1]), a.num("level", -1));
        } else if (cmd == "div256" && a.pos.size() == 3) { // convertTo(CV_8U, 1/256)
            sgmx::png::write_gray(a.pos[2], sgmx::host::convert_to_u8(sgmx::png::read_gray16(a.pos[1]), 1.0 / 256.0));
        } else if (cmd == "sat8" && a.pos.size() == 3) { // convertTo(CV_8U)
            sgmx::png::write_gray(a.pos[2], sgmx::host::convert_to_u8(sgmx::png::read_gray16(a.pos[1]), 1.0));
        } else if (cmd == "norm" && a.pos.size() == 3) {
            sgmx::png::write_gray(a.pos[2], sgmx::host::normalize_minmax_u8(sgmx::png::read_gray8(a.pos[1])));
        } else if (cmd == "equalize" && a.pos.size() == 3) {
            sgmx::png::write_gray(a.pos[2], sgmx::host::equalize_hist(sgmx::png::read_gray8(a.pos[1])));
        } else if (cmd == "yml" && a.pos.size() == 3) {
            if (!sgmx::host::write_yml_u16(a.pos[2], "disparity_matrix", sgmx::png::read_gray16(a.pos[1])))
                return 2;
        } else if (cmd == "fnv" && a.pos.size() == 2) {
            std::printf("%016llx\n", (unsigned long long)sgmx::host::fnv_u16(sgmx::png::read_gray16(a.pos[1])));
        } else if (cmd == "list" && a.pos.size() == 2) {
            std::vector<std::string> l;
            if (!sgmx::host::read_string_list(a.pos[1], l))
                return 3;
            for (auto &s : l)
                std::cout << s << "\n";
        } else if (cmd == "config" && a.pos.size() == 2) {
            std::map<std::string, std::string> kv;
            if (!sgmx::host::read_config_file(a.pos[1], kv))
                return 3;
            for (auto &e : kv)
                std::cout << e.first << "=" << e.second << "\n";
        } else if (cmd == "desc" && a.pos.size() == 2) {
            sgmx::DescriptorTensor t;
            sgmx::LoadRepresentationFromFile(a.pos[1], &t);
            double sum = 0;
            for (float v : t.v)
                sum += v;
            std::printf("%llu %llu %llu %.9g\n", (unsigned long long)t.H, (unsigned long long)t.W, (unsigned long long)t.K, sum);
        } else if (cmd == "shard" && a.pos.size() == 4) {
            long lo, hi;
            sgmx::host::shard_range(std::atol(a.pos[1].c_str()), std::atoi(a.pos[2].c_str()), std::atoi(a.pos[3].c_str()), &lo, &hi);
            std::printf("%ld %ld\n", lo, hi);
        } else {
            std::cerr << "host_tool: bad command" << std::endl;
            return 1;
        }
        return 0;
    });
}
