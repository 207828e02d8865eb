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
        } else if (cmd == "zfuzz" && a.pos.size() == 2) {
            // The codec's own inflate and deflate against zlib: streams of five kinds of data from every zlib level,
            // strategy and memory level must decode to the input; damaged, truncated and wrongly sized streams
            // must be rejected (or, if a flipped bit happens to survive the Adler-32 check, decode to the input)
            // without a write past the output; the own encoder's streams must decode with both inflaters.
            uint64_t rs = 88172645463325252ull;
            auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 16); };
            const int rounds = std::atoi(a.pos[1].c_str());
            long cases = 0, bad = 0;
            const size_t sizes[] = {0, 1, 2, 100, 4097, 70000, 300000};
            const int strategies[] = {Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED};
            for (int round = 0; round < rounds; round++)
                for (int kind = 0; kind < 5; kind++)
                    for (int level = 0; level <= 9; level++)
                        for (int strat : strategies)
                            for (size_t size : sizes) {
                                std::vector<unsigned char> src(size);
                                for (size_t i = 0; i < size; i++)
                                    src[i] = kind == 0   ? (unsigned char)rnd()
                                             : kind == 1 ? (unsigned char)(rnd() % 4 == 0 ? rnd() : 0)
                                             : kind == 2 ? (unsigned char)(i / 7 + rnd() % 3)
                                             : kind == 3 ? (unsigned char)"abcabcabd"[i % 9]
                                                         : (unsigned char)(rnd() % (1 + (i >> 8) % 40));
                                std::vector<unsigned char> z(size + size / 8 + 4096);
                                z_stream zs;
                                std::memset(&zs, 0, sizeof(zs));
                                if (deflateInit2(&zs, level, Z_DEFLATED, 15, 1 + (int)(cases % 9), strat) != Z_OK)
                                    return 4;
                                zs.next_in = src.data();
                                zs.avail_in = (uInt)size;
                                zs.next_out = z.data();
                                zs.avail_out = (uInt)z.size();
                                const int rc = deflate(&zs, Z_FINISH);
                                const size_t zl = zs.total_out;
                                deflateEnd(&zs);
                                if (rc != Z_STREAM_END)
                                    return 4;
                                cases++;
                                std::vector<unsigned char> out(size + 1, 0xAB);
                                if (!sgmx::png::inflate_zlib_exact(z.data(), zl, out.data(), size) ||
                                    (size && std::memcmp(out.data(), src.data(), size)) || out[size] != 0xAB)
                                    bad++;
                                if (size > 1 && sgmx::png::inflate_zlib_exact(z.data(), zl, out.data(), size - 1))
                                    bad++;
                                if (zl > 8) {
                                    std::vector<unsigned char> zc(z.begin(), z.begin() + zl), o3(size + 1, 0xCD);
                                    zc[2 + rnd() % (zl - 6)] ^= (unsigned char)(1u << (rnd() % 8));
                                    const bool r = sgmx::png::inflate_zlib_exact(zc.data(), zl, o3.data(), size);
                                    if (o3[size] != 0xCD || (r && size && std::memcmp(o3.data(), src.data(), size)))
                                        bad++;
                                    if (size && sgmx::png::inflate_zlib_exact(z.data(), zl - 3, o3.data(), size))
                                        bad++;
                                }
                                if (level == 0 && strat == Z_DEFAULT_STRATEGY) { // own encoder, both decoders
                                    const std::vector<unsigned char> own = sgmx::png::zlib_huffman_only(src.data(), size);
                                    std::vector<unsigned char> o4(size + 1);
                                    uLongf ol = (uLongf)o4.size();
                                    if (uncompress(o4.data(), &ol, own.data(), (uLong)own.size()) != Z_OK || ol != size ||
                                        (size && std::memcmp(o4.data(), src.data(), size)))
                                        bad++;
                                    if (!sgmx::png::inflate_zlib_exact(own.data(), own.size(), o4.data(), size) ||
                                        (size && std::memcmp(o4.data(), src.data(), size)))
                                        bad++;
                                }
                            }
            std::printf("cases %ld bad %ld\n", cases, bad);
            return bad ? 5 : 0;
        } else if (cmd == "pool" && a.pos.size() == 2) {
            // The batch driver's worker pool: jobs submitted from several threads at once (as the decoder, staging and
            // writer stages do), urgent jobs among them, every item run exactly once, exceptions reported to the
            // submitting thread only.
            sgmx::host::Pool pool(std::atoi(a.pos[1].c_str()));
            std::atomic<long> sum(0), caught(0);
            std::vector<std::thread> submitters;
            for (int t = 0; t < 6; t++)
                submitters.emplace_back([&, t]() {
                    for (int round = 0; round < 40; round++) {
                        const int n = 1 + (t * 7 + round * 3) % 37;
                        std::vector<int> hits(n, 0);
                        try {
                            pool.run(n, [&](int i) {
                                hits[i]++;
                                sum += i;
                                if (t == 5 && round % 8 == 0 && i == n / 2)
                                    throw std::runtime_error("item failed");
                            }, t % 2 == 1);
                            for (int h : hits)
                                if (h != 1)
                                    sum += 1000000000L; // every item exactly once
                        } catch (const std::exception &) {
                            caught++;
                        }
                    }
                });
            for (auto &t : submitters)
                t.join();
            long want = 0;
            for (int t = 0; t < 6; t++)
                for (int round = 0; round < 40; round++) {
                    const int n = 1 + (t * 7 + round * 3) % 37;
                    if (!(t == 5 && round % 8 == 0))
                        want += (long)n * (n - 1) / 2;
                }
            // failing jobs add some of their items before the exception stops them: leave them out of the sum check
            long failing = 0;
            for (int round = 0; round < 40; round += 8) {
                const int n = 1 + (5 * 7 + round * 3) % 37;
                failing += (long)n * (n - 1) / 2;
            }
            const long got = sum.load();
            std::printf("caught %ld sum %s\n", caught.load(), got >= want && got <= want + failing ? "ok" : "BAD");
            return caught.load() == 5 && got >= want && got <= want + failing ? 0 : 5;
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
