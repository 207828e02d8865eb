// sgm_batch -- counterpart of scripts/run_sgm_batch_census.py:14-34 and run_sgm_batch_cnn.py:22-42.
//   sgm_batch census LEFT_DIR RIGHT_DIR OUT_DIR [P1 P2] [--disp-range 256] [--window-sz 3]
//   sgm_batch cnn DATA_DIR [P1 P2] [--disp-range 255]          (DATA_DIR/representation -> DATA_DIR/depth)
// common: [--gpus N [--share-gpu: rehearsal, every worker on device 0]] [--rank R --world W] [--batch B: pairs per launch group, default sized from the first image (up to 32 within 48 GB of device scratch)] [--threads 16: host worker threads per GPU worker, shared by decoders, copies and encoders] [--paths 8] [--token _10.png] [--no-write]
//
// Same file selection, output directories and per-pair outputs as the scripts, which start one
// sgm_single / sgm_cnn process per pair, strictly one after the other.  Here the pair list is cut
// into contiguous slices, one per GPU (stereo pairs are independent, nothing is exchanged between
// GPUs): --gpus N forks one worker process per GPU before anything touches HIP and collects the
// workers' counters over pipes; --rank/--world (or RANK / WORLD_SIZE / LOCAL_RANK in the
// environment) run one slice under an external launcher.  Within a worker, PNG decoding of the
// next group and encoding of the previous one overlap the GPU work on the current one.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <dirent.h>
#include <future>
#include <sys/wait.h>
#include <thread>
#include <unistd.h>

#include "cli_common.hpp"
#include "host_util.hpp"
#include "png_io.hpp"

namespace {

struct Item {
    std::string left, right, prefix;
};

std::vector<std::string> list_files(const std::string &dir)
{
    std::vector<std::string> v;
    DIR *d = ::opendir(dir.c_str());
    if (!d)
        throw std::runtime_error("cannot list " + dir);
    while (struct dirent *e = ::readdir(d)) {
        const std::string n = e->d_name;
        struct stat st;
        if (::stat((dir + "/" + n).c_str(), &st) == 0 && S_ISREG(st.st_mode))
            v.push_back(n);
    }
    ::closedir(d);
    std::sort(v.begin(), v.end());
    return v;
}

// run_sgm_batch_census.py:25-31: files of LEFT_DIR whose name contains "_10.png", same name in RIGHT_DIR
std::vector<Item> list_census(const std::string &ldir, const std::string &rdir, const std::string &token)
{
    std::vector<Item> v;
    for (auto &n : list_files(ldir))
        if (n.find(token) != std::string::npos)
            v.push_back({ldir + "/" + n, rdir + "/" + n, sgmx::host::stem(n)});
    return v;
}

// run_sgm_batch_cnn.py:33-41: files whose name contains "left"; prefix = name[:9]; right = prefix + "_right.bin"
std::vector<Item> list_cnn(const std::string &dir)
{
    std::vector<Item> v;
    for (auto &n : list_files(dir))
        if (n.find("left") != std::string::npos)
            v.push_back({dir + "/" + n, dir + "/" + n.substr(0, 9) + "_right.bin", n.substr(0, 9)});
    return v;
}

// run_sgm_batch_census.py:14-23
void make_output_dirs(const std::string &out)
{
    if (sgmx::host::is_dir(out))
        std::cout << "WARNING: path exists - " << out << std::endl;
    for (const char *sub : {"", "/disparities", "/norm_hist", "/interpolation"})
        if (!sgmx::host::mkdirs(out + sub))
            throw std::runtime_error("cannot create " + out + sub);
}

using sgmx::host::Pool;

void write_outputs(const std::string &out, const std::string &prefix, const sgmx::Mat &disp)
{
    sgmx::png::write_gray(out + "/disparities/" + prefix + ".png", disp);
    sgmx::png::write_gray(out + "/norm_hist/" + prefix + ".png",
                          sgmx::host::normalize_minmax_u8(sgmx::host::convert_to_u8(disp, 1.0 / 256.0)));
}

struct Counters {
    long frames = 0;
    double seconds = 0;
    uint64_t checksum_xor = 0;
};

struct Group {
    std::vector<Item> items;
    std::vector<sgmx::Mat> left, right;
};

struct Options {
    bool cnn = false, write = true;
    std::string out;
    int P1 = 3, P2 = 60, disp_range = 256, window_sz = 3, paths = 8, batch = 0, threads = 16;
};

int env_int(const char *name, int def);

// Pairs per launch group when --batch is not given: 32 fill the chip at KITTI size; larger images take
// what fits 48 GB of the engine's scratch (8 byte volumes of w h Dp bytes per pair dominate it).
int default_batch(const std::vector<Item> &items, const Options &o)
{
    if (items.empty())
        return 1;
    const sgmx::png::Raw &hdr = sgmx::png::decode(items[0].left);
    const int Dp = o.disp_range <= 32 ? 32 : (o.disp_range <= 64 ? 64 : (o.disp_range <= 128 ? 128 : (o.disp_range <= 256 ? 256 : (o.disp_range <= 384 ? 384 : 512))));
    const double per_pair = (double)hdr.W * hdr.H * ((double)o.paths * Dp + 24.0);
    const double fit = 48e9 / per_pair;
    return (int)std::max(1.0, std::min(32.0, fit));
}

Counters run_slice(const std::vector<Item> &items, const Options &o_in, int device)
{
    Options o = o_in;
    if (!o.cnn && o.batch <= 0)
        o.batch = default_batch(items, o);
    Counters c;
    const auto t0 = std::chrono::steady_clock::now();
    if (o.cnn) {
        recon::SGMParams p = {o.disp_range, 0, o.P1, o.P2};
        recon::SGM sgm(p, device, o.paths);
        // The two descriptor files of a pair (hundreds of MB at KITTI size) are read by two threads, one pair
        // ahead of the GPU; the previous pair's PNG files are written meanwhile.
        struct Loaded {
            sgmx::DescriptorTensor l, r;
        };
        auto load = [&](size_t i) {
            auto p = std::make_shared<Loaded>();
            auto fr = std::async(std::launch::async, [&, i]() { sgmx::LoadRepresentationFromFile(items[i].right, &p->r); });
            sgmx::LoadRepresentationFromFile(items[i].left, &p->l);
            fr.get();
            return p;
        };
        std::future<std::shared_ptr<Loaded>> next;
        std::future<void> writer;
        if (!items.empty())
            next = std::async(std::launch::async, load, (size_t)0);
        try {
            for (size_t i = 0; i < items.size(); i++) {
                const Item &it = items[i];
                std::cout << it.left.substr(it.left.rfind('/') + 1) << std::endl;
                const std::shared_ptr<Loaded> cur = next.get();
                if (i + 1 < items.size())
                    next = std::async(std::launch::async, load, i + 1);
                auto disp = std::make_shared<sgmx::Mat>(sgm.ComputeFromDescriptors<sgmx::Mat>(cur->l, cur->r));
                c.checksum_xor ^= sgmx::host::fnv_u16(*disp);
                c.frames++;
                if (writer.valid())
                    writer.get();
                if (o.write)
                    writer = std::async(std::launch::async, [disp, &o, &it]() { write_outputs(o.out, it.prefix, *disp); });
            }
            if (writer.valid())
                writer.get();
        } catch (...) {
            if (next.valid())
                next.wait();
            if (writer.valid())
                writer.wait();
            throw;
        }
    } else {
        recon::StereoSGMParams p = {o.disp_range, o.window_sz, o.P1, o.P2};
        recon::StereoSGM sgm(p, device, o.paths, o.batch);
        Pool pool(o.threads);
        const size_t ngroups = (items.size() + o.batch - 1) / o.batch;
        std::atomic<long> dec_us(0), load_us(0); // SGMX_BATCH_TIMING: decoder thread time, wall time of the group loads
        auto load = [&](size_t g) {
            const auto tl0 = std::chrono::steady_clock::now();
            Group grp;
            const size_t lo = g * o.batch, hi = std::min(items.size(), lo + o.batch);
            grp.items.assign(items.begin() + lo, items.begin() + hi);
            grp.left.resize(hi - lo);
            grp.right.resize(hi - lo);
            pool.run((int)(2 * (hi - lo)), [&](int i) {
                const auto td0 = std::chrono::steady_clock::now();
                if (i & 1)
                    grp.right[i / 2] = sgmx::png::read_gray8(grp.items[i / 2].right);
                else
                    grp.left[i / 2] = sgmx::png::read_gray8(grp.items[i / 2].left);
                dec_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - td0).count();
            });
            load_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tl0).count();
            return grp;
        };
        // Page-locked staging, two sets: while the engine works on set g & 1, the decoder threads of the next
        // group and the encoder threads of the previous one run; the previous group's disparities stay in the
        // other output buffer until its writer is done (it is joined before the buffer is used again).
        struct Staging {
            uint8_t *L = nullptr, *R = nullptr;
            uint16_t *D = nullptr;
            size_t pixels = 0;
        } stg[2];
        auto release = [&]() {
            for (auto &s : stg) {
                sgm.host_free(s.L);
                sgm.host_free(s.R);
                sgm.host_free(s.D);
                s = Staging();
            }
        };
        auto ensure = [&](Staging &s, size_t pixels) {
            if (pixels <= s.pixels)
                return;
            sgm.host_free(s.L);
            sgm.host_free(s.R);
            sgm.host_free(s.D);
            s.L = static_cast<uint8_t *>(sgm.host_alloc(pixels));
            s.R = static_cast<uint8_t *>(sgm.host_alloc(pixels));
            s.D = static_cast<uint16_t *>(sgm.host_alloc(pixels * 2));
            s.pixels = pixels;
        };
        std::atomic<uint64_t> cks(0);
        std::future<Group> next;
        std::future<void> writer;
        try {
            if (ngroups)
                next = std::async(std::launch::async, load, (size_t)0);
            // SGMX_BATCH_TIMING=1: where the driving thread's time goes, on stderr at the end of the slice
            const bool timing = env_int("SGMX_BATCH_TIMING", 0) != 0;
            double t_wait_dec = 0, t_stage = 0, t_engine = 0, t_wait_wr = 0;
            auto now = []() { return std::chrono::steady_clock::now(); };
            auto since = [](std::chrono::steady_clock::time_point a) {
                return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
            };
            for (size_t g = 0; g < ngroups; g++) {
                auto tp = now();
                auto grp = std::make_shared<Group>(next.get());
                t_wait_dec += since(tp);
                const int n = (int)grp->items.size();
                Staging &s = stg[g & 1];
                size_t need = 0;
                for (int k = 0; k < n; k++) {
                    if (grp->left[k].cols != grp->right[k].cols || grp->left[k].rows != grp->right[k].rows)
                        throw std::runtime_error("left and right image differ in size: " + grp->items[k].left);
                    need += (size_t)grp->left[k].cols * grp->left[k].rows;
                }
                ensure(s, need);
                // pair k of the group sits at pixel offset off[k] of the staging buffers
                auto off = std::make_shared<std::vector<size_t>>(n + 1, 0);
                for (int k = 0; k < n; k++)
                    (*off)[k + 1] = (*off)[k] + (size_t)grp->left[k].cols * grp->left[k].rows;
                tp = now();
                pool.run(2 * n, [&](int i) {
                    const sgmx::Mat &m = (i & 1) ? grp->right[i / 2] : grp->left[i / 2];
                    std::memcpy(((i & 1) ? s.R : s.L) + (*off)[i / 2], m.data, (size_t)m.cols * m.rows);
                }, true);
                t_stage += since(tp);
                // the next group's decoders get the pool once this group's copies are through it, and then run
                // beside the engine call
                if (g + 1 < ngroups)
                    next = std::async(std::launch::async, load, g + 1);
                tp = now();
                // runs of equal size go to the engine in one call
                for (int i = 0; i < n;) {
                    const int W = grp->left[i].cols, H = grp->left[i].rows;
                    int j = i;
                    while (j < n && grp->left[j].cols == W && grp->left[j].rows == H)
                        j++;
                    sgm.compute_batch(j - i, s.L + (*off)[i], s.R + (*off)[i], W, H, s.D + (*off)[i]);
                    i = j;
                }
                t_engine += since(tp);
                for (int k = 0; k < n; k++)
                    std::cout << grp->items[k].left.substr(grp->items[k].left.rfind('/') + 1) << "\n";
                c.frames += n;
                tp = now();
                if (writer.valid())
                    writer.get();
                t_wait_wr += since(tp);
                uint16_t *D = s.D;
                writer = std::async(std::launch::async, [grp, off, D, n, &o, &cks, &pool]() {
                    pool.run(n, [&](int k) {
                        const sgmx::Mat disp(grp->left[k].rows, grp->left[k].cols, sgmx::MAT_16U, D + (*off)[k]);
                        cks.fetch_xor(sgmx::host::fnv_u16(disp));
                        if (o.write)
                            write_outputs(o.out, grp->items[k].prefix, disp);
                    });
                });
            }
            if (writer.valid())
                writer.get();
            if (timing)
                std::fprintf(stderr, "sgm_batch timing (s): wait for decoders %.3f, staging copies %.3f, engine calls %.3f, wait for writers %.3f\n",
                             t_wait_dec, t_stage, t_engine, t_wait_wr);
            if (timing)
                std::fprintf(stderr, "sgm_batch timing (s): group loads %.3f wall, %.3f summed over the decoder threads\n",
                             load_us.load() * 1e-6, dec_us.load() * 1e-6);
        } catch (...) {
            if (next.valid())
                next.wait();
            if (writer.valid())
                writer.wait();
            release();
            throw;
        }
        release();
        c.checksum_xor = cks.load();
    }
    c.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return c;
}

int env_int(const char *name, int def)
{
    const char *v = std::getenv(name);
    return v && *v ? std::atoi(v) : def;
}

} // namespace

int main(int argc, char **argv)
{
    return sgmx::cli::guarded("sgm_batch", [&]() {
        const sgmx::cli::Args a = sgmx::cli::parse(argc, argv, {"no-write", "share-gpu"});
        const char *usage = "usage:\n  sgm_batch census left_dir right_dir out_dir [P1 P2] [--gpus N] [--batch B] ...\n"
                            "  sgm_batch cnn data_dir [P1 P2] [--gpus N] ...";
        if (a.pos.empty() || (a.pos[0] != "census" && a.pos[0] != "cnn")) {
            std::cerr << usage << std::endl;
            return 1;
        }
        Options o;
        o.cnn = a.pos[0] == "cnn";
        const size_t fixed = o.cnn ? 2 : 4;
        if (a.pos.size() != fixed && a.pos.size() != fixed + 2) {
            std::cerr << usage << std::endl;
            return 1;
        }
        if (a.pos.size() == fixed + 2) {
            o.P1 = sgmx::cli::to_int(a.pos[fixed], "P1");
            o.P2 = sgmx::cli::to_int(a.pos[fixed + 1], "P2");
        }
        o.disp_range = a.num("disp-range", o.cnn ? 255 : 256);
        o.window_sz = a.num("window-sz", 3);
        o.paths = a.num("paths", 8);
        o.batch = std::max(0, a.num("batch", 0)); // 0: from the first image, in run_slice
        o.threads = std::max(1, a.num("threads", 16));
        o.write = !a.has("no-write");
        std::vector<Item> items;
        if (o.cnn) {
            items = list_cnn(a.pos[1] + "/representation"); // run_sgm_batch_cnn.py:22-23
            o.out = a.pos[1] + "/depth";
        } else {
            items = list_census(a.pos[1], a.pos[2], a.str("token", "_10.png"));
            o.out = a.pos[3];
        }
        const int gpus = a.num("gpus", 0);
        if (gpus > 64)
            throw std::invalid_argument("--gpus: at most 64");
        int rank = a.num("rank", env_int("RANK", 0)), world = a.num("world", env_int("WORLD_SIZE", 1));
        if (world < 1 || rank < 0 || rank >= world)
            throw std::invalid_argument("bad --rank / --world");
        if (o.write && rank == 0)
            make_output_dirs(o.out);

        std::vector<Counters> per;
        if (gpus <= 1) {
            long lo, hi;
            sgmx::host::shard_range((long)items.size(), rank, world, &lo, &hi);
            const int device = a.num("device", gpus == 1 ? 0 : env_int("LOCAL_RANK", 0));
            per.push_back(run_slice(std::vector<Item>(items.begin() + lo, items.begin() + hi), o, device));
            world = 1; // this process reports its own slice
        } else {
            // one worker process per GPU, forked before this process has touched HIP
            struct Child {
                pid_t pid;
                int fd;
            };
            std::vector<Child> kids;
            std::cout.flush();
            for (int r = 0; r < gpus; r++) {
                int fds[2];
                if (::pipe(fds) != 0)
                    throw std::runtime_error("pipe failed");
                const pid_t pid = ::fork();
                if (pid < 0)
                    throw std::runtime_error("fork failed");
                if (pid == 0) {
                    ::close(fds[0]);
                    int status = 0;
                    Counters c;
                    try {
                        long lo, hi;
                        sgmx::host::shard_range((long)items.size(), r, gpus, &lo, &hi);
                        c = run_slice(std::vector<Item>(items.begin() + lo, items.begin() + hi), o, a.has("share-gpu") ? 0 : r);
                    } catch (const std::exception &e) {
                        std::cerr << "sgm_batch[rank " << r << "]: " << e.what() << std::endl;
                        status = 2;
                    }
                    if (::write(fds[1], &c, sizeof(c)) != (ssize_t)sizeof(c))
                        status = 2;
                    ::close(fds[1]);
                    std::cout.flush();
                    ::_exit(status);
                }
                ::close(fds[1]);
                kids.push_back({pid, fds[0]});
            }
            bool failed = false;
            for (auto &k : kids) {
                Counters c;
                const bool got = ::read(k.fd, &c, sizeof(c)) == (ssize_t)sizeof(c);
                ::close(k.fd);
                int st = 0;
                ::waitpid(k.pid, &st, 0);
                failed = failed || !got || !WIFEXITED(st) || WEXITSTATUS(st) != 0;
                per.push_back(c);
            }
            if (failed)
                throw std::runtime_error("a worker failed");
            world = gpus;
        }
        Counters tot;
        for (auto &c : per) {
            tot.frames += c.frames;
            tot.seconds = std::max(tot.seconds, c.seconds);
            tot.checksum_xor ^= c.checksum_xor;
        }
        std::printf("pairs: %ld  ranks: %d  wall: %.3f s  %.2f pairs/s  checksum xor: %016llx\n", tot.frames, world,
                    tot.seconds, tot.frames / std::max(tot.seconds, 1e-9), (unsigned long long)tot.checksum_xor);
        return 0;
    });
}
