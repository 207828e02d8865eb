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

#include "batch_pipeline.hpp"

namespace {

typedef sgmx::host::BatchItem Item;
typedef sgmx::host::BatchCounters Counters;

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

struct Options {
    bool cnn = false, write = true;
    std::string out;
    int P1 = 3, P2 = 60, disp_range = 256, window_sz = 3, paths = 8, batch = 0, threads = 16;
};

int env_int(const char *name, int def) { return sgmx::host::batch_env_int(name, def); }

Counters run_slice(const std::vector<Item> &items, const Options &o_in, int device)
{
    Options o = o_in;
    if (!o.cnn && o.batch <= 0)
        o.batch = items.empty() ? 1 : sgmx::host::default_batch(items[0].left, o.disp_range, o.paths);
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
        const Counters r = sgmx::host::run_census_pipeline(
            sgm, items, o.batch, o.threads,
            [](const Item &it) { std::cout << it.left.substr(it.left.rfind('/') + 1) << "\n"; },
            [&o](const Item &it, const sgmx::Mat &disp) {
                if (o.write)
                    write_outputs(o.out, it.prefix, disp);
            });
        c.frames = r.frames;
        c.checksum_xor = r.checksum_xor;
    }
    c.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return c;
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
