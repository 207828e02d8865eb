// batch_pipeline.hpp -- the census batch pipeline shared by sgm_batch (scripts/run_sgm_batch_census.py:25-34's loop)
// and dense_stereo (reconstruction/main/dense_stereo/main.cc:76-114's loop): the pairs of a list run through
// sgmx_compute_census_u8_batch in launch groups, with the PNG decoders of the next group and the encoders of the
// previous one on a pool of host threads beside the engine call, page-locked staging in two sets.
#pragma once
#include <atomic>
#include <chrono>
#include <future>
#include <thread>

#include "cli_common.hpp"
#include "host_util.hpp"
#include "png_io.hpp"

namespace sgmx {
namespace host {

struct BatchItem {
    std::string left, right, prefix;
};

struct BatchCounters {
    long frames = 0;
    double seconds = 0;
    uint64_t checksum_xor = 0;
};

struct BatchGroup {
    std::vector<BatchItem> items;
    std::vector<sgmx::Mat> left, right;
};

inline int batch_env_int(const char *name, int def)
{
    const char *v = std::getenv(name);
    return v && *v ? std::atoi(v) : def;
}

// Pairs per launch group when the caller names none: 32 fill the chip at KITTI size; larger images take
// what fits 48 GB of the engine's scratch (8 byte volumes of w h Dp bytes per pair dominate it).
inline int default_batch(const std::string &first_image, int disp_range, int paths)
{
    const sgmx::png::Raw &hdr = sgmx::png::decode(first_image);
    const int Dp = disp_range <= 32 ? 32 : (disp_range <= 64 ? 64 : (disp_range <= 128 ? 128 : (disp_range <= 256 ? 256 : (disp_range <= 384 ? 384 : 512))));
    const double per_pair = (double)hdr.W * hdr.H * ((double)paths * Dp + 24.0);
    const double fit = 48e9 / per_pair;
    return (int)std::max(1.0, std::min(32.0, fit));
}

// items through `sgm` (a context with max_batch >= batch) in groups of `batch` pairs.  announce(item) is called on the
// driving thread once the item's group is computed, in list order; write(item, disp) on a pool thread, any order.
// Returns the pair count and the XOR of the disparity maps' FNV checksums.
template <class Announce, class Write>
BatchCounters run_census_pipeline(recon::StereoSGM &sgm, const std::vector<BatchItem> &items, int batch, int threads,
                                  Announce announce, Write write)
{
    typedef BatchGroup Group;
    BatchCounters c;
    auto env_int = batch_env_int;
    Pool pool(threads);
    const size_t ngroups = (items.size() + batch - 1) / batch;
    std::atomic<long> dec_us(0), load_us(0); // SGMX_BATCH_TIMING: decoder thread time, wall time of the group loads
    auto load = [&](size_t g) {
        const auto tl0 = std::chrono::steady_clock::now();
        Group grp;
        const size_t lo = g * batch, hi = std::min(items.size(), lo + batch);
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
                announce(grp->items[k]);
            c.frames += n;
            tp = now();
            if (writer.valid())
                writer.get();
            t_wait_wr += since(tp);
            uint16_t *D = s.D;
            writer = std::async(std::launch::async, [grp, off, D, n, &write, &cks, &pool]() {
                pool.run(n, [&](int k) {
                    const sgmx::Mat disp(grp->left[k].rows, grp->left[k].cols, sgmx::MAT_16U, D + (*off)[k]);
                    cks.fetch_xor(sgmx::host::fnv_u16(disp));
                    write(grp->items[k], disp);
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
    return c;
}

} // namespace host
} // namespace sgmx
