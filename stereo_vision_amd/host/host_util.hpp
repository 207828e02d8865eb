// host_util.hpp -- caller-side helpers of the CLI programs: what the reference's mains get from
// OpenCV / boost around the SGM call (8-bit previews, image lists, config files, output naming).
// None of this is SGM arithmetic; the OpenCV functions restated here are not part of the reference
// tree (version unpinned), their semantics are taken from the 2.4-era library the mains target.
#ifndef SGMX_HOST_UTIL_HPP
#define SGMX_HOST_UTIL_HPP

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <sys/stat.h>
#include <vector>

#include "sgmx_recon.hpp"

namespace sgmx {
namespace host {

inline int cv_round(double v) { return (int)std::lrint(v); } // cvRound: nearest, ties to even
inline unsigned char sat_u8(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// src.convertTo(dst, CV_8U, alpha) for a CV_16U source (sgm_single/main.cc:45 uses alpha = 1/256,
// dense_stereo/main.cc:101 alpha = 1): saturate_cast<uchar>(cvRound(v * alpha)), float arithmetic.
template <class MatT> inline Mat convert_to_u8(const MatT &src, double alpha = 1.0)
{
    Mat dst(src.rows, src.cols, MAT_8U);
    const float a = (float)alpha;
    for (int y = 0; y < src.rows; y++) {
        const uint16_t *s = reinterpret_cast<const uint16_t *>(src.data + (size_t)y * (size_t)src.step);
        unsigned char *d = dst.data + (size_t)y * dst.step;
        for (int x = 0; x < src.cols; x++)
            d[x] = sat_u8(cv_round((float)s[x] * a));
    }
    return dst;
}

// cv::normalize(src, dst, 0, 255, cv::NORM_MINMAX) on CV_8U (sgm_single/main.cc:56, sgm_cnn/main.cc:50):
// scale = 255 / (max - min) (0 for a constant image), shift = -min * scale, then the float convertTo.
inline Mat normalize_minmax_u8(const Mat &src)
{
    int lo = 255, hi = 0;
    for (int y = 0; y < src.rows; y++)
        for (int x = 0; x < src.cols; x++) {
            const int v = src.at<unsigned char>(y, x);
            lo = v < lo ? v : lo;
            hi = v > hi ? v : hi;
        }
    const double scale = hi > lo ? 255.0 / (hi - lo) : 0.0, shift = -lo * scale;
    const float fs = (float)scale, fsh = (float)shift;
    Mat dst(src.rows, src.cols, MAT_8U);
    for (int y = 0; y < src.rows; y++)
        for (int x = 0; x < src.cols; x++) {
            const float t = (float)src.at<unsigned char>(y, x) * fs; // two rounded float operations, no fused multiply-add
            dst.at<unsigned char>(y, x) = sat_u8(cv_round(t + fsh));
        }
    return dst;
}

// cv::equalizeHist on CV_8U (dense_stereo/main.cc:108, stereo_eval/main.cc:55).
inline Mat equalize_hist(const Mat &src)
{
    long hist[256] = {0};
    for (int y = 0; y < src.rows; y++)
        for (int x = 0; x < src.cols; x++)
            hist[src.at<unsigned char>(y, x)]++;
    const long total = (long)src.rows * src.cols;
    int i = 0;
    while (i < 255 && !hist[i])
        i++;
    unsigned char lut[256] = {0};
    Mat dst(src.rows, src.cols, MAT_8U);
    if (hist[i] == total) {
        for (int y = 0; y < src.rows; y++)
            for (int x = 0; x < src.cols; x++)
                dst.at<unsigned char>(y, x) = (unsigned char)i;
        return dst;
    }
    const float scale = 255.f / (float)(total - hist[i]);
    long sum = 0;
    for (lut[i++] = 0; i < 256; i++) {
        sum += hist[i];
        lut[i] = sat_u8(cv_round((float)sum * scale));
    }
    for (int y = 0; y < src.rows; y++)
        for (int x = 0; x < src.cols; x++)
            dst.at<unsigned char>(y, x) = lut[src.at<unsigned char>(y, x)];
    return dst;
}

// sgm_single/main.cc:47-51: strip the directory and the last extension.
inline std::string stem(std::string p)
{
    const size_t slash = p.rfind('/');
    if (slash != std::string::npos)
        p.erase(0, slash + 1);
    const size_t dot = p.rfind('.');
    if (dot != std::string::npos)
        p.erase(dot);
    return p;
}

inline bool mkdirs(const std::string &path)
{
    std::string cur;
    for (size_t i = 0; i <= path.size(); i++) {
        if (i == path.size() || path[i] == '/') {
            if (!cur.empty() && cur != "/" && ::mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST)
                return false;
        }
        if (i < path.size())
            cur.push_back(path[i]);
    }
    return true;
}

inline bool is_dir(const std::string &p)
{
    struct stat st;
    return ::stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

// core::FormatHelper::readStringList (core/format_helper.cc:232-245): an OpenCV FileStorage XML
// file whose first top-level node is a sequence of strings (config_files/kitti_00_lst.xml:1-7:
// <opencv_storage><imagelist> "/image_0/000000.png" "/image_1/000000.png" ... ).  Entries are
// separated by white space; quotes are optional and allow embedded blanks.
inline bool read_string_list(const std::string &filename, std::vector<std::string> &l)
{
    l.clear();
    std::ifstream f(filename);
    if (!f)
        return false;
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string s = ss.str();
    size_t p = s.find("<opencv_storage>");
    if (p == std::string::npos)
        return false;
    p += 16;
    // first element below the root
    size_t open = std::string::npos;
    for (size_t i = p; i + 1 < s.size(); i++)
        if (s[i] == '<' && s[i + 1] != '!' && s[i + 1] != '?') {
            open = i;
            break;
        }
    if (open == std::string::npos || s[open + 1] == '/')
        return false;
    const size_t close = s.find('>', open);
    if (close == std::string::npos)
        return false;
    std::string name = s.substr(open + 1, close - open - 1);
    const size_t sp = name.find_first_of(" \t\r\n");
    if (sp != std::string::npos)
        name.erase(sp);
    const size_t end = s.find("</" + name + ">", close);
    if (end == std::string::npos)
        return false;
    const std::string body = s.substr(close + 1, end - close - 1);
    if (body.find('<') != std::string::npos)
        return false; // a map or nested node, not a plain sequence
    auto unescape = [](std::string t) {
        static const char *ent[][2] = {{"&quot;", "\""}, {"&apos;", "'"}, {"&lt;", "<"}, {"&gt;", ">"}, {"&amp;", "&"}};
        for (auto &e : ent)
            for (size_t q = t.find(e[0]); q != std::string::npos; q = t.find(e[0], q + 1))
                t.replace(q, std::strlen(e[0]), e[1]);
        return t;
    };
    size_t i = 0;
    while (i < body.size()) {
        while (i < body.size() && std::isspace((unsigned char)body[i]))
            i++;
        if (i >= body.size())
            break;
        size_t j;
        if (body[i] == '"') {
            j = body.find('"', i + 1);
            if (j == std::string::npos)
                return false;
            l.push_back(unescape(body.substr(i + 1, j - i - 1)));
            i = j + 1;
        } else {
            j = i;
            while (j < body.size() && !std::isspace((unsigned char)body[j]))
                j++;
            l.push_back(unescape(body.substr(i, j - i)));
            i = j;
        }
    }
    return true;
}

// boost::program_options config file as dense_stereo/main.cc:160-166 parses it with
// allow_unregistered: `key = value` lines, `#` starts a comment, blank lines ignored.
inline bool read_config_file(const std::string &filename, std::map<std::string, std::string> &kv)
{
    std::ifstream f(filename);
    if (!f)
        return false;
    std::string line;
    auto trim = [](std::string t) {
        const size_t a = t.find_first_not_of(" \t\r\n");
        if (a == std::string::npos)
            return std::string();
        return t.substr(a, t.find_last_not_of(" \t\r\n") - a + 1);
    };
    while (std::getline(f, line)) {
        const size_t hash = line.find('#');
        if (hash != std::string::npos)
            line.erase(hash);
        const size_t eq = line.find('=');
        if (eq == std::string::npos)
            continue;
        const std::string k = trim(line.substr(0, eq));
        if (!k.empty())
            kv[k] = trim(line.substr(eq + 1));
    }
    return true;
}

// cv::FileStorage(mat_fname, WRITE) << "disparity_matrix" << mat_disp (dense_stereo/main.cc:110-111)
// for a CV_16U matrix: OpenCV's YAML 1.0 layout (dt: w = unsigned 16-bit).
template <class MatT> inline bool write_yml_u16(const std::string &path, const std::string &name, const MatT &m)
{
    FILE *f = std::fopen(path.c_str(), "w");
    if (!f)
        return false;
    std::fprintf(f, "%%YAML:1.0\n%s: !!opencv-matrix\n   rows: %d\n   cols: %d\n   dt: w\n   data: [ ", name.c_str(), m.rows,
                 m.cols);
    int col = 11;
    const long n = (long)m.rows * m.cols;
    long k = 0;
    for (int y = 0; y < m.rows; y++) {
        const uint16_t *s = reinterpret_cast<const uint16_t *>(m.data + (size_t)y * (size_t)m.step);
        for (int x = 0; x < m.cols; x++, k++) {
            char tok[16];
            const int len = std::snprintf(tok, sizeof(tok), "%u%s", (unsigned)s[x], k + 1 < n ? "," : "");
            if (col + len + 1 > 80) {
                std::fputs("\n       ", f);
                col = 7;
            }
            std::fputs(tok, f);
            col += len;
            if (k + 1 < n) {
                std::fputc(' ', f);
                col++;
            }
        }
    }
    std::fputs(" ]\n", f);
    return std::fclose(f) == 0;
}

// FNV-style checksum of a disparity map, one update per uint16 pixel.
template <class MatT> inline uint64_t fnv_u16(const MatT &m)
{
    uint64_t h = 1469598103934665603ull;
    for (int y = 0; y < m.rows; y++) {
        const uint16_t *s = reinterpret_cast<const uint16_t *>(m.data + (size_t)y * (size_t)m.step);
        for (int x = 0; x < m.cols; x++)
            h = (h ^ s[x]) * 1099511628211ull;
    }
    return h;
}

// Contiguous slice [lo, hi) of n items for rank r of w: ceil(n / w) per rank.
inline void shard_range(long n, int r, int w, long *lo, long *hi)
{
    const long per = (n + w - 1) / w;
    *lo = r * per < n ? r * per : n;
    *hi = *lo + per < n ? *lo + per : n;
}

// Worker threads of a slice: started once (after the fork into per-GPU workers), they serve the decoders, the
// staging copies and the encoders alike, so the three stages share `threads` cores instead of each starting its
// own set per group, and a thread's codec buffers (png_io.hpp: scratch) live from image to image.
class Pool {
  public:
    explicit Pool(int threads)
    {
        for (int t = 0; t < std::max(1, threads); t++)
            th_.emplace_back([this]() { work(); });
    }
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> l(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_)
            t.join();
    }
    // f(0) .. f(n-1), handed to the workers one item at a time; returns when all are done and rethrows the first
    // exception.  Items of an `urgent` job (the staging copies the engine call is waiting for) are taken before
    // those of the jobs already queued.
    template <class F> void run(int n, F f, bool urgent = false)
    {
        if (n <= 0)
            return;
        Job job;
        job.n = n;
        job.f = [&f](int i) { f(i); };
        {
            std::lock_guard<std::mutex> l(m_);
            if (urgent)
                q_.push_front(&job);
            else
                q_.push_back(&job);
        }
        cv_.notify_all();
        std::unique_lock<std::mutex> l(m_);
        job.finished.wait(l, [&job]() { return job.done == job.n; });
        if (!job.err.empty())
            throw std::runtime_error(job.err);
    }

  private:
    struct Job {
        std::function<void(int)> f;
        int n = 0, next = 0, done = 0; // guarded by the pool's mutex
        std::string err;
        std::condition_variable finished;
    };
    void work()
    {
        std::unique_lock<std::mutex> l(m_);
        for (;;) {
            cv_.wait(l, [this]() { return stop_ || !q_.empty(); });
            if (q_.empty())
                return;
            Job *job = q_.front();
            const int i = job->next++;
            if (job->next >= job->n)
                q_.pop_front(); // all items handed out; run() still waits for them to finish
            const bool skip = !job->err.empty();
            l.unlock();
            std::string err;
            if (!skip) {
                try {
                    job->f(i);
                } catch (const std::exception &e) {
                    err = e.what();
                }
            }
            l.lock();
            if (!err.empty() && job->err.empty())
                job->err = err;
            if (++job->done == job->n)
                job->finished.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Job *> q_;
    bool stop_ = false;
};

} // namespace host
} // namespace sgmx
#endif
