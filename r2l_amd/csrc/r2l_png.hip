// PNG writer pool: host threads that encode 8-bit frames to .png files while the GPU renders the next ones.
//
// Replaces, in the test-set evaluation loop, the reference's inline `imageio.imwrite(filename, to8b(rgb))` of every
// prediction and ground-truth frame (/root/reference/main.py:337-344, render_path).  At MI355X speed a 400x400 frame is
// rendered in 4.2 ms; encoding its two PNGs in Python (PIL: ~15 ms each, the GIL held around zlib) made the test-set loop
// encoder-bound (6.5 ms/frame with 12 Python threads, profiles/r03_e2e_render.txt).  Here a job is (path, pixels in a
// caller-owned pinned buffer, an optional HIP event to wait for: the frame's asynchronous device-to-host copy); a worker
// waits for the event, filters the scanlines (Sub), deflates them with zlib and writes the file — no GIL, no Python objects.
// Pixels are exactly the bytes handed over (PNG is lossless); only the compressed representation differs from imageio's.
// Host code only (no kernels): it lives in libr2l_hip.so so that the drop-in is one file, like the ray-shard reader.
#include "r2l_common.h"

#include <zlib.h>

#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct PngJob {
    int64_t id;
    std::string path;
    const unsigned char* pixels;
    int H, W, C;
    hipEvent_t ready;  // or nullptr
};

void put_be32(unsigned char* p, uint32_t v) {
    p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v;
}
void chunk(std::vector<unsigned char>& out, const char* type, const unsigned char* data, size_t n) {
    const size_t at = out.size();
    out.resize(at + 12 + n);
    put_be32(&out[at], (uint32_t)n);
    memcpy(&out[at + 4], type, 4);
    if (n) memcpy(&out[at + 8], data, n);
    put_be32(&out[at + 8 + n], (uint32_t)crc32(crc32(0L, Z_NULL, 0), &out[at + 4], (uInt)(n + 4)));
}

// "" on success, else what went wrong
std::string encode_png(const PngJob& j, int level, std::vector<unsigned char>& raw, std::vector<unsigned char>& z,
                       std::vector<unsigned char>& file) {
    const int bpp = j.C;
    const size_t row = (size_t)j.W * bpp;
    raw.resize((row + 1) * (size_t)j.H);
    for (int y = 0; y < j.H; ++y) {  // filter type 1 (Sub): byte - byte of the pixel to the left
        const unsigned char* s = j.pixels + (size_t)y * row;
        unsigned char* d = &raw[(size_t)y * (row + 1)];
        d[0] = 1;
        for (int k = 0; k < bpp && (size_t)k < row; ++k) d[1 + k] = s[k];
        for (size_t k = bpp; k < row; ++k) d[1 + k] = (unsigned char)(s[k] - s[k - bpp]);
    }
    uLongf zn = compressBound((uLong)raw.size());
    z.resize(zn);
    if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), level) != Z_OK) return j.path + ": zlib compress2 failed";
    file.clear();
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    file.insert(file.end(), sig, sig + 8);
    unsigned char ihdr[13];
    put_be32(ihdr, (uint32_t)j.W);
    put_be32(ihdr + 4, (uint32_t)j.H);
    ihdr[8] = 8;                                               // bit depth
    ihdr[9] = (unsigned char)(j.C == 1 ? 0 : (j.C == 3 ? 2 : 6));  // grey / RGB / RGBA
    ihdr[10] = ihdr[11] = ihdr[12] = 0;
    chunk(file, "IHDR", ihdr, 13);
    chunk(file, "IDAT", z.data(), (size_t)zn);
    chunk(file, "IEND", nullptr, 0);
    FILE* f = fopen(j.path.c_str(), "wb");
    if (!f) return j.path + ": cannot open for writing";
    const bool ok = fwrite(file.data(), 1, file.size(), f) == file.size();
    if (fclose(f) != 0 || !ok) return j.path + ": short write";
    return "";
}

}  // namespace

struct r2l_png_writer {
    int level = 1;
    int device = 0;
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::deque<PngJob> queue;
    int64_t next_id = 0;
    int64_t done_below = 0;        // every id < done_below is finished (ids finish out of order: see finished)
    std::vector<int64_t> finished; // finished ids >= done_below
    std::string error;             // first failure
    bool stop = false;

    void worker() {
        (void)hipSetDevice(device);  // hipEventSynchronize on the opener's device
        std::vector<unsigned char> raw, z, file;
        for (;;) {
            PngJob j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return stop || !queue.empty(); });
                if (queue.empty()) return;
                j = queue.front();
                queue.pop_front();
            }
            std::string err;
            if (j.ready != nullptr && hipEventSynchronize(j.ready) != hipSuccess) err = j.path + ": hipEventSynchronize failed";
            if (err.empty()) err = encode_png(j, level, raw, z, file);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!err.empty() && error.empty()) error = err;
                finished.push_back(j.id);
                for (bool moved = true; moved;) {  // advance the low-water mark
                    moved = false;
                    for (size_t i = 0; i < finished.size(); ++i)
                        if (finished[i] == done_below) {
                            finished[i] = finished.back();
                            finished.pop_back();
                            ++done_below;
                            moved = true;
                            break;
                        }
                }
            }
            cv_done.notify_all();
        }
    }
};

extern "C" int r2l_png_writer_open(int n_threads, int level, r2l_png_writer** out) {
    R2L_REQUIRE(out != nullptr && n_threads >= 1 && n_threads <= 256 && level >= 0 && level <= 9,
                "r2l_png_writer_open: out is NULL, or n_threads (1..256) / zlib level (0..9) out of range");
    r2l_png_writer* w = new r2l_png_writer();
    w->level = level;
    if (hipGetDevice(&w->device) != hipSuccess) w->device = 0;
    for (int i = 0; i < n_threads; ++i) w->threads.emplace_back([w] { w->worker(); });
    *out = w;
    return 0;
}

extern "C" int r2l_png_writer_submit(r2l_png_writer* w, const char* path, const unsigned char* pixels, int H, int W, int C,
                                     void* ready_event, int64_t* job_id) {
    R2L_REQUIRE(w && path && pixels && H > 0 && W > 0 && (C == 1 || C == 3 || C == 4),
                "r2l_png_writer_submit: NULL writer / path / pixels, or H, W <= 0, or C not in {1, 3, 4}");
    {
        std::lock_guard<std::mutex> lk(w->mu);
        PngJob j{w->next_id++, path, pixels, H, W, C, (hipEvent_t)ready_event};
        if (job_id) *job_id = j.id;
        w->queue.push_back(std::move(j));
    }
    w->cv_job.notify_one();
    return 0;
}

// Blocks until job `job_id` (and every earlier one) is on disk; job_id < 0: every job submitted so far.  Returns non-zero if
// any job failed since the last wait that reported a failure (r2l_last_error: the first of them) — the error is handed over
// ONCE and cleared, so one bad path does not fail every later evaluation of a long-lived shared writer.
extern "C" int r2l_png_writer_wait(r2l_png_writer* w, int64_t job_id) {
    R2L_REQUIRE(w != nullptr, "r2l_png_writer_wait: NULL writer");
    std::unique_lock<std::mutex> lk(w->mu);
    const int64_t upto = (job_id < 0 || job_id >= w->next_id) ? w->next_id : job_id + 1;  // (an id not handed out yet: everything so far)
    w->cv_done.wait(lk, [&] { return w->done_below >= upto; });
    if (!w->error.empty()) {
        r2l_set_error_msg(w->error.c_str());
        w->error.clear();
        return (int)hipErrorUnknown;
    }
    return 0;
}

extern "C" int r2l_png_writer_close(r2l_png_writer* w) {
    if (w == nullptr) return 0;
    int rc = r2l_png_writer_wait(w, -1);
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->stop = true;
    }
    w->cv_job.notify_all();
    for (auto& t : w->threads) t.join();
    delete w;
    return rc;
}
