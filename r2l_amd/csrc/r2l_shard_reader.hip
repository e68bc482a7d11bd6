// Ray-shard reader: a host thread pool that streams `.npy` ray shards straight into caller-owned pinned buffers.
//
// Replaces, for --data_mode rays, the reference's DataLoader stack: BlenderDataset_v2.__getitem__ (np.load of one
// [4096,9] f32 shard, dataset/load_blender.py:257-324), InfiniteSamplerWrapper (random permutations of the file
// list forever, main.py:759-776) and the batch_size=N_rand collate + pin_memory of main.py:794-806.  One training
// step at MI355X speed consumes 20 shards (2.95 MB) every 25 ms per GPU; Python worker processes, pickled tensors and
// a fresh pinned allocation per batch are the wrong tools for that, a few pread() threads writing into a ring of
// pinned slots are enough.  Host code only (no kernels) -- it lives in libr2l_hip.so so that the drop-in is one file.
//
// On-disk format (kept): NumPy .npy v1/v2/v3, descr '<f4', C order, shape (rows, 9) -- rows [o(3), d(3), rgb(3)].
#include "r2l_common.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct NpyInfo {
    int64_t rows = 0, cols = 0, data_offset = 0;
};

// Parses the header of a .npy file; returns "" on success or a message.
std::string npy_parse(int fd, const std::string& path, NpyInfo* out) {
    unsigned char head[12];
    if (pread(fd, head, 12, 0) != 12 || memcmp(head, "\x93NUMPY", 6) != 0) return path + ": not a .npy file";
    const int major = head[6];
    int64_t hlen, hoff;
    if (major == 1) {
        hlen = head[8] | (head[9] << 8);
        hoff = 10;
    } else if (major == 2 || major == 3) {
        hlen = (int64_t)head[8] | ((int64_t)head[9] << 8) | ((int64_t)head[10] << 16) | ((int64_t)head[11] << 24);
        hoff = 12;
    } else {
        return path + ": unsupported .npy version";
    }
    if (hlen <= 0 || hlen > 65536) return path + ": bad .npy header length";
    std::string h((size_t)hlen, '\0');
    if (pread(fd, &h[0], (size_t)hlen, hoff) != hlen) return path + ": truncated .npy header";
    auto value_of = [&](const char* key) -> std::string {
        size_t k = h.find(key);
        if (k == std::string::npos) return "";
        k = h.find(':', k);
        if (k == std::string::npos) return "";
        size_t b = h.find_first_not_of(" ", k + 1);
        return b == std::string::npos ? "" : h.substr(b);
    };
    const std::string descr = value_of("'descr'");
    if (descr.compare(0, 5, "'<f4'") != 0 && descr.compare(0, 5, "'|f4'") != 0)
        return path + ": dtype must be little-endian float32";
    if (value_of("'fortran_order'").compare(0, 5, "False") != 0) return path + ": must be C-ordered";
    const std::string shape = value_of("'shape'");
    long long r = 0, c = 0;
    if (sscanf(shape.c_str(), "(%lld, %lld)", &r, &c) != 2 || r <= 0 || c <= 0)
        return path + ": shape must be (rows, cols)";
    out->rows = r;
    out->cols = c;
    out->data_offset = hoff + hlen;
    return "";
}

struct Rng {  // splitmix64: enough for shuffling file orders, seedable per rank
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint64_t below(uint64_t n) {  // unbiased
        const uint64_t lim = UINT64_MAX - UINT64_MAX % n;
        uint64_t v;
        do v = next(); while (v >= lim);
        return v % n;
    }
};

struct Slot {
    char* base = nullptr;
    int pending = 0;       // reads outstanding
    bool ready = false;    // all reads landed, not yet handed out
    std::string error;
};

struct Job {
    int slot;
    int pos;               // position of the shard inside the batch
    int64_t file;
};

}  // namespace

struct r2l_reader {
    std::vector<std::string> paths;
    int files_per_batch = 0, depth = 0;
    int64_t rows = 0, cols = 0, shard_bytes = 0;
    Rng rng{0};
    std::vector<int64_t> order;
    int64_t cursor = 0;
    std::vector<Slot> slots;
    std::deque<int> filling;           // slots in fill order (front = next to hand out)
    std::deque<Job> jobs;
    std::mutex mu;
    std::condition_variable cv_jobs, cv_ready;
    std::vector<std::thread> workers;
    bool stop = false;
    int64_t files_read = 0;

    int64_t next_file() {              // InfiniteSampler (main.py:759-767): a fresh permutation whenever one is used up
        if (cursor == (int64_t)order.size()) {
            for (int64_t i = (int64_t)order.size() - 1; i > 0; --i) std::swap(order[i], order[rng.below((uint64_t)i + 1)]);
            cursor = 0;
        }
        return order[cursor++];
    }
    void schedule(int slot) {          // caller holds mu
        Slot& s = slots[slot];
        s.pending = files_per_batch;
        s.ready = false;
        s.error.clear();
        filling.push_back(slot);
        for (int p = 0; p < files_per_batch; ++p) jobs.push_back(Job{slot, p, next_file()});
        cv_jobs.notify_all();
    }
    void work() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_jobs.wait(lk, [&] { return stop || !jobs.empty(); });
                if (stop) return;
                j = jobs.front();
                jobs.pop_front();
            }
            std::string err;
            const std::string& path = paths[(size_t)j.file];
            int fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
            if (fd < 0) {
                err = path + ": " + strerror(errno);
            } else {
                NpyInfo info;
                err = npy_parse(fd, path, &info);
                if (err.empty() && (info.rows != rows || info.cols != cols)) {
                    char b[128];
                    snprintf(b, sizeof b, ": shape (%lld, %lld), expected (%lld, %lld)", (long long)info.rows,
                             (long long)info.cols, (long long)rows, (long long)cols);
                    err = path + b;
                }
                if (err.empty()) {
                    char* dst = slots[j.slot].base + (int64_t)j.pos * shard_bytes;
                    int64_t done = 0;
                    while (done < shard_bytes) {
                        ssize_t n = pread(fd, dst + done, (size_t)(shard_bytes - done), info.data_offset + done);
                        if (n <= 0) {
                            err = path + ": short read";
                            break;
                        }
                        done += n;
                    }
                }
                close(fd);
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                Slot& s = slots[j.slot];
                if (!err.empty() && s.error.empty()) s.error = err;
                ++files_read;
                if (--s.pending == 0) {
                    s.ready = true;
                    cv_ready.notify_all();
                }
            }
        }
    }
};

extern "C" {

int r2l_npy_shape(const char* path, int64_t* rows, int64_t* cols) {
    int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) {
        r2l_set_error_msg((std::string(path) + ": " + strerror(errno)).c_str());
        return 1;
    }
    NpyInfo info;
    std::string err = npy_parse(fd, path, &info);
    close(fd);
    if (!err.empty()) {
        r2l_set_error_msg(err.c_str());
        return 1;
    }
    *rows = info.rows;
    *cols = info.cols;
    return 0;
}

int r2l_reader_open(const char* const* paths, int64_t n_paths, int files_per_batch, int n_threads, uint64_t seed,
                    void* const* slot_ptrs, int depth, r2l_reader** out) {
    if (n_paths <= 0 || files_per_batch <= 0 || n_threads <= 0 || depth < 2 || !paths || !slot_ptrs || !out) {
        r2l_set_error_msg("r2l_reader_open: bad arguments (need >= 1 path, >= 1 thread, >= 2 slots)");
        return 1;
    }
    int64_t rows, cols;
    if (r2l_npy_shape(paths[0], &rows, &cols)) return 1;
    r2l_reader* r = new r2l_reader;
    r->paths.assign(paths, paths + n_paths);
    r->files_per_batch = files_per_batch;
    r->depth = depth;
    r->rows = rows;
    r->cols = cols;
    r->shard_bytes = rows * cols * 4;
    r->rng.s = seed;
    r->order.resize((size_t)n_paths);
    for (int64_t i = 0; i < n_paths; ++i) r->order[(size_t)i] = i;
    r->cursor = n_paths;  // forces the first shuffle
    r->slots.resize((size_t)depth);
    for (int i = 0; i < depth; ++i) r->slots[(size_t)i].base = (char*)slot_ptrs[i];
    {
        std::lock_guard<std::mutex> lk(r->mu);
        for (int i = 0; i < depth; ++i) r->schedule(i);
    }
    for (int i = 0; i < n_threads; ++i) r->workers.emplace_back([r] { r->work(); });
    *out = r;
    return 0;
}

int r2l_reader_info(r2l_reader* r, int64_t* rows, int64_t* cols, int64_t* files_read) {
    std::lock_guard<std::mutex> lk(r->mu);
    if (rows) *rows = r->rows;
    if (cols) *cols = r->cols;
    if (files_read) *files_read = r->files_read;
    return 0;
}

/* Blocks until the oldest scheduled batch has landed; *slot = index of the pinned slot that holds it
 * ([files_per_batch * rows, cols] f32).  The slot stays untouched until r2l_reader_release(slot). */
int r2l_reader_next(r2l_reader* r, int* slot) {
    std::unique_lock<std::mutex> lk(r->mu);
    if (r->filling.empty()) {
        r2l_set_error_msg("r2l_reader_next: every slot is checked out (release one first)");
        return 1;
    }
    const int s = r->filling.front();
    r->cv_ready.wait(lk, [&] { return r->slots[(size_t)s].ready; });
    r->filling.pop_front();
    r->slots[(size_t)s].ready = false;
    if (!r->slots[(size_t)s].error.empty()) {
        r2l_set_error_msg(r->slots[(size_t)s].error.c_str());
        r->schedule(s);  // keep the ring alive; the caller sees the error for this batch
        return 1;
    }
    *slot = s;
    return 0;
}

int r2l_reader_release(r2l_reader* r, int slot) {
    std::lock_guard<std::mutex> lk(r->mu);
    if (slot < 0 || slot >= r->depth) {
        r2l_set_error_msg("r2l_reader_release: bad slot");
        return 1;
    }
    r->schedule(slot);
    return 0;
}

int r2l_reader_close(r2l_reader* r) {
    if (!r) return 0;
    {
        std::lock_guard<std::mutex> lk(r->mu);
        r->stop = true;
        r->cv_jobs.notify_all();
    }
    for (auto& t : r->workers) t.join();
    delete r;
    return 0;
}

}  // extern "C"
