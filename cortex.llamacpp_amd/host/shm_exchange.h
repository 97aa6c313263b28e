// shm_exchange.h — the exchange step of a row split between PROCESSES of one machine through a shared-memory segment: the transport the engine's own row split
// (tp_split.cc, `"split_mode": "row"`) uses where its ranks cannot each own a GPU (RCCL refuses two ranks on one device) - validation rigs, this pool's one-GPU
// boxes.  It implements tp_host_exchange_fn (tp_comm.h): op 0 = sum in place over n floats, op 1 = all-gather of n floats per rank.  Ranks that own their
// devices exchange over RCCL / xGMI instead and never come here.
//
// Layout: [Header | P slots of `cap` floats].  An exchange moves the message in pieces of at most `cap` floats: every rank copies its piece into its slot,
// barrier, every rank reads all slots (sum in rank order: the same bits on every rank), barrier.  The barrier is a generation counter in the segment; a rank
// that waits longer than the bound, or sees the `dead` word (set by whoever gave up, or by rank 0 when a worker process has exited), leaves with an error -
// nobody waits for ever, and the step that failed names why.  Plain C++ (no HIP): tests/host/shm_exchange_test.cc runs it between forked processes on CPU.
#pragma once

#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <functional>
#include <new>
#include <string>

namespace mi355 {

class ShmExchange {
  public:
    struct Header {
        std::atomic<uint32_t> arrive, gen, dead;
        uint32_t size;
        uint64_t cap;
    };
    ~ShmExchange() { close(); }
    // rank 0: a new anonymous segment (memfd: the descriptor survives exec and is handed to the workers by number)
    bool create(int size, size_t cap_floats, std::string &err) {
        size_ = size; rank_ = 0; cap_ = cap_floats;
        bytes_ = sizeof(Header) + 64 + (size_t)size * cap_floats * sizeof(float);
        fd_ = (int)memfd_create("mi355_row_split", 0);
        if (fd_ < 0 || ftruncate(fd_, (off_t)bytes_) != 0) { err = "row split: cannot create the shared exchange segment"; return false; }
        if (!map(err)) return false;
        new (hdr()) Header();
        hdr()->arrive.store(0); hdr()->gen.store(0); hdr()->dead.store(0); hdr()->size = (uint32_t)size; hdr()->cap = cap_floats;
        return true;
    }
    // a worker: the segment rank 0 made
    bool attach(int fd, int rank, int size, size_t cap_floats, std::string &err) {
        fd_ = fd; rank_ = rank; size_ = size; cap_ = cap_floats;
        bytes_ = sizeof(Header) + 64 + (size_t)size * cap_floats * sizeof(float);
        if (!map(err)) return false;
        if (hdr()->size != (uint32_t)size || hdr()->cap != cap_floats) { err = "row split: the exchange segment has another geometry"; return false; }
        return true;
    }
    void close() {
        if (base_) munmap(base_, bytes_);
        base_ = nullptr;
        if (fd_ >= 0) ::close(fd_);
        fd_ = -1;
    }
    int fd() const { return fd_; }
    int rank() const { return rank_; }
    void set_timeout_ms(long ms) { timeout_ms_ = ms; }
    void set_liveness(std::function<bool()> f) { alive_ = std::move(f); }     // rank 0: "are the workers still running" (checked while it waits)
    void mark_dead() { if (base_) hdr()->dead.store(1); }
    bool dead() const { return base_ && hdr()->dead.load() != 0; }
    const std::string &error() const { return err_; }

    // tp_host_exchange_fn: 0 on success
    int exchange(float *buf, size_t n, int op) {
        if (!base_) return -1;
        for (size_t off = 0; off < n; off += cap_) {
            const size_t m = n - off < cap_ ? n - off : cap_;
            float *mine = slot(rank_);
            std::memcpy(mine, op == 0 ? buf + off : buf + (size_t)rank_ * n + off, m * sizeof(float));
            if (!barrier()) return -1;
            if (op == 0) {
                float *dst = buf + off;
                const float *s0 = slot(0);
                for (size_t i = 0; i < m; i++) dst[i] = s0[i];
                for (int q = 1; q < size_; q++) { const float *s = slot(q); for (size_t i = 0; i < m; i++) dst[i] += s[i]; }
            } else {
                for (int q = 0; q < size_; q++) if (q != rank_) std::memcpy(buf + (size_t)q * n + off, slot(q), m * sizeof(float));
            }
            if (!barrier()) return -1;
        }
        return 0;
    }
    static int callback(void *user, float *buf, size_t n, int op) { return static_cast<ShmExchange *>(user)->exchange(buf, n, op); }

  private:
    bool map(std::string &err) {
        void *p = mmap(nullptr, bytes_, PROT_READ | PROT_WRITE, MAP_SHARED, fd_, 0);
        if (p == MAP_FAILED) { err = "row split: cannot map the shared exchange segment"; return false; }
        base_ = static_cast<uint8_t *>(p);
        return true;
    }
    Header *hdr() const { return reinterpret_cast<Header *>(base_); }
    float *slot(int q) const { return reinterpret_cast<float *>(base_ + sizeof(Header) + 64) + (size_t)q * cap_; }
    bool barrier() {
        Header *h = hdr();
        if (h->dead.load()) { err_ = "a rank of the row split has given up or exited"; return false; }
        const uint32_t g = h->gen.load(std::memory_order_acquire);
        if (h->arrive.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)size_) {
            h->arrive.store(0, std::memory_order_relaxed);
            h->gen.store(g + 1, std::memory_order_release);
            return true;
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0; h->gen.load(std::memory_order_acquire) == g; spins++) {
            if (h->dead.load()) { err_ = "a rank of the row split has given up or exited"; return false; }
            if (spins < 2000) continue;                      // a peer that is one memcpy behind
            sched_yield();
            if ((spins & 1023) == 0) {
                if (alive_ && !alive_()) { h->dead.store(1); err_ = "a worker process of the row split has exited"; return false; }
                const long ms = (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
                if (ms > timeout_ms_) { h->dead.store(1); err_ = "rank " + std::to_string(rank_) + " waited " + std::to_string(ms / 1000) + " s for the other ranks at an exchange"; return false; }
            }
        }
        return true;
    }
    int fd_ = -1, rank_ = 0, size_ = 1;
    size_t cap_ = 0, bytes_ = 0;
    uint8_t *base_ = nullptr;
    long timeout_ms_ = 120000;
    std::function<bool()> alive_;
    std::string err_;
};

}  // namespace mi355
