// tp_split.cc — see tp_split.h.  Reference: the engine process drives every visible device from one /loadmodel (src/llama_engine.cc:609-611, n_gpu_layers; no
// split key at :553-658 - SURVEY.md §2b proposes split_mode / tensor_split / main_gpu).  Keys read here:
//   split_mode   "row"      the model's rows / columns cut over the ranks; anything else: this file is not involved
//   tensor_split [..]       llama.cpp's per-device proportions; only EVEN splits exist here (cuts fall on head and 256-element boundaries): its non-zero
//                           entries count the ranks, unequal entries are refused
//   split_ranks  N          (new) the number of ranks, when it is not the number of visible devices - more ranks than devices makes ranks SHARE devices and
//                           exchange through shared memory: a validation rig, not a deployment
//   main_gpu     d          rank 0's device; rank r takes device (d + r) mod visible devices
#include "tp_split.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <poll.h>
#include <signal.h>
#include <spawn.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "hip_backend.h"
#include "log.h"
#include "shm_exchange.h"
#include "tp_comm.h"

extern char **environ;

namespace mi355 {
namespace {

enum Cmd : uint32_t { C_INIT = 1, C_DECODE, C_KV_CLEAR, C_KV_SEQ_RM, C_KV_SEQ_ADD, C_KV_SEQ_CP, C_SET_EMBD, C_QUIT, C_REPLY, C_DECODE_EMBD, C_LOAD };

bool write_all(int fd, const void *p, size_t n) {
    const uint8_t *b = static_cast<const uint8_t *>(p);
    while (n > 0) {
        const ssize_t k = ::send(fd, b, n, MSG_NOSIGNAL);
        if (k < 0) { if (errno == EINTR) continue; return false; }
        b += k; n -= (size_t)k;
    }
    return true;
}
// false on end of stream, error, or when `timeout_ms` (< 0: none) runs out before n bytes have come
bool read_all(int fd, void *p, size_t n, long timeout_ms) {
    uint8_t *b = static_cast<uint8_t *>(p);
    const auto t0 = std::chrono::steady_clock::now();
    while (n > 0) {
        if (timeout_ms >= 0) {
            const long left = timeout_ms - (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
            if (left <= 0) return false;
            pollfd pf{fd, POLLIN, 0};
            const int r = ::poll(&pf, 1, (int)std::min<long>(left, 1000));
            if (r < 0) { if (errno == EINTR) continue; return false; }
            if (r == 0) continue;
        }
        const ssize_t k = ::recv(fd, b, n, 0);
        if (k == 0) return false;
        if (k < 0) { if (errno == EINTR) continue; return false; }
        b += k; n -= (size_t)k;
    }
    return true;
}
bool send_msg(int fd, uint32_t type, const void *payload, size_t len) {
    const uint32_t h[2] = {type, (uint32_t)len};
    return write_all(fd, h, sizeof h) && (len == 0 || write_all(fd, payload, len));
}
bool recv_msg(int fd, uint32_t &type, std::vector<uint8_t> &payload, long timeout_ms) {
    uint32_t h[2];
    if (!read_all(fd, h, sizeof h, timeout_ms)) return false;
    if (h[1] > (1u << 30)) return false;
    type = h[0];
    payload.resize(h[1]);
    return h[1] == 0 || read_all(fd, payload.data(), h[1], timeout_ms);
}
std::string hex(const uint8_t *p, size_t n) {
    static const char *d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; i++) { s += d[p[i] >> 4]; s += d[p[i] & 15]; }
    return s;
}
bool unhex(const std::string &s, std::vector<uint8_t> &out) {
    if (s.size() & 1) return false;
    out.resize(s.size() / 2);
    auto v = [](char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : -1; };
    for (size_t i = 0; i < out.size(); i++) { const int a = v(s[2 * i]), b = v(s[2 * i + 1]); if (a < 0 || b < 0) return false; out[i] = (uint8_t)(a * 16 + b); }
    return true;
}
long step_timeout_ms() {
    const char *e = getenv("MI355_TP_STEP_TIMEOUT_S");
    const long s = e ? atol(e) : 120;
    return (s > 0 ? s : 120) * 1000;
}
// bin/mi355_tp_worker beside lib/ (where this library was loaded from), or what MI355_TP_WORKER names
std::string worker_path() {
    if (const char *e = getenv("MI355_TP_WORKER")) return e;
    Dl_info di;
    if (dladdr(reinterpret_cast<void *>(&tp_split_worker_main), &di) && di.dli_fname) {
        std::string p = di.dli_fname;
        const size_t sl = p.find_last_of('/');
        const std::string dir = sl == std::string::npos ? "." : p.substr(0, sl);
        return dir + "/../bin/mi355_tp_worker";
    }
    return "mi355_tp_worker";
}

struct Worker {
    pid_t pid = -1;
    int fd = -1, rank = 0;
    bool exited = false;
    int status = 0;
};

class SplitBackend : public IBackend {
  public:
    SplitBackend() : timeout_ms_(step_timeout_ms()) {}
    ~SplitBackend() override { shutdown(); }

    std::unique_ptr<IBackend> local;
    std::vector<Worker> workers;
    std::unique_ptr<ShmExchange> shm;

    int n_ctx() const override { return local->n_ctx(); }
    int n_batch() const override { return local->n_batch(); }
    int n_ubatch() const override { return local->n_ubatch(); }
    int n_vocab() const override { return local->n_vocab(); }
    int n_embd() const override { return local->n_embd(); }
    const Vocab &vocab() const override { return local->vocab(); }
    const char *last_error() const override { return !err_.empty() ? err_.c_str() : local->last_error(); }

    int decode(const BatchView &b) override {
        if (dead_) return -1;
        // the batch to every worker first, then this rank's own launch: the exchange kernels of the ranks meet on the devices
        const size_t n = (size_t)b.n_tokens;
        msg_.resize(4 + n * 13);
        uint8_t *p = msg_.data();
        const int32_t nt = b.n_tokens;
        memcpy(p, &nt, 4); p += 4;
        memcpy(p, b.token, n * 4); p += n * 4;
        memcpy(p, b.pos, n * 4); p += n * 4;
        memcpy(p, b.seq_id, n * 4); p += n * 4;
        memcpy(p, b.logits, n);
        if (!broadcast(C_DECODE, msg_.data(), msg_.size())) return -1;
        const int rc = local->decode(b);
        if (shm && shm->dead()) { fail("row split: " + (shm->error().empty() ? std::string("an exchange was abandoned") : shm->error()) + describe_exits()); return -1; }
        std::vector<int32_t> rcs;
        if (!collect(rcs)) return -1;
        for (size_t i = 0; i < rcs.size(); i++)
            if (rcs[i] != rc) { fail("row split: rank " + std::to_string(workers[i].rank) + " answered a batch with " + std::to_string(rcs[i]) + ", rank 0 with " + std::to_string(rc)); return -1; }
        return rc;
    }
    // LLaVA over the split: the projector file is rank 0's (the image tower is one pass of small launches per picture, nothing of it is on the token path); the
    // picture's embedding rows - replicated rows of the residual stream, like every rank's copy of an embedding-table row - travel to the workers with the batch
    bool multimodal() const override { return local->multimodal(); }
    bool image_check(const uint8_t *bytes, size_t n, std::string &err) override { return local->image_check(bytes, n, err); }
    int image_embed(const uint8_t *bytes, size_t n, std::vector<float> &rows, std::string &err) override { return dead_ ? -1 : local->image_embed(bytes, n, rows, err); }
    int decode_embd(const float *rows, int n, int pos0, int seq) override {
        if (dead_ || n <= 0) return -1;
        const size_t E = (size_t)local->n_embd();
        msg_.resize(12 + (size_t)n * E * 4);
        const int32_t h[3] = {n, pos0, seq};
        memcpy(msg_.data(), h, 12);
        memcpy(msg_.data() + 12, rows, (size_t)n * E * 4);
        if (!broadcast(C_DECODE_EMBD, msg_.data(), msg_.size())) return -1;
        const int rc = local->decode_embd(rows, n, pos0, seq);
        if (shm && shm->dead()) { fail("row split: " + (shm->error().empty() ? std::string("an exchange was abandoned") : shm->error()) + describe_exits()); return -1; }
        std::vector<int32_t> rcs;
        if (!collect(rcs)) return -1;
        for (size_t i = 0; i < rcs.size(); i++)
            if (rcs[i] != rc) { fail("row split: rank " + std::to_string(workers[i].rank) + " answered an embeddings batch with " + std::to_string(rcs[i]) + ", rank 0 with " + std::to_string(rc)); return -1; }
        return rc;
    }
    const float *logits_ith(int i) override { return dead_ ? nullptr : local->logits_ith(i); }
    int argmax_ith(int i) override { return dead_ ? -1 : local->argmax_ith(i); }
    int topk_ith(int i, int k, const std::vector<int32_t> &t, const std::vector<float> &bi, const std::vector<int32_t> &c, float r, float f, float pr, int32_t *toks, float *lg) override {
        return dead_ ? -1 : local->topk_ith(i, k, t, bi, c, r, f, pr, toks, lg);
    }
    void topk_batch(std::vector<TopkRequest> &reqs) override { if (!dead_) local->topk_batch(reqs); else for (auto &q : reqs) q.ok = false; }
    int topk_max_k() const override { return local->topk_max_k(); }
    int topk_max_adj() const override { return local->topk_max_adj(); }
    int pooling_type() const override { return local->pooling_type(); }
    bool is_encoder() const override { return local->is_encoder(); }
    void set_embeddings(bool on) override {
        const int32_t v = on ? 1 : 0;
        if (!dead_ && broadcast(C_SET_EMBD, &v, 4)) { local->set_embeddings(on); std::vector<int32_t> r; collect(r); }
    }
    const float *embeddings_ith(int i) override { return dead_ ? nullptr : local->embeddings_ith(i); }
    void kv_clear() override {
        if (!dead_ && broadcast(C_KV_CLEAR, nullptr, 0)) { local->kv_clear(); std::vector<int32_t> r; collect(r); }
    }
    bool kv_seq_rm(int seq, int p0, int p1) override {
        const int32_t a[3] = {seq, p0, p1};
        if (dead_ || !broadcast(C_KV_SEQ_RM, a, sizeof a)) return false;
        const bool ok = local->kv_seq_rm(seq, p0, p1);
        std::vector<int32_t> r;
        if (!collect(r)) return false;
        for (size_t i = 0; i < r.size(); i++) if ((r[i] != 0) != ok) { fail("row split: rank " + std::to_string(workers[i].rank) + " disagrees about a KV removal"); return false; }
        return ok;
    }
    void kv_seq_add(int seq, int p0, int p1, int delta) override {
        const int32_t a[4] = {seq, p0, p1, delta};
        if (!dead_ && broadcast(C_KV_SEQ_ADD, a, sizeof a)) { local->kv_seq_add(seq, p0, p1, delta); std::vector<int32_t> r; collect(r); }
    }
    void kv_seq_cp(int src, int dst, int p0, int p1) override {
        const int32_t a[4] = {src, dst, p0, p1};
        if (!dead_ && broadcast(C_KV_SEQ_CP, a, sizeof a)) { local->kv_seq_cp(src, dst, p0, p1); std::vector<int32_t> r; collect(r); }
    }

    // a worker that has exited (reaped here): checked while this rank waits at an exchange, and when a reply does not come
    bool workers_alive() {
        bool all = true;
        for (auto &w : workers) {
            if (w.exited) { all = false; continue; }
            int st = 0;
            if (w.pid > 0 && waitpid(w.pid, &st, WNOHANG) == w.pid) { w.exited = true; w.status = st; all = false; }
        }
        return all;
    }
    void fail(const std::string &why) {
        if (dead_) return;
        dead_ = true;
        err_ = why;
        log_line(LOG_ERROR, "%s", why.c_str());
        if (shm) shm->mark_dead();
        else tp_abort();                                // (RCCL: what this rank has queued must not wait for a peer that is gone)
        for (auto &w : workers) if (!w.exited && w.pid > 0) kill(w.pid, SIGKILL);
    }
    bool collect(std::vector<int32_t> &rcs) {          // one reply per worker, bounded; a missing one fails the backend and names the rank
        rcs.clear();
        for (auto &w : workers) {
            uint32_t type = 0;
            std::vector<uint8_t> pl;
            const auto t0 = std::chrono::steady_clock::now();
            bool got = false;
            for (;;) {
                if (recv_msg(w.fd, type, pl, 1000)) { got = true; break; }
                workers_alive();
                if (w.exited) break;
                if ((long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms_) break;
                pollfd pf{w.fd, POLLIN, 0};
                if (::poll(&pf, 1, 0) > 0 && (pf.revents & (POLLHUP | POLLERR)) && !(pf.revents & POLLIN)) break;
            }
            if (!got || type != C_REPLY || pl.size() < 4) {
                workers_alive();
                fail("row split: rank " + std::to_string(w.rank) + (w.exited ? " exited" + exit_text(w.status) : " did not answer within " + std::to_string(timeout_ms_ / 1000) + " s") +
                     " (device work of the other ranks is abandoned)");
                return false;
            }
            int32_t rc;
            memcpy(&rc, pl.data(), 4);
            if (rc < 0 && pl.size() > 4) log_line(LOG_ERROR, "row split: rank %d: %s", w.rank, std::string(pl.begin() + 4, pl.end()).c_str());
            rcs.push_back(rc);
        }
        return true;
    }
    void shutdown() {
        for (auto &w : workers) if (!w.exited && w.fd >= 0 && !dead_) send_msg(w.fd, C_QUIT, nullptr, 0);
        const auto t0 = std::chrono::steady_clock::now();
        for (auto &w : workers) {
            while (!w.exited && w.pid > 0) {
                int st = 0;
                if (waitpid(w.pid, &st, WNOHANG) == w.pid) { w.exited = true; w.status = st; break; }
                if ((long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > 10000) { kill(w.pid, SIGKILL); waitpid(w.pid, &st, 0); w.exited = true; break; }
                std::this_thread::sleep_for(std::chrono::milliseconds(5));
            }
            if (w.fd >= 0) ::close(w.fd);
            w.fd = -1;
        }
        workers.clear();
        local.reset();                                  // (the context's exchanges are over before the group goes)
        tp_set_host_exchange(nullptr, nullptr, 0, 1);
        tp_shutdown();
        shm.reset();
    }

  private:
    static std::string exit_text(int st) {
        if (WIFSIGNALED(st)) return " on signal " + std::to_string(WTERMSIG(st));
        if (WIFEXITED(st)) return " with code " + std::to_string(WEXITSTATUS(st));
        return "";
    }
    std::string describe_exits() {
        workers_alive();
        std::string s;
        for (auto &w : workers) if (w.exited) s += "; rank " + std::to_string(w.rank) + " exited" + exit_text(w.status);
        return s;
    }
    bool broadcast(uint32_t type, const void *payload, size_t len) {
        for (auto &w : workers)
            if (!send_msg(w.fd, type, payload, len)) { workers_alive(); fail("row split: rank " + std::to_string(w.rank) + " is gone" + (w.exited ? exit_text(w.status) : std::string())); return false; }
        return true;
    }
    bool dead_ = false;
    std::string err_;
    long timeout_ms_;
    std::vector<uint8_t> msg_;
};

int count_devices() {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

}  // namespace

bool tp_split_requested(const Json &body) { return body["split_mode"].is_string() && body["split_mode"].as_string() == "row"; }

std::unique_ptr<IBackend> make_split_backend(const Json &body, BackendInfo &info, std::string &err) {
    const int devices = count_devices();
    if (devices < 1) { err = "split_mode row: no HIP device visible"; return nullptr; }
    int ranks = devices;
    if (body["tensor_split"].is_array() && body["tensor_split"].size() > 0) {
        double first = 0.0;
        int nz = 0;
        for (const Json &v : body["tensor_split"].items()) {
            const double x = v.is_number() ? v.as_double() : 0.0;
            if (x <= 0.0) continue;
            if (nz == 0) first = x;
            else if (x != first) { err = "tensor_split: this row split is even - every rank takes the same share of the heads and of the 256-element blocks"; return nullptr; }
            nz++;
        }
        if (nz > 0) ranks = nz;
    }
    if (body["split_ranks"].is_number()) ranks = (int)body["split_ranks"].as_int();
    if (ranks < 2 || ranks > 8) { err = "split_mode row: needs 2 to 8 ranks (visible devices: " + std::to_string(devices) + "; split_ranks makes ranks share devices, for validation)"; return nullptr; }
    if (tp_active()) { err = "split_mode row: a row-split model is already loaded in this process (one group per process)"; return nullptr; }
    const int main_gpu = body.value<int>("main_gpu", 0);
    const bool shared = ranks > devices;
    if (shared) log_line(LOG_WARN, "split_mode row: %d ranks on %d device(s) - ranks SHARE devices and exchange through shared memory: a validation set-up, not a deployment", ranks, devices);
    const std::string exe = worker_path();
    if (access(exe.c_str(), X_OK) != 0) { err = "split_mode row: worker program not found: " + exe + " (python cortex.llamacpp_amd/build.py builds it; MI355_TP_WORKER overrides the path)"; return nullptr; }

    std::unique_ptr<SplitBackend> sb(new SplitBackend);
    uint8_t id[128] = {0};
    size_t shm_cap = 0;
    if (shared) {
        shm_cap = (size_t)4 << 20;                         // floats per rank and piece (16 MB): longer messages go in pieces
        sb->shm.reset(new ShmExchange);
        if (!sb->shm->create(ranks, shm_cap, err)) return nullptr;
        sb->shm->set_timeout_ms(step_timeout_ms());
        SplitBackend *raw = sb.get();
        sb->shm->set_liveness([raw] { return raw->workers_alive(); });
    } else {
        std::string e2;
        if (hipSetDevice(main_gpu % devices) != hipSuccess || tp_unique_id(id, sizeof id, e2) < 0) { err = "split_mode row: " + (e2.empty() ? std::string("hipSetDevice failed") : e2); return nullptr; }
    }
    // ---- the workers: fresh processes, each told its rank, its device and how the group meets
    for (int r = 1; r < ranks; r++) {
        int sv[2];
        if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv) != 0) { err = "split_mode row: socketpair failed"; return nullptr; }
        fcntl(sv[0], F_SETFD, FD_CLOEXEC);                 // this side stays here (later workers must not inherit it)
        const std::string fd_arg = std::to_string(sv[1]);
        char *argv[] = {const_cast<char *>(exe.c_str()), const_cast<char *>(fd_arg.c_str()), nullptr};
        pid_t pid = -1;
        const int rc = posix_spawn(&pid, exe.c_str(), nullptr, nullptr, argv, environ);
        ::close(sv[1]);
        if (rc != 0) { ::close(sv[0]); err = "split_mode row: cannot start " + exe + ": " + strerror(rc); return nullptr; }
        Worker w;
        w.pid = pid; w.fd = sv[0]; w.rank = r;
        sb->workers.push_back(w);
        Json init = Json::object();
        init["rank"] = r; init["size"] = ranks; init["device"] = (main_gpu + r) % devices;
        init["transport"] = shared ? "shm" : "rccl";
        init["shm_fd"] = shared ? sb->shm->fd() : -1;
        init["shm_cap"] = (int64_t)shm_cap;
        init["timeout_ms"] = (int64_t)step_timeout_ms();
        init["rccl_id"] = hex(id, sizeof id);
        const std::string text = init.dump();
        if (!send_msg(w.fd, C_INIT, text.data(), text.size())) { err = "split_mode row: rank " + std::to_string(r) + " did not take its instructions"; return nullptr; }
    }
    // what a worker loads: this request's body for its rank and device.  The projector file is rank 0's alone (the image tower is nothing of the token path), but
    // its presence raises the context (hip_backend.cc: 2048 cells, 4096 for an image grid) - the workers must hold the same number of cells, so with `mmproj`
    // they are told what to load only once rank 0 knows its own context
    const bool has_mmproj = body["mmproj"].is_string() && !body["mmproj"].as_string().empty();
    auto send_loads = [&](int n_ctx_of_rank0) -> bool {
        for (auto &w : sb->workers) {
            Json wb = body;
            wb["split_mode"] = "none"; wb["tp_rank"] = w.rank; wb["tp_size"] = ranks; wb["main_gpu"] = (main_gpu + w.rank) % devices; wb["logits_to_host"] = false;
            if (has_mmproj) { wb["mmproj"] = Json(); wb["ctx_len"] = n_ctx_of_rank0; }
            const std::string text = wb.dump();
            if (!send_msg(w.fd, C_LOAD, text.data(), text.size())) { err = "split_mode row: rank " + std::to_string(w.rank) + " did not take its load request"; return false; }
        }
        return true;
    };
    if (!has_mmproj && !send_loads(0)) { sb->fail(err); return nullptr; }
    // ---- this process = rank 0
    if (shared) tp_set_host_exchange(&ShmExchange::callback, sb->shm.get(), 0, ranks);
    else {
        std::string e2;
        if (tp_init(0, ranks, id, sizeof id, e2) != 0) { err = "split_mode row: " + e2; sb->fail(err); return nullptr; }
    }
    Json lb = body;
    lb["split_mode"] = "none"; lb["tp_rank"] = 0; lb["tp_size"] = ranks; lb["main_gpu"] = main_gpu % devices;
    BackendInfo li;
    std::string lerr;
    sb->local = make_hip_backend(lb, li, lerr);
    if (!sb->local) { err = lerr.empty() ? "split_mode row: rank 0 failed to load" : lerr; sb->fail(err); return nullptr; }      // (the workers are ended: nobody waits for their loads)
    if (has_mmproj && !send_loads(sb->local->n_ctx())) { sb->fail(err); return nullptr; }
    // ---- every worker's answer to its load
    std::string werr;
    uint64_t vram = li.vram;
    for (auto &w : sb->workers) {
        uint32_t type = 0;
        std::vector<uint8_t> pl;
        bool got = false;
        const auto t0 = std::chrono::steady_clock::now();
        const long load_timeout = std::max<long>(step_timeout_ms(), 900000);        // (a 70B shard: planes are expanded at load)
        while (!got) {
            if (recv_msg(w.fd, type, pl, 1000)) { got = true; break; }
            sb->workers_alive();
            if (w.exited || (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > load_timeout) break;
        }
        Json rep;
        if (!got || type != C_REPLY || !Json::parse(std::string(pl.begin(), pl.end()), rep)) { if (werr.empty()) werr = "rank " + std::to_string(w.rank) + " did not report its load"; continue; }
        if (!rep["ok"].as_bool()) { if (werr.empty()) werr = "rank " + std::to_string(w.rank) + ": " + rep["error"].str_or("load failed"); continue; }
        vram += (uint64_t)rep["vram"].as_int();
    }
    if (!werr.empty()) { err = "split_mode row: " + werr; sb->fail(err); return nullptr; }
    info = li;
    info.vram = vram;
    log_line(LOG_INFO, "split_mode row: %d ranks (%s), rank 0 on device %d", ranks, shared ? "shared devices, shared-memory exchange" : "one device each, RCCL", main_gpu % devices);
    return sb;
}

int tp_split_worker_main(int fd) {
    uint32_t type = 0;
    std::vector<uint8_t> pl;
    if (!recv_msg(fd, type, pl, 60000) || type != C_INIT) return 2;
    Json init;
    if (!Json::parse(std::string(pl.begin(), pl.end()), init)) return 2;
    const int rank = (int)init["rank"].as_int(), size = (int)init["size"].as_int(), device = (int)init["device"].as_int();
    auto reply_load = [&](bool ok, const std::string &error, uint64_t vram) {
        Json r = Json::object();
        r["ok"] = ok; r["error"] = error; r["vram"] = (int64_t)vram;
        const std::string t = r.dump();
        send_msg(fd, C_REPLY, t.data(), t.size());
    };
    std::string err;
    ShmExchange shm;
    if (hipSetDevice(device) != hipSuccess) { reply_load(false, "hipSetDevice(" + std::to_string(device) + ") failed", 0); return 3; }
    if (init["transport"].str_or("") == "shm") {
        if (!shm.attach((int)init["shm_fd"].as_int(), rank, size, (size_t)init["shm_cap"].as_int(), err)) { reply_load(false, err, 0); return 3; }
        shm.set_timeout_ms((long)init["timeout_ms"].as_int());
        const pid_t parent = getppid();
        shm.set_liveness([parent] { return getppid() == parent; });      // the engine process is gone: so is this rank
        tp_set_host_exchange(&ShmExchange::callback, &shm, rank, size);
    } else {
        std::vector<uint8_t> id;
        if (!unhex(init["rccl_id"].str_or(""), id) || id.size() != 128) { reply_load(false, "bad RCCL id", 0); return 3; }
        if (tp_init(rank, size, id.data(), id.size(), err) != 0) { reply_load(false, err, 0); return 3; }
    }
    if (!recv_msg(fd, type, pl, -1) || type != C_LOAD) { tp_set_host_exchange(nullptr, nullptr, 0, 1); tp_shutdown(); return type == C_QUIT ? 0 : 2; }   // (rank 0's own load failed: told to go)
    Json load_body;
    if (!Json::parse(std::string(pl.begin(), pl.end()), load_body)) { reply_load(false, "bad load request", 0); return 3; }
    BackendInfo info;
    std::unique_ptr<IBackend> be = make_hip_backend(load_body, info, err);
    if (!be) { reply_load(false, err, 0); tp_set_host_exchange(nullptr, nullptr, 0, 1); tp_shutdown(); return 3; }
    reply_load(true, "", info.vram);

    auto reply_rc = [&](int32_t rc, const char *msg) {
        std::vector<uint8_t> out(4);
        memcpy(out.data(), &rc, 4);
        if (rc < 0 && msg) out.insert(out.end(), msg, msg + strlen(msg));
        return send_msg(fd, C_REPLY, out.data(), out.size());
    };
    int code = 0;
    for (;;) {
        if (!recv_msg(fd, type, pl, -1)) { code = 4; break; }          // the engine process closed the socket (or died)
        if (type == C_QUIT) break;
        const int32_t *a = reinterpret_cast<const int32_t *>(pl.data());
        bool sent = true;
        switch (type) {
            case C_DECODE: {
                if (pl.size() < 4) { code = 5; break; }
                const int32_t n = a[0];
                if (n < 0 || pl.size() != 4 + (size_t)n * 13) { code = 5; break; }
                BatchView b;
                b.n_tokens = n; b.token = a + 1; b.pos = a + 1 + n; b.seq_id = a + 1 + 2 * n;
                b.logits = reinterpret_cast<const int8_t *>(pl.data() + 4 + (size_t)n * 12);
                const int rc = be->decode(b);
                sent = reply_rc(rc, be->last_error());
                break;
            }
            case C_DECODE_EMBD: {
                if (pl.size() < 12) { code = 5; break; }
                const int32_t n = a[0];
                if (n <= 0 || pl.size() != 12 + (size_t)n * (size_t)be->n_embd() * 4) { code = 5; break; }
                const int rc = be->decode_embd(reinterpret_cast<const float *>(pl.data() + 12), n, a[1], a[2]);
                sent = reply_rc(rc, be->last_error());
                break;
            }
            case C_KV_CLEAR: be->kv_clear(); sent = reply_rc(0, nullptr); break;
            case C_KV_SEQ_RM: sent = pl.size() == 12 && reply_rc(be->kv_seq_rm(a[0], a[1], a[2]) ? 1 : 0, nullptr); break;
            case C_KV_SEQ_ADD: if (pl.size() == 16) be->kv_seq_add(a[0], a[1], a[2], a[3]); sent = reply_rc(0, nullptr); break;
            case C_KV_SEQ_CP: if (pl.size() == 16) be->kv_seq_cp(a[0], a[1], a[2], a[3]); sent = reply_rc(0, nullptr); break;
            case C_SET_EMBD: if (pl.size() == 4) be->set_embeddings(a[0] != 0); sent = reply_rc(0, nullptr); break;
            default: code = 5; break;
        }
        if (code || !sent) { if (!code) code = 4; break; }
    }
    be.reset();
    tp_set_host_exchange(nullptr, nullptr, 0, 1);
    tp_shutdown();
    return code;
}

}  // namespace mi355
