// mi355_server.cc — the HTTP host in front of the engine library: the routes the reference's example server exposes
// (/root/reference/examples/server/server.cc:262-273), so that the reference's e2e scripts
// (.github/scripts/e2e-test-server-linux-and-mac.sh: POST /loadmodel -> POST /v1/chat/completions (stream and not) ->
// POST /v1/embeddings -> POST /unloadmodel) can be pointed at this process unchanged:
//
//     POST /loadmodel   POST /unloadmodel   POST /v1/chat/completions   POST /v1/embeddings   POST /modelstatus
//     GET /models       DELETE /destroy     (+ GET /healthz)
//
// Like the reference's host it holds no model logic: it dlopen()s the engine library (here libmi355_llama.so, the C-ABI of
// include/mi355_llama.h; there ./engines/cortex.llamacpp through get_engine, server.cc:14-24), hands every request body to the
// engine call of the same name and relays what the engine's callback delivers - status["status_code"] becomes the HTTP status,
// a streaming completion becomes a chunked text/event-stream of the callback's "data" strings (server.cc:133-160), and a client
// that goes away mid-stream stops the generation (ForceStopInferencing, server.cc:27-33).
//
// The transport is written directly on POSIX sockets (httplib / drogon are not in this image): HTTP/1.1, Content-Length bodies,
// keep-alive, one thread per connection, at most 64 at a time (the reference's task queue is a 64-thread pool, server.cc:283-285).
//
//     mi355_server [host] [port] [--lib path/to/libmi355_llama.so]        (defaults 127.0.0.1 3928, as the reference)
#include <arpa/inet.h>
#include <dlfcn.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <signal.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../host/json.h"

using mi355::Json;

namespace {

// ------------------------------------------------------------------------------------------------ the engine library
typedef void (*engine_callback)(const char *status_json, const char *body_json, void *user);
struct EngineApi {
    void *dl = nullptr;
    void *(*create)() = nullptr;
    void (*destroy)(void *) = nullptr;
    void (*load_model)(void *, const char *, engine_callback, void *) = nullptr;
    void (*unload_model)(void *, const char *, engine_callback, void *) = nullptr;
    void (*get_model_status)(void *, const char *, engine_callback, void *) = nullptr;
    void (*get_models)(void *, const char *, engine_callback, void *) = nullptr;
    void (*chat_completion)(void *, const char *, engine_callback, void *) = nullptr;
    void (*embedding)(void *, const char *, engine_callback, void *) = nullptr;
    void (*stop_inferencing)(void *, const char *) = nullptr;
    const char *(*last_error)() = nullptr;

    bool open(const std::string &path, std::string &err) {
        dl = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!dl) { err = dlerror(); return false; }
        struct { const char *name; void **slot; } syms[] = {
            {"mi355_engine_create", (void **)&create}, {"mi355_engine_destroy", (void **)&destroy},
            {"mi355_engine_load_model", (void **)&load_model}, {"mi355_engine_unload_model", (void **)&unload_model},
            {"mi355_engine_get_model_status", (void **)&get_model_status}, {"mi355_engine_get_models", (void **)&get_models},
            {"mi355_engine_handle_chat_completion", (void **)&chat_completion}, {"mi355_engine_handle_embedding", (void **)&embedding},
            {"mi355_engine_stop_inferencing", (void **)&stop_inferencing}, {"mi355_last_error", (void **)&last_error},
        };
        for (auto &s : syms) {
            *s.slot = dlsym(dl, s.name);
            if (!*s.slot) { err = std::string("missing symbol ") + s.name; return false; }
        }
        return true;
    }
};

// what the engine's callback delivers, queued for the connection thread (the engine calls back from its own threads, and for a
// completion after the call has returned: server.cc:36-58)
struct ResultQueue {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::pair<std::string, std::string>> q;
    void push(const char *st, const char *body) {
        { std::lock_guard<std::mutex> l(m); q.emplace_back(st ? st : "{}", body ? body : "{}"); }
        cv.notify_one();
    }
    std::pair<std::string, std::string> pop() {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [this] { return !q.empty(); });
        auto r = std::move(q.front());
        q.pop_front();
        return r;
    }
};
// The engine calls back from its own threads, possibly after the HTTP exchange is over (the client left early) and - for a cancelled stream - possibly
// never with a final result.  So the callback's user pointer is not an owning reference but a ticket number into a registry: a callback whose ticket is
// gone (the final result was delivered, or the connection thread has dropped the exchange) is ignored instead of touching freed memory, and an exchange that
// never sees its final callback is dropped by the connection thread when it is done with it.
struct CallRegistry {
    std::mutex m;
    std::unordered_map<uintptr_t, std::shared_ptr<ResultQueue>> live;
    uintptr_t next = 1;
    uintptr_t add(const std::shared_ptr<ResultQueue> &q) { std::lock_guard<std::mutex> l(m); const uintptr_t id = next++; live[id] = q; return id; }
    std::shared_ptr<ResultQueue> find(uintptr_t id) { std::lock_guard<std::mutex> l(m); auto it = live.find(id); return it == live.end() ? nullptr : it->second; }
    void drop(uintptr_t id) { std::lock_guard<std::mutex> l(m); live.erase(id); }
};
CallRegistry g_calls;
void on_result(const char *status_json, const char *body_json, void *user) {
    const uintptr_t id = reinterpret_cast<uintptr_t>(user);
    std::shared_ptr<ResultQueue> q = g_calls.find(id);
    if (!q) return;                                           // a late or repeated callback of an exchange that is over
    bool last = true;
    Json st;
    if (status_json && Json::parse(status_json, st)) last = st.value<bool>("is_done", true) || st.value<bool>("has_error", false);
    if (last) g_calls.drop(id);
    q->push(status_json, body_json);
}
// header values that are echoed into a response: anything with a control character (a bare LF would split the response) is dropped
std::string echo_safe(const std::string &v) {
    for (unsigned char c : v) if ((c < 0x20 && c != '\t') || c == 0x7f) return std::string();
    return v;
}

// ------------------------------------------------------------------------------------------------ HTTP
struct Request {
    std::string method, path, body;
    std::vector<std::pair<std::string, std::string>> headers;   // names lower-cased
    bool keep_alive = true;
    const std::string &header(const std::string &name) const {
        static const std::string none;
        for (const auto &h : headers) if (h.first == name) return h.second;
        return none;
    }
};

constexpr size_t kMaxHeader = 64 * 1024, kMaxBody = 256u * 1024 * 1024;
constexpr int kIdleMs = 30000;

bool send_all(int fd, const char *p, size_t n) {
    while (n > 0) {
        const ssize_t w = ::send(fd, p, n, MSG_NOSIGNAL);
        if (w < 0) { if (errno == EINTR) continue; return false; }
        p += w; n -= (size_t)w;
    }
    return true;
}
bool send_all(int fd, const std::string &s) { return send_all(fd, s.data(), s.size()); }

// reads into buf until it holds at least `want` bytes or the delimiter; false on EOF / error / idle time-out
bool fill(int fd, std::string &buf, size_t limit) {
    if (buf.size() >= limit) return false;
    pollfd p{fd, POLLIN, 0};
    const int pr = ::poll(&p, 1, kIdleMs);
    if (pr <= 0) return false;
    char tmp[16384];
    const ssize_t r = ::recv(fd, tmp, sizeof tmp, 0);
    if (r <= 0) return false;
    buf.append(tmp, (size_t)r);
    return true;
}

std::string lower(std::string s) { for (auto &c : s) if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a'); return s; }
std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t')) a++;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r')) b--;
    return s.substr(a, b - a);
}

// 0 = a request was read, 1 = the peer closed between requests, < 0 = malformed (-400, -413, -501: the status to answer with)
int read_request(int fd, std::string &buf, Request &rq) {
    size_t he;
    while ((he = buf.find("\r\n\r\n")) == std::string::npos) {
        if (buf.size() > kMaxHeader) return -431;
        const bool had = !buf.empty();
        if (!fill(fd, buf, kMaxHeader + 4)) return had ? -400 : 1;
    }
    const std::string head = buf.substr(0, he);
    size_t ls = head.find("\r\n");
    const std::string start = head.substr(0, ls);
    const size_t s1 = start.find(' '), s2 = start.rfind(' ');
    if (s1 == std::string::npos || s2 == s1) return -400;
    rq.method = start.substr(0, s1);
    rq.path = start.substr(s1 + 1, s2 - s1 - 1);
    const std::string ver = start.substr(s2 + 1);
    if (ver.compare(0, 5, "HTTP/") != 0) return -400;
    const size_t qm = rq.path.find('?');
    if (qm != std::string::npos) rq.path.resize(qm);
    rq.headers.clear();
    while (ls != std::string::npos && ls + 2 <= head.size()) {
        const size_t ln = head.find("\r\n", ls + 2);
        const std::string line = head.substr(ls + 2, ln == std::string::npos ? std::string::npos : ln - ls - 2);
        const size_t c = line.find(':');
        if (c != std::string::npos) rq.headers.emplace_back(lower(trim(line.substr(0, c))), trim(line.substr(c + 1)));
        ls = ln;
    }
    const std::string conn = lower(rq.header("connection"));
    rq.keep_alive = ver == "HTTP/1.0" ? conn == "keep-alive" : conn != "close";
    if (!rq.header("transfer-encoding").empty() && lower(rq.header("transfer-encoding")) != "identity") return -501;   // (chunked request bodies: no client of these routes sends them)
    size_t clen = 0;
    const std::string &cl = rq.header("content-length");
    if (!cl.empty()) {
        char *e = nullptr;
        const unsigned long long v = strtoull(cl.c_str(), &e, 10);
        if (!e || *e || v > kMaxBody) return v > kMaxBody ? -413 : -400;
        clen = (size_t)v;
    }
    if (rq.header("expect") == "100-continue" && !send_all(fd, "HTTP/1.1 100 Continue\r\n\r\n")) return 1;
    buf.erase(0, he + 4);
    while (buf.size() < clen) if (!fill(fd, buf, clen + kMaxHeader)) return -400;
    rq.body = buf.substr(0, clen);
    buf.erase(0, clen);
    return 0;
}

const char *reason(int code) {
    switch (code) {
        case 200: return "OK"; case 400: return "Bad Request"; case 404: return "Not Found"; case 405: return "Method Not Allowed";
        case 409: return "Conflict"; case 413: return "Payload Too Large"; case 431: return "Request Header Fields Too Large";
        case 500: return "Internal Server Error"; case 501: return "Not Implemented"; case 503: return "Service Unavailable";
        default: return code >= 200 && code < 300 ? "OK" : code >= 400 && code < 500 ? "Client Error" : "Server Error";
    }
}
std::string head_of(int code, const std::string &ctype, const std::string &origin, bool keep_alive, long content_length) {
    std::string h = "HTTP/1.1 " + std::to_string(code) + " " + reason(code) + "\r\n";
    h += "Content-Type: " + ctype + "\r\n";
    h += "Access-Control-Allow-Origin: " + echo_safe(origin) + "\r\n";        // (server.cc:164-165: the request's Origin, echoed)
    if (content_length >= 0) h += "Content-Length: " + std::to_string(content_length) + "\r\n";
    else h += "Transfer-Encoding: chunked\r\nCache-Control: no-cache\r\n";
    h += keep_alive ? "Connection: keep-alive\r\n" : "Connection: close\r\n";
    h += "\r\n";
    return h;
}
bool send_json(int fd, int code, const std::string &body, const Request &rq) {
    return send_all(fd, head_of(code, "application/json; charset=utf-8", rq.header("origin"), rq.keep_alive, (long)body.size()) + body);
}
std::string error_body(const std::string &msg) { Json j = Json::object(); j["message"] = msg; return j.dump(); }

// ------------------------------------------------------------------------------------------------ the server
struct Server {
    EngineApi api;
    void *engine = nullptr;
    std::atomic<bool> running{true};
    std::atomic<int> live{0};
    int listen_fd = -1;

    typedef void (*engine_fn)(void *, const char *, engine_callback, void *);

    struct Call {                       // one engine call and its result queue; the ticket is dropped when the exchange ends, whatever the engine still does
        std::shared_ptr<ResultQueue> q;
        uintptr_t id = 0;
        Call() = default;
        Call(const Call &) = delete;
        Call &operator=(const Call &) = delete;
        ~Call() { if (id) g_calls.drop(id); }
        ResultQueue *operator->() const { return q.get(); }
    };
    void call(Call &c, engine_fn fn, const std::string &body) {
        c.q = std::make_shared<ResultQueue>();
        c.id = g_calls.add(c.q);
        fn(engine, body.empty() ? "{}" : body.c_str(), on_result, reinterpret_cast<void *>(c.id));
    }
    static int status_code(const std::string &status_json, bool *done = nullptr, bool *error = nullptr) {
        Json st;
        if (!Json::parse(status_json, st)) { if (done) *done = true; if (error) *error = true; return 500; }
        if (done) *done = st.value<bool>("is_done", true);
        if (error) *error = st.value<bool>("has_error", false);
        return st.value<int>("status_code", 200);
    }

    // one result, one JSON response (process_non_stream_res, server.cc:124-131)
    bool relay_one(int fd, const Request &rq, engine_fn fn) {
        Call q;
        call(q, fn, rq.body);
        const auto r = q->pop();
        return send_json(fd, status_code(r.first), r.second, rq);
    }
    // a streaming completion: the "data" string of every callback as one chunk, until is_done / has_error (process_stream_res, server.cc:133-160)
    bool relay_stream(int fd, const Request &rq, const std::string &model_id) {
        Call q;
        call(q, api.chat_completion, rq.body);
        auto first = q->pop();
        bool done = false, error = false;
        int code = status_code(first.first, &done, &error);
        Json res;
        const bool shaped = Json::parse(first.second, res) && res["data"].is_string();
        if (!shaped) return send_json(fd, code, first.second, rq);        // refused before any token (unknown model, bad body): a plain JSON error
        if (!send_all(fd, head_of(200, "text/event-stream", rq.header("origin"), rq.keep_alive, -1))) { api.stop_inferencing(engine, model_id.c_str()); return false; }
        bool ok = true;
        for (;;) {
            const std::string &data = res["data"].as_string();
            if (!data.empty()) {
                char len[32];
                snprintf(len, sizeof len, "%zx\r\n", data.size());
                if (!send_all(fd, std::string(len) + data + "\r\n")) { ok = false; break; }
            }
            if (done || error) break;
            auto nx = q->pop();
            code = status_code(nx.first, &done, &error);
            res = Json();
            if (!Json::parse(nx.second, res)) res = Json::object();
        }
        if (!ok) { api.stop_inferencing(engine, model_id.c_str()); return false; }     // the client went away: stop generating for it
        return send_all(fd, "0\r\n\r\n");
    }

    // false: close the connection
    bool handle(int fd, const Request &rq) {
        const std::string &m = rq.method, &p = rq.path;
        if (m == "OPTIONS") {
            const std::string acrh = echo_safe(rq.header("access-control-request-headers"));
            std::string h = "HTTP/1.1 204 No Content\r\nAccess-Control-Allow-Origin: " + echo_safe(rq.header("origin")) +
                            "\r\nAccess-Control-Allow-Methods: GET, POST, DELETE, OPTIONS\r\nAccess-Control-Allow-Headers: " +
                            (acrh.empty() ? std::string("Content-Type, Authorization") : acrh) +
                            "\r\nContent-Length: 0\r\n" + (rq.keep_alive ? "Connection: keep-alive\r\n" : "Connection: close\r\n") + "\r\n";
            return send_all(fd, h);
        }
        if (m == "POST" && p == "/loadmodel") return relay_one(fd, rq, api.load_model);
        if (m == "POST" && p == "/unloadmodel") return relay_one(fd, rq, api.unload_model);
        if (m == "POST" && p == "/modelstatus") return relay_one(fd, rq, api.get_model_status);
        if (m == "POST" && p == "/v1/embeddings") return relay_one(fd, rq, api.embedding);
        if ((m == "GET" || m == "POST") && p == "/models") return relay_one(fd, rq, api.get_models);
        if (m == "POST" && p == "/v1/chat/completions") {
            Json body;
            const bool parsed = Json::parse(rq.body, body);
            if (parsed && body.value<bool>("stream", false)) return relay_stream(fd, rq, body.value<std::string>("model", "invalid_model"));
            return relay_one(fd, rq, api.chat_completion);
        }
        if (m == "GET" && p == "/healthz") return send_json(fd, 200, "{\"status\":\"ok\"}", rq);
        if (m == "DELETE" && p == "/destroy") {
            const bool ok = send_json(fd, 200, "{\"message\":\"Server stopped\"}", rq);
            stop();
            (void)ok;
            return false;
        }
        static const char *known[] = {"/loadmodel", "/unloadmodel", "/modelstatus", "/v1/embeddings", "/v1/chat/completions", "/models", "/destroy", "/healthz"};
        for (const char *k : known) if (p == k) return send_json(fd, 405, error_body("method not allowed"), rq);
        return send_json(fd, 404, error_body("no such route"), rq);
    }

    void serve(int fd) {
        int one = 1;
        setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        std::string buf;
        for (;;) {
            Request rq;
            const int r = read_request(fd, buf, rq);
            if (r == 1) break;
            if (r < 0) { rq.keep_alive = false; send_json(fd, -r, error_body("malformed request"), rq); break; }
            if (!handle(fd, rq) || !rq.keep_alive || !running) break;
        }
        ::shutdown(fd, SHUT_RDWR);
        ::close(fd);
        live--;
    }

    void stop() {
        running = false;
        if (listen_fd >= 0) ::shutdown(listen_fd, SHUT_RDWR);
    }

    int run(const std::string &host, int port) {
        listen_fd = ::socket(AF_INET, SOCK_STREAM, 0);
        if (listen_fd < 0) { perror("socket"); return 1; }
        int one = 1;
        setsockopt(listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
        sockaddr_in addr{};
        addr.sin_family = AF_INET;
        addr.sin_port = htons((uint16_t)port);
        if (host == "0.0.0.0" || host.empty()) addr.sin_addr.s_addr = INADDR_ANY;
        else if (host == "localhost") addr.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
        else if (inet_pton(AF_INET, host.c_str(), &addr.sin_addr) != 1) { fprintf(stderr, "bad host %s\n", host.c_str()); return 1; }
        if (::bind(listen_fd, (sockaddr *)&addr, sizeof addr) != 0 || ::listen(listen_fd, 128) != 0) {
            fprintf(stderr, "\ncouldn't bind to server socket: hostname=%s port=%d\n\n", host.c_str(), port);
            return 1;
        }
        socklen_t al = sizeof addr;
        getsockname(listen_fd, (sockaddr *)&addr, &al);
        fprintf(stderr, "HTTP server listening: %s:%d\n", host.c_str(), (int)ntohs(addr.sin_port));
        fflush(stderr);
        while (running) {
            pollfd p{listen_fd, POLLIN, 0};
            const int pr = ::poll(&p, 1, 100);
            if (pr <= 0 || !running) continue;
            const int fd = ::accept(listen_fd, nullptr, nullptr);
            if (fd < 0) continue;
            if (live.load() >= 64) {            // the reference's pool has 64 workers; beyond that a connection is told to come back
                Request rq; rq.keep_alive = false;
                send_json(fd, 503, error_body("server busy"), rq);
                ::close(fd);
                continue;
            }
            live++;
            std::thread(&Server::serve, this, fd).detach();
        }
        ::close(listen_fd);
        for (int i = 0; i < 300 && live.load() > 0; i++) std::this_thread::sleep_for(std::chrono::milliseconds(10));   // let open exchanges finish
        return live.load() > 0 ? -1 : 0;     // -1: connection threads are still inside engine calls (main must not destroy the engine under them)
    }
};

Server *g_server = nullptr;
std::atomic_flag g_terminating = ATOMIC_FLAG_INIT;
void on_signal(int) {
    if (g_terminating.test_and_set()) _exit(1);     // a second interrupt ends the process at once (server.cc:78-88)
    if (g_server) g_server->running = false;
}

std::string default_lib(const char *argv0) {
    if (const char *e = getenv("MI355_LLAMA_LIB")) return e;
    std::string self = argv0 ? argv0 : "";
    char real[4096];
    const ssize_t n = readlink("/proc/self/exe", real, sizeof real - 1);
    if (n > 0) { real[n] = 0; self = real; }
    const size_t sl = self.rfind('/');
    const std::string dir = sl == std::string::npos ? "." : self.substr(0, sl);
    return dir + "/../lib/libmi355_llama.so";
}

}  // namespace

int main(int argc, char **argv) {
    std::string host = "127.0.0.1", lib;
    int port = 3928, pos = 0;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--lib" && i + 1 < argc) lib = argv[++i];
        else if (a == "-h" || a == "--help") { printf("usage: %s [host] [port] [--lib libmi355_llama.so]\n", argv[0]); return 0; }
        else if (pos == 0) { host = a; pos++; }
        else if (pos == 1) { port = atoi(a.c_str()); pos++; }
    }
    if (lib.empty()) lib = default_lib(argv[0]);
    Server s;
    std::string err;
    if (!s.api.open(lib, err)) { fprintf(stderr, "cannot load the engine library %s: %s\n", lib.c_str(), err.c_str()); return 2; }
    s.engine = s.api.create();
    if (!s.engine) { fprintf(stderr, "mi355_engine_create failed: %s\n", s.api.last_error()); return 2; }
    g_server = &s;
    struct sigaction sa{};
    sa.sa_handler = on_signal;
    sigemptyset(&sa.sa_mask);
    sigaction(SIGINT, &sa, nullptr);
    sigaction(SIGTERM, &sa, nullptr);
    signal(SIGPIPE, SIG_IGN);
    const int rc = s.run(host, port);
    if (rc < 0) {                       // detached connection threads still hold the engine: leave the teardown to the process exit
        fflush(nullptr);
        _exit(0);
    }
    s.api.destroy(s.engine);
    return rc;
}
