// shm_exchange_test.cc — host/shm_exchange.h between forked processes on CPU: `shm_exchange_test <ranks>` forks ranks - 1 children that attach to the parent's
// segment, runs all-reduces and all-gathers of several lengths (shorter than, equal to and longer than a piece; a piece of `cap` floats) and checks every float
// on every rank; then the failure paths: a rank that exits makes the others leave their wait with an error instead of hanging, and a rank that never arrives
// runs the others into the time bound.  Exit code 0 = all checks passed.
#include <sys/wait.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../cortex.llamacpp_amd/host/shm_exchange.h"

using mi355::ShmExchange;

static int run_rank(ShmExchange &x, int rank, int size, size_t cap) {
    const size_t lens[] = {1, 7, cap - 1, cap, cap + 1, 3 * cap + 5};
    for (int rep = 0; rep < 3; rep++)
        for (size_t n : lens) {
            std::vector<float> a(n);
            for (size_t i = 0; i < n; i++) a[i] = (float)((i % 97) + 1) * (float)(rank + 1) + (float)rep;
            if (x.exchange(a.data(), n, 0) != 0) return 10;
            for (size_t i = 0; i < n; i++) {
                float want = 0.0f;
                for (int q = 0; q < size; q++) want += (float)((i % 97) + 1) * (float)(q + 1) + (float)rep;      // rank order: the same association
                if (a[i] != want) return 11;
            }
            std::vector<float> g(n * (size_t)size, -1.0f);
            for (size_t i = 0; i < n; i++) g[(size_t)rank * n + i] = (float)(rank * 1000) + (float)(i % 513);
            if (x.exchange(g.data(), n, 1) != 0) return 12;
            for (int q = 0; q < size; q++)
                for (size_t i = 0; i < n; i++)
                    if (g[(size_t)q * n + i] != (float)(q * 1000) + (float)(i % 513)) return 13;
        }
    return 0;
}

int main(int argc, char **argv) {
    const int size = argc > 1 ? atoi(argv[1]) : 2;
    const size_t cap = 1000;
    std::string err;
    // ---- 1. the exchanges themselves
    {
        ShmExchange x;
        if (!x.create(size, cap, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        std::vector<pid_t> kids;
        for (int r = 1; r < size; r++) {
            const pid_t p = fork();
            if (p == 0) {
                ShmExchange y;
                std::string e;
                if (!y.attach(dup(x.fd()), r, size, cap, e)) _exit(20);
                _exit(run_rank(y, r, size, cap));
            }
            kids.push_back(p);
        }
        int rc = run_rank(x, 0, size, cap);
        for (pid_t p : kids) { int st = 0; waitpid(p, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = rc ? rc : 30 + (WIFEXITED(st) ? WEXITSTATUS(st) : 99); }
        if (rc) { fprintf(stderr, "exchange checks failed: %d\n", rc); return rc; }
    }
    // ---- 2. a rank exits before an exchange: rank 0's liveness check ends the wait of everybody
    {
        ShmExchange x;
        if (!x.create(size, cap, err)) return 1;
        x.set_timeout_ms(30000);
        std::vector<pid_t> kids;
        for (int r = 1; r < size; r++) {
            const pid_t p = fork();
            if (p == 0) {
                if (r == size - 1) _exit(0);                        // never takes part
                ShmExchange y;
                std::string e;
                if (!y.attach(dup(x.fd()), r, size, cap, e)) _exit(20);
                y.set_timeout_ms(30000);
                float v = 1.0f;
                _exit(y.exchange(&v, 1, 0) != 0 && y.dead() ? 0 : 21);       // must come back with an error
            }
            kids.push_back(p);
        }
        const pid_t gone = kids.back();
        x.set_liveness([gone] { int st = 0; return waitpid(gone, &st, WNOHANG) == 0; });
        float v = 1.0f;
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = x.exchange(&v, 1, 0);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rc == 0 || s > 10.0 || x.error().find("exited") == std::string::npos) { fprintf(stderr, "a dead rank was not noticed (rc %d, %.1f s, '%s')\n", rc, s, x.error().c_str()); return 40; }
        for (size_t i = 0; i + 1 < kids.size(); i++) { int st = 0; waitpid(kids[i], &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { fprintf(stderr, "a peer of the dead rank hung or passed\n"); return 41; } }
    }
    // ---- 3. the time bound: nobody else ever arrives
    if (size > 1) {
        ShmExchange x;
        if (!x.create(size, cap, err)) return 1;
        x.set_timeout_ms(300);
        float v = 1.0f;
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = x.exchange(&v, 1, 0);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rc == 0 || s > 5.0 || x.error().find("waited") == std::string::npos) { fprintf(stderr, "the time bound did not end the wait (rc %d, %.1f s)\n", rc, s); return 50; }
    }
    printf("shm exchange: all checks passed (%d ranks)\n", size);
    return 0;
}
