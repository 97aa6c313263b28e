// mi355_tp_worker.cc — one further rank of a row split that an engine process formed for itself (`"split_mode": "row"`, host/tp_split.cc).  The engine starts
// this program once per rank > 0 with the number of an inherited socket; everything else - which device, how the group meets, what to load, every batch -
// arrives over that socket.  The program holds no logic: it dlopen()s the engine library (beside it, ../lib/libmi355_llama.so, or MI355_LLAMA_LIB) and hands
// the socket to mi355_tp_worker_main.  It touches no GPU before it has been told which one is its own.
#include <dlfcn.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <string>

int main(int argc, char **argv) {
    if (argc != 2) { fprintf(stderr, "usage: mi355_tp_worker <socket fd>   (started by the engine library, not by hand)\n"); return 64; }
    std::string lib;
    if (const char *e = getenv("MI355_LLAMA_LIB")) lib = e;
    else {
        char real[4096];
        const ssize_t n = readlink("/proc/self/exe", real, sizeof real - 1);
        std::string self = argv[0];
        if (n > 0) { real[n] = 0; self = real; }
        const size_t sl = self.rfind('/');
        lib = (sl == std::string::npos ? std::string(".") : self.substr(0, sl)) + "/../lib/libmi355_llama.so";
    }
    void *dl = dlopen(lib.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!dl) { fprintf(stderr, "mi355_tp_worker: %s\n", dlerror()); return 65; }
    typedef int (*main_fn)(int);
    main_fn fn = reinterpret_cast<main_fn>(dlsym(dl, "mi355_tp_worker_main"));
    if (!fn) { fprintf(stderr, "mi355_tp_worker: %s lacks mi355_tp_worker_main\n", lib.c_str()); return 65; }
    return fn(atoi(argv[1]));
}
