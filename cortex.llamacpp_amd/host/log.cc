// log.cc — see log.h.  One process-wide logger behind a mutex (the reference's trantor::Logger is process-wide too); the file sink rewrites the file
// with its newest half once it holds more than max_lines lines, which keeps "the last N lines" without a background thread.
#include "log.h"

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <ctime>
#include <deque>
#include <mutex>

namespace mi355 {

namespace {
std::mutex g_m;
int g_level = LOG_INFO;
FILE *g_file = nullptr;
std::string g_path;
int g_max_lines = 0;
std::deque<std::string> g_tail;        // lines written since the file was (re)opened, newest last
log_callback g_cb = nullptr;
void *g_cb_user = nullptr;
bool g_stderr = true;
const char *level_name(int l) { static const char *n[] = {"TRACE", "DEBUG", "INFO", "WARN", "ERROR", "FATAL"}; return l >= 0 && l <= 5 ? n[l] : "INFO"; }
}  // namespace

void log_set_level(int level) { std::lock_guard<std::mutex> lk(g_m); g_level = level < 0 ? 0 : level > 5 ? 5 : level; }
int log_level() { std::lock_guard<std::mutex> lk(g_m); return g_level; }

bool log_set_file(const std::string &path, int max_lines) {
    std::lock_guard<std::mutex> lk(g_m);
    if (g_file) { fclose(g_file); g_file = nullptr; }
    g_tail.clear();
    g_path = path; g_max_lines = max_lines;
    if (path.empty()) { g_stderr = true; return true; }
    g_file = fopen(path.c_str(), "a");
    if (!g_file) { g_stderr = true; return false; }
    g_stderr = false;                              // the reference replaces trantor's output function: lines go to the file only
    return true;
}

void log_set_callback(log_callback cb, void *user) { std::lock_guard<std::mutex> lk(g_m); g_cb = cb; g_cb_user = user; }

void log_line(int level, const char *fmt, ...) {
    std::lock_guard<std::mutex> lk(g_m);
    if (level < g_level) return;
    char msg[2048];
    va_list ap; va_start(ap, fmt); vsnprintf(msg, sizeof msg, fmt, ap); va_end(ap);
    if (g_cb) g_cb(level, msg, g_cb_user);
    char ts[40];
    const auto now = std::chrono::system_clock::now();
    const std::time_t t = std::chrono::system_clock::to_time_t(now);
    std::tm tmv; gmtime_r(&t, &tmv);
    const int us = (int)(std::chrono::duration_cast<std::chrono::microseconds>(now.time_since_epoch()).count() % 1000000);
    snprintf(ts, sizeof ts, "%04d%02d%02d %02d:%02d:%02d.%06d UTC", tmv.tm_year + 1900, tmv.tm_mon + 1, tmv.tm_mday, tmv.tm_hour, tmv.tm_min, tmv.tm_sec, us);
    std::string line = std::string(ts) + " " + level_name(level) + " " + msg + "\n";
    if (g_file) {
        fputs(line.c_str(), g_file); fflush(g_file);
        if (g_max_lines > 0) {
            g_tail.push_back(line);
            if ((int)g_tail.size() > 2 * g_max_lines) {       // keep the newest max_lines lines
                while ((int)g_tail.size() > g_max_lines) g_tail.pop_front();
                if (FILE *f = freopen(g_path.c_str(), "w", g_file)) { g_file = f; for (const auto &l : g_tail) fputs(l.c_str(), g_file); fflush(g_file); }
            }
        }
    } else if (g_stderr && !g_cb) {
        fputs(line.c_str(), stderr);
    }
}

}  // namespace mi355
