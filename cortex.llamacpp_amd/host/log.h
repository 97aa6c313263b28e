// log.h — the engine's logger: what the reference does with trantor::Logger + trantor::FileLogger (src/llama_engine.cc:289-303 Load / Unload,
// :502-504 SetLogLevel, :510-548 SetFileLogger, and the LOG_INFO / LOG_WARN / LOG_ERROR lines around model load and request errors).  Levels are
// trantor's numbers (kTrace 0, kDebug 1, kInfo 2, kWarn 3, kError 4, kFatal 5) so an EngineI adapter passes them through unchanged.  Sinks: stderr
// (default), a file kept to its last `max_lines` lines (SetFileLogger), and / or a callback (the adapter's bridge back into the host's own logger).
#pragma once

#include <string>

namespace mi355 {

enum { LOG_TRACE = 0, LOG_DEBUG = 1, LOG_INFO = 2, LOG_WARN = 3, LOG_ERROR = 4, LOG_FATAL = 5 };

void log_set_level(int level);                                   // messages below it are dropped (default LOG_INFO)
int log_level();
bool log_set_file(const std::string &path, int max_lines);       // "" closes the file; false if it cannot be opened
typedef void (*log_callback)(int level, const char *line, void *user);
void log_set_callback(log_callback cb, void *user);              // nullptr removes it
void log_line(int level, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

}  // namespace mi355
