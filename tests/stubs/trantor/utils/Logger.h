// STUB of <trantor/utils/Logger.h> for tests/test_adapter_compiles.py: the LogLevel enumeration (the numbers the C-ABI passes through), setLogLevel and
// LOG_* stream macros that discard their input.  Enough for the adapter and the reference's enginei.h to compile; it logs nothing.
#pragma once
#include <ostream>

namespace trantor {
class Logger {
 public:
  enum LogLevel { kTrace = 0, kDebug, kInfo, kWarn, kError, kFatal, kNumberOfLogLevels };
  static void setLogLevel(LogLevel l) { level() = l; }
  static LogLevel logLevel() { return level(); }
 private:
  static LogLevel& level() { static LogLevel l = kInfo; return l; }
};
struct NullLogStream {
  template <class T> NullLogStream& operator<<(const T&) { return *this; }
};
}  // namespace trantor
#define LOG_TRACE ::trantor::NullLogStream()
#define LOG_DEBUG ::trantor::NullLogStream()
#define LOG_INFO ::trantor::NullLogStream()
#define LOG_WARN ::trantor::NullLogStream()
#define LOG_ERROR ::trantor::NullLogStream()
