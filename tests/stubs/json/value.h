// STUB of jsoncpp's <json/value.h> for tests/test_adapter_compiles.py: just enough of Json::Value / Json::Reader / Json::FastWriter for
// integration/mi355_engine_adapter.cc to COMPILE against the reference's enginei.h.  Not a JSON implementation (parse() fails, write() returns "{}"):
// the test checks that the adapter class is concrete and that get_engine links, nothing else.
#pragma once
#include <map>
#include <string>

namespace Json {
class Value {
 public:
  Value() = default;
  Value(bool b) : b_(b) {}
  Value& operator[](const char* k) { return kids_[k]; }
  Value& operator[](const std::string& k) { return kids_[k]; }
  bool asBool() const { return b_; }
 private:
  bool b_ = false;
  std::map<std::string, Value> kids_;
};
class Reader {
 public:
  bool parse(const std::string&, Value&) { return false; }
  bool parse(const char*, Value&) { return false; }
};
class FastWriter {
 public:
  std::string write(const Value&) { return "{}"; }
};
}  // namespace Json
