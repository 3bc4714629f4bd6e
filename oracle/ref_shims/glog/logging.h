// -*- c++ -*-
// Stand-in for the glog macros the reference uses (TEST INFRASTRUCTURE ONLY, see Eigen/Dense here).
// FATAL / failed CHECK: the message goes to stderr and std::abort() is called, as glog does.
#pragma once
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>

namespace google {
enum { GLOG_INFO = 0, GLOG_WARNING = 1, GLOG_ERROR = 2, GLOG_FATAL = 3 };
inline int& shim_min_level() {
  static int lvl = GLOG_WARNING;
  return lvl;
}
class LogMessage {
 public:
  LogMessage(const char* file, int line, int sev) : sev_(sev) { ss_ << file << ":" << line << "] "; }
  [[noreturn]] void die() {
    std::cerr << ss_.str() << std::endl;
    std::abort();
  }
  ~LogMessage() {
    if (sev_ >= GLOG_FATAL) die();
    if (sev_ >= shim_min_level()) std::cerr << ss_.str() << std::endl;
  }
  std::ostream& stream() { return ss_; }

 private:
  int sev_;
  std::ostringstream ss_;
};
class LogMessageFatal : public LogMessage {
 public:
  LogMessageFatal(const char* file, int line) : LogMessage(file, line, GLOG_FATAL) {}
  [[noreturn]] ~LogMessageFatal() { die(); }
};
struct Voidify {
  void operator&(std::ostream&) {}
};
}  // namespace google
#define LOG_INFO ::google::LogMessage(__FILE__, __LINE__, ::google::GLOG_INFO)
#define LOG_WARNING ::google::LogMessage(__FILE__, __LINE__, ::google::GLOG_WARNING)
#define LOG_ERROR ::google::LogMessage(__FILE__, __LINE__, ::google::GLOG_ERROR)
#define LOG_FATAL ::google::LogMessageFatal(__FILE__, __LINE__)
#define LOG(sev) LOG_##sev.stream()
#define VLOG(n) if (false) LOG(INFO)
#define LOG_IF(sev, cond) !(cond) ? (void)0 : ::google::Voidify() & LOG(sev)
#define CHECK(cond) (cond) ? (void)0 : ::google::Voidify() & LOG(FATAL) << "Check failed: " #cond " "
#define CHECK_OP_(a, b, op) CHECK((a)op(b))
#define CHECK_EQ(a, b) CHECK_OP_(a, b, ==)
#define CHECK_NE(a, b) CHECK_OP_(a, b, !=)
#define CHECK_LE(a, b) CHECK_OP_(a, b, <=)
#define CHECK_LT(a, b) CHECK_OP_(a, b, <)
#define CHECK_GE(a, b) CHECK_OP_(a, b, >=)
#define CHECK_GT(a, b) CHECK_OP_(a, b, >)
#define CHECK_NEAR(a, b, tol) CHECK(std::abs((a) - (b)) <= (tol))
#define CHECK_NOTNULL(p) (p)
#define DCHECK(cond) CHECK(cond)
