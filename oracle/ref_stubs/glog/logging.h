// TEST INFRASTRUCTURE ONLY (oracle). Minimal stand-in for <glog/logging.h> so that the reference's headers build
// without glog: CHECK aborts with its message like glog's, LOG(FATAL) aborts, every other log line is swallowed.
#pragma once
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>
namespace gv_ref_stub {
struct CheckSink {
    bool failed;
    std::ostringstream ss;
    explicit CheckSink(bool f) : failed(f) {}
    ~CheckSink() {
        if (failed) {
            std::cerr << "CHECK failed: " << ss.str() << std::endl;
            std::abort();
        }
    }
    template <class T> CheckSink &operator<<(const T &v) {
        if (failed) ss << v;
        return *this;
    }
    CheckSink &operator<<(std::ostream &(*)(std::ostream &)) { return *this; }
};
}  // namespace gv_ref_stub
namespace google {
enum { INFO = 0, WARNING = 1, ERROR = 2, FATAL = 3 };
inline void InitGoogleLogging(const char *) {}
}  // namespace google
static int FLAGS_minloglevel = 0;
static bool FLAGS_logtostderr = true, FLAGS_log_prefix = false;
static std::string FLAGS_log_dir;
#define CHECK(cond) gv_ref_stub::CheckSink(!(cond))
#define LOG(severity) gv_ref_stub::CheckSink(google::severity == google::FATAL)
#define LOG_IF(severity, cond) gv_ref_stub::CheckSink(google::severity == google::FATAL && (cond))
#define LOG_EVERY_N(severity, n) gv_ref_stub::CheckSink(false)
