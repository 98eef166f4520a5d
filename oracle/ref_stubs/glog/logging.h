// TEST INFRASTRUCTURE ONLY (oracle). Minimal stand-in for <glog/logging.h> so the
// reference's host-compilable arithmetic headers (core/optimizer.h, which uses CHECK at
// optimizer.h:52,116) build without glog. A failed CHECK aborts, as glog's would.
#pragma once
#include <cstdlib>
#include <iostream>
#include <sstream>
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
};
}  // namespace gv_ref_stub
#define CHECK(cond) gv_ref_stub::CheckSink(!(cond))
