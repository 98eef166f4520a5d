// Internal glue shared by the translation units of libgvk.so (not part of the ABI).
#pragma once
#include <stdarg.h>

// Records a printf-style message for gvk_last_error() (thread-local) and returns `code`.
int gvk_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// CPUs this process may really use: the affinity mask capped by the container's cgroup quota (a 256-thread host seen
// from a 16-CPU container reports 256 hardware threads).
int gvk_cpu_budget(void);
