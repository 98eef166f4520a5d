"""Constants and small helpers shared by the package (the reference's libgraphvite module level:
src/graphvite.cu:62-105, include/bind.h:53-76, include/util/io.h)."""
import enum
import logging

auto = 0  # kAuto, include/util/common.h:29

logger = logging.getLogger("graphvite_amd")


class dtype(enum.IntEnum):
    uint32 = 0
    uint64 = 1
    float32 = 2
    float64 = 3


# typeid(T).name() under the Itanium ABI, which is what the reference's template names are built from
dtype2name = {dtype.uint32: "j", dtype.uint64: "m", dtype.float32: "f", dtype.float64: "d"}

uint32, uint64, float32, float64 = dtype.uint32, dtype.uint64, dtype.float32, dtype.float64


def KiB(x):
    return x << 10


def MiB(x):
    return x << 20


def GiB(x):
    return x << 30


class io(object):
    """pretty-printing helpers bound as libgraphvite.io (include/util/io.h:41-103)."""

    kBlockWidth = 40

    @staticmethod
    def size_string(size):
        size = float(size)
        if size >= 1 << 40:
            return "%.3g TiB" % (size / (1 << 40))
        if size >= 1 << 30:
            return "%.3g GiB" % (size / (1 << 30))
        if size >= 1 << 20:
            return "%.3g MiB" % (size / (1 << 20))
        if size >= 1 << 10:
            return "%.3g KiB" % (size / (1 << 10))
        return "%d B" % int(size)

    @staticmethod
    def yes_no(x):
        return "yes" if x else "no"

    @staticmethod
    def block(content):
        line = "<" * io.kBlockWidth
        return "\n%s\n%s\n%s" % (line, content, ">" * io.kBlockWidth)

    @staticmethod
    def header(content):
        pad = max(io.kBlockWidth - len(content) - 2, 0)
        return "%s %s %s" % ("-" * (pad // 2), content, "-" * (pad - pad // 2))


def init_logging(level=logging.INFO, dir="", verbose=False):
    """Counterpart of graphvite.init_logging (python/graphvite/base.py:61-83)."""
    logger.setLevel(level)
    if not logger.handlers:
        handler = logging.StreamHandler()
        logger.addHandler(handler)
    fmt = "%(levelname).1s %(asctime)s %(filename)s:%(lineno)d] %(message)s" if verbose else "%(message)s"
    for handler in logger.handlers:
        handler.setFormatter(logging.Formatter(fmt))
    if dir:
        import os
        fh = logging.FileHandler(os.path.join(dir, "graphvite_amd.log"))
        fh.setFormatter(logging.Formatter(fmt))
        logger.addHandler(fh)
    _native_logging(level)


_native_sink = None


def _native_logging(level):
    """The native runtime (include/gvx.h gvx_set_logging; glog in the reference, src/graphvite.cu:81-88) logs through this
    package's logger: messages below `level` are dropped at the source."""
    global _native_sink
    import ctypes as C
    from . import _lib
    try:
        lib = _lib.lib()
    except _lib.NativeLibraryError:
        return
    levels = {0: logging.INFO, 1: logging.WARNING, 2: logging.ERROR, 3: logging.CRITICAL}

    def sink(severity, message, user):
        logger.log(levels.get(severity, logging.INFO), (message or b"").decode("utf-8", "replace"))
    sink_type = C.CFUNCTYPE(None, C.c_int, C.c_char_p, C.c_void_p)
    _native_sink = sink_type(sink)
    lib.gvx_set_logging.restype = None
    lib.gvx_set_logging.argtypes = [C.c_int, sink_type, C.c_void_p]
    threshold = 0 if level <= logging.INFO else (1 if level <= logging.WARNING else (2 if level <= logging.ERROR else 3))
    lib.gvx_set_logging(threshold, _native_sink, None)


def cpu_budget():
    """CPUs this process can really use: the smaller of its affinity mask and its cgroup CPU quota.
    `os.cpu_count()` reports the machine (256 hardware threads on the MI355X boxes) even when the container is capped
    (e.g. cpu.max = "1600000 100000" = 16 CPUs); running 255 sampler threads under such a cap gets them throttled for
    most of every 100 ms period."""
    import math
    import os
    try:
        count = len(os.sched_getaffinity(0))
    except AttributeError:
        count = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2
            q, period = f.read().split()
            if q != "max":
                quota = float(q) / float(period)
    except (OSError, ValueError):
        try:  # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = float(f.read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        count = min(count, max(int(math.floor(quota)), 1))
    return max(count, 1)
