// ROCTx ranges around the phases of the loop (operator apply, orthogonalisation, host tridiagonal step, Ritz step), so
// that `rocprofv3 --marker-trace --kernel-trace` shows the kernels grouped per phase.
//
// The ROCTx library is bound lazily and only when it matters: if a profiler has already mapped
// librocprofiler-sdk-roctx (rocprofv3 does) the ranges go to it; otherwise LL_ROCTX=1 loads it; otherwise every call
// is a cheap no-op and the library has no link-time dependency on any tracing package.
#pragma once

#include <dlfcn.h>

#include <cstdlib>

namespace ll {

struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char* names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
    void* h = nullptr;
    for (const char* nm : names)
      if ((h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
    const char* e = std::getenv("LL_ROCTX");
    if (!h && e && std::atoi(e) != 0)
      for (const char* nm : names)
        if ((h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return;
    push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
    pop = (int (*)())dlsym(h, "roctxRangePop");
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
  static Roctx& get() {
    static Roctx r;
    return r;
  }
};

struct TraceRange {  // RAII: roctxRangePush / roctxRangePop
  bool on;
  explicit TraceRange(const char* name) : on(Roctx::get().push != nullptr) {
    if (on) Roctx::get().push(name);
  }
  ~TraceRange() {
    if (on) Roctx::get().pop();
  }
  TraceRange(const TraceRange&) = delete;
  TraceRange& operator=(const TraceRange&) = delete;
};

}  // namespace ll
