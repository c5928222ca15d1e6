// One-off ~85 ms stalls seen once per process in launch-bound runs (tools/stall_probe.py): is it the HIP runtime, and what
// triggers it?  Launches trivial kernels on one stream the way the Lanczos loop does (5 launches + an event record per
// "iteration", a wait on the previous iteration's event) and prints every host call that took longer than 5 ms with
// the launch count and the time since the first HIP call.
//   hipcc --offload-arch=gfx950 -O2 tools/runtime_stall_probe.hip -o tools/_build/runtime_stall_probe
//   tools/_build/runtime_stall_probe [mode]   mode 0: launches + events; 1: launches only, one sync per 38 iterations;
//                                             2: like 0 plus a 16 MB hipMalloc/hipFree every 38 iterations
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void tiny(double* p) {
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const int mode = argc > 1 ? std::atoi(argv[1]) : 0;
  const double t0 = now();
  double* d;
  hipMalloc(&d, 4096);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t ev[4];
  for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  long launches = 0;
  int stalls = 0;
  auto timed = [&](const char* what, auto&& f) {
    const double a = now();
    f();
    const double b = now();
    if (b - a > 5e-3) {
      std::printf("  mode %d: %s took %.1f ms at launch %ld, %.3f s after start\n", mode, what, (b - a) * 1e3, launches, b - t0);
      ++stalls;
    }
  };
  const int iters = argc > 2 ? std::atoi(argv[2]) : 20000;
  for (int it = 0; it < iters; ++it) {
    for (int k = 0; k < 5; ++k) {
      timed("launch", [&] { hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s, d); });
      ++launches;
    }
    if (mode != 1) {
      timed("event record", [&] { hipEventRecord(ev[it % 4], s); });
      if (it > 0) timed("event sync", [&] { hipEventSynchronize(ev[(it - 1) % 4]); });
    }
    if (it % 38 == 37) {
      timed("stream sync", [&] { hipStreamSynchronize(s); });
      if (mode == 2) {
        void* q;
        timed("hipMalloc 16 MB", [&] { hipMalloc(&q, 16 << 20); });
        timed("hipFree", [&] { hipFree(q); });
      }
    }
  }
  hipStreamSynchronize(s);
  std::printf("mode %d: %ld launches in %.2f s, %d host calls above 5 ms\n", mode, launches, now() - t0, stalls);
  return 0;
}
