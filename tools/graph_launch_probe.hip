// Would hipGraph replay help the launch-bound regime (n <= 1e6: 5-7 kernels of 3-30 us per Lanczos iteration)?
// One "iteration" = 7 dependent tiny kernels + an event the host waits for one iteration later (the loop's shape).
//   A: 7 stream launches + hipEventRecord + hipEventSynchronize(previous)
//   B: the 7 kernels captured once into a graph, hipGraphLaunch per iteration (same arguments every time)
//   C: like B, plus hipGraphExecKernelNodeSetParams on all 7 nodes before every launch (pointers change per iteration)
//   D: 4 iterations (28 nodes) per graph launch, no parameter updates
// Reports host time per iteration and device busy span per iteration.
//   hipcc --offload-arch=gfx950 -O2 tools/graph_launch_probe.hip -o tools/_build/graph_launch_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                        \
  do {                                                               \
    hipError_t e = (x);                                              \
    if (e != hipSuccess) {                                           \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e));      \
      return 1;                                                      \
    }                                                                \
  } while (0)

__global__ void tiny(double* p, const double* q, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = q[i] * 1.0000001 + 1e-9;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const int n = 1 << 14, iters = 5000, K = 7;
  double *a, *b;
  CK(hipMalloc(&a, n * 8 * 8));
  CK(hipMalloc(&b, n * 8 * 8));
  CK(hipMemset(a, 0, n * 8 * 8));
  CK(hipMemset(b, 0, n * 8 * 8));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t ev[4];
  for (auto& evt : ev) CK(hipEventCreateWithFlags(&evt, hipEventDisableTiming));
  auto launch_iter = [&](int it) {
    for (int k = 0; k < K; ++k) {
      double* dst = (k & 1) ? a + (size_t)(it % 8) * n : b + (size_t)(it % 8) * n;
      const double* src = (k & 1) ? b + (size_t)(it % 8) * n : a + (size_t)(it % 8) * n;
      hipLaunchKernelGGL(tiny, dim3(n / 256), dim3(256), 0, s, dst, src, n);
    }
  };
  // ---- A
  for (int it = 0; it < 200; ++it) launch_iter(it);
  CK(hipStreamSynchronize(s));
  double t0 = now();
  for (int it = 0; it < iters; ++it) {
    launch_iter(it);
    CK(hipEventRecord(ev[it % 4], s));
    if (it > 0) CK(hipEventSynchronize(ev[(it - 1) % 4]));
  }
  CK(hipStreamSynchronize(s));
  std::printf("A stream launches            : %.2f us per iteration (%d kernels + event record + wait on the previous event)\n", (now() - t0) / iters * 1e6, K);
  // ---- capture one iteration
  hipGraph_t g;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  launch_iter(0);
  CK(hipStreamEndCapture(s, &g));
  hipGraphExec_t ge;
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  size_t nn = 0;
  CK(hipGraphGetNodes(g, nullptr, &nn));
  std::vector<hipGraphNode_t> nodes(nn);
  CK(hipGraphGetNodes(g, nodes.data(), &nn));
  for (int it = 0; it < 200; ++it) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  t0 = now();
  for (int it = 0; it < iters; ++it) {
    CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(ev[it % 4], s));
    if (it > 0) CK(hipEventSynchronize(ev[(it - 1) % 4]));
  }
  CK(hipStreamSynchronize(s));
  std::printf("B graph replay, fixed params : %.2f us per iteration (%zu nodes)\n", (now() - t0) / iters * 1e6, nn);
  // ---- C: update all nodes' params before each launch
  t0 = now();
  for (int it = 0; it < iters; ++it) {
    for (size_t k = 0; k < nn; ++k) {
      double* dst = (k & 1) ? a + (size_t)(it % 8) * n : b + (size_t)(it % 8) * n;
      const double* src = (k & 1) ? b + (size_t)(it % 8) * n : a + (size_t)(it % 8) * n;
      int nv = n;
      void* args[3] = {&dst, &src, &nv};
      hipKernelNodeParams p = {};
      p.func = (void*)tiny;
      p.gridDim = dim3(n / 256);
      p.blockDim = dim3(256);
      p.kernelParams = args;
      CK(hipGraphExecKernelNodeSetParams(ge, nodes[k], &p));
    }
    CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(ev[it % 4], s));
    if (it > 0) CK(hipEventSynchronize(ev[(it - 1) % 4]));
  }
  CK(hipStreamSynchronize(s));
  std::printf("C graph replay + SetParams   : %.2f us per iteration\n", (now() - t0) / iters * 1e6);
  // ---- D: 4 iterations per graph
  hipGraph_t g4;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int j = 0; j < 4; ++j) launch_iter(j);
  CK(hipStreamEndCapture(s, &g4));
  hipGraphExec_t ge4;
  CK(hipGraphInstantiate(&ge4, g4, nullptr, nullptr, 0));
  for (int it = 0; it < 50; ++it) CK(hipGraphLaunch(ge4, s));
  CK(hipStreamSynchronize(s));
  t0 = now();
  for (int it = 0; it < iters / 4; ++it) {
    CK(hipGraphLaunch(ge4, s));
    CK(hipEventRecord(ev[it % 4], s));
    if (it > 0) CK(hipEventSynchronize(ev[(it - 1) % 4]));
  }
  CK(hipStreamSynchronize(s));
  std::printf("D graph of 4 iterations      : %.2f us per iteration\n", (now() - t0) / (iters / 4 * 4) * 1e6);
  // device-only floor: everything enqueued up front
  t0 = now();
  for (int it = 0; it < 500; ++it) launch_iter(it);
  CK(hipStreamSynchronize(s));
  std::printf("E stream launches, no events : %.2f us per iteration\n", (now() - t0) / 500 * 1e6);
  return 0;
}
