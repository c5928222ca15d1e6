// Cost of a grid-wide barrier on the box at hand (cooperative launch, cooperative_groups::grid_group::sync), for the
// "one cooperative kernel per iteration" idea of DESIGN.md section 8: per-sync latency by grid size.
//   hipcc --offload-arch=gfx950 -O3 tools/gridsync_probe.hip -o /tmp/gridsync_probe && /tmp/gridsync_probe
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>

#include <cstdio>

namespace cg = cooperative_groups;

__global__ __launch_bounds__(256) void sync_loop(int nsync, double* out) {
  cg::grid_group grid = cg::this_grid();
  double acc = 0.0;
  for (int i = 0; i < nsync; ++i) {
    acc += (double)i;
    grid.sync();
  }
  if (acc < 0) out[0] = acc;
}

int main() {
  double* d_out;
  if (hipMalloc(&d_out, 8) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int grid : {256, 512, 1024, 2048}) {
    for (int nsync : {1, 101}) {
      void* args[] = {(void*)&nsync, (void*)&d_out};
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipError_t e = hipLaunchCooperativeKernel((const void*)sync_loop, dim3(grid), dim3(256), args, 0, 0);
        if (e != hipSuccess) {
          std::printf("grid %d: cooperative launch failed: %s\n", grid, hipGetErrorString(e));
          (void)hipGetLastError();
          break;
        }
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep) best = ms < best ? ms : best;
      }
      std::printf("grid %4d x 256 threads, %3d syncs: %.1f us\n", grid, nsync, best * 1e3);
    }
  }
  return 0;
}
