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

// Hand-rolled barrier: one agent-scope counter, thread 0 of every workgroup adds and polls (bounded, so that a grid
// that is not fully resident cannot hang the device), then a workgroup barrier.
__global__ __launch_bounds__(256) void custom_sync_loop(int nsync, unsigned* counter, double* out, unsigned* timeouts) {
  double acc = 0.0;
  for (int i = 0; i < nsync; ++i) {
    acc += (double)i;
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      const unsigned target = (unsigned)(i + 1) * gridDim.x;
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      long spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > 2000000) {
          atomicAdd(timeouts, 1u);
          break;
        }
      }
    }
    __syncthreads();
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
  unsigned *d_cnt, *d_to;
  if (hipMalloc(&d_cnt, 4) != hipSuccess || hipMalloc(&d_to, 4) != hipSuccess) return 1;
  for (int grid : {256, 512, 1024}) {
    for (int nsync : {1, 101}) {
      float best = 1e30f;
      unsigned timeouts = 0;
      for (int rep = 0; rep < 4; ++rep) {
        (void)hipMemset(d_cnt, 0, 4);
        (void)hipMemset(d_to, 0, 4);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(custom_sync_loop, dim3(grid), dim3(256), 0, 0, nsync, d_cnt, d_out, d_to);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(&timeouts, d_to, 4, hipMemcpyDeviceToHost);
        if (rep) best = ms < best ? ms : best;
      }
      std::printf("custom barrier, grid %4d x 256 threads, %3d syncs: %.1f us (timeouts %u)\n", grid, nsync, best * 1e3, timeouts);
    }
  }
  return 0;
}
