// Does a large hipMalloc on one host thread stall kernel launches of another?  (Round 5: a "cold" device allocation costs
// ~28 ms per GB on this system -- 0.5 s for a 17 GB column-scratch segment -- inside whatever iteration needs it; could a
// helper thread make it ahead of need, behind the pass that is running?)
//   hipcc --offload-arch=gfx950 -O2 -pthread -o malloc_overlap malloc_overlap.hip && ./malloc_overlap [GB]
// Prints the duration of every kernel of a stream of ~5 ms kernels and the wall time between their completions while a
// second thread allocates (and frees) GB-sized buffers.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void k_spin(double *a, int iters) {
  double x = a[threadIdx.x + blockIdx.x * blockDim.x];
  for (int i = 0; i < iters; i++) x = x * 1.0000001 + 1e-9;
  a[threadIdx.x + blockIdx.x * blockDim.x] = x;
}

int main(int argc, char **argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 40.0;
  double *a;
  hipMalloc(&a, sizeof(double) * 256 * 1024 * 64);
  hipMemset(a, 0, sizeof(double) * 256 * 1024 * 64);
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  std::atomic<int> phase{0};
  std::vector<double> t_alloc;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  std::thread helper([&] {
    while (phase.load() == 0) std::this_thread::yield();
    for (int r = 0; r < 3; r++) {
      void *p = nullptr;
      const double s = now();
      hipError_t e = hipMalloc(&p, (size_t)(gb * 1e9));
      const double m = now();
      t_alloc.push_back(s - t0);
      t_alloc.push_back(m - s);
      printf("helper: hipMalloc of %.0f GB #%d: %s in %.3f s (started at %.3f s)\n", gb, r, hipGetErrorString(e), m - s, s - t0);
      std::this_thread::sleep_for(std::chrono::milliseconds(300));
      // keep it: the next one is cold again
    }
    phase = 2;
  });
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  double last = now();
  for (int i = 0; i < 2000 && phase.load() != 2; i++) {
    if (i == 40) phase = 1;
    hipEventRecord(e0, st);
    hipLaunchKernelGGL(k_spin, dim3(256 * 64), dim3(256), 0, st, a, 40000);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double t = now();
    if (i < 5 || ms > 8.0 || (t - last) > 0.012 || i % 50 == 0) printf("kernel %4d at %.3f s: %.2f ms on the device, %.2f ms since the last one ended\n", i, t - t0, ms, 1e3 * (t - last));
    last = t;
  }
  helper.join();
  return 0;
}
