// Probe: what a chain of dependent launches costs on one stream vs the same chain replayed from a hipGraph (round 6, small-batch study).
// hipcc --offload-arch=gfx950 -O3 graph_gap_probe.hip -o graph_gap_probe && ./graph_gap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void work(float* p, int iters) {
  float v = p[blockIdx.x * blockDim.x + threadIdx.x];
  for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0001f, 0.5f);
  p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const int N = 93, blocks = 256, threads = 256;
  float* d; CHK(hipMalloc(&d, blocks * threads * 4)); CHK(hipMemset(d, 0, blocks * threads * 4));
  hipStream_t s, s2; CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t ef, ej; CHK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CHK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  for (int iters : {1, 2000, 8000}) {
    for (int fork = 0; fork < 2; ++fork) {
      auto chain = [&](hipStream_t st) {
        for (int i = 0; i < N; ++i) {
          if (fork && i == 40) { hipEventRecord(ef, st); hipStreamWaitEvent(s2, ef, 0); for (int j = 0; j < 8; ++j) hipLaunchKernelGGL(work, dim3(blocks / 2), dim3(threads), 0, s2, d + blocks * threads / 2, iters); hipEventRecord(ej, s2); }
          hipLaunchKernelGGL(work, dim3(fork && i >= 40 && i < 48 ? blocks / 2 : blocks), dim3(threads), 0, st, d, iters);
          if (fork && i == 48) hipStreamWaitEvent(st, ej, 0);
        }
      };
      // stream launches, synchronous steps
      for (int w = 0; w < 5; ++w) { chain(s); CHK(hipStreamSynchronize(s)); }
      double t0 = now_us();
      const int R = 50;
      for (int r = 0; r < R; ++r) { chain(s); CHK(hipStreamSynchronize(s)); }
      double t_stream = (now_us() - t0) / R;
      // graph
      hipGraph_t g; hipGraphExec_t ge;
      CHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      chain(s);
      CHK(hipStreamEndCapture(s, &g));
      CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int w = 0; w < 5; ++w) { CHK(hipGraphLaunch(ge, s)); CHK(hipStreamSynchronize(s)); }
      t0 = now_us();
      for (int r = 0; r < R; ++r) { CHK(hipGraphLaunch(ge, s)); CHK(hipStreamSynchronize(s)); }
      double t_graph = (now_us() - t0) / R;
      printf("iters %5d fork %d: %d launches per step: stream %.1f us (%.2f per launch), graph %.1f us (%.2f per launch)\n", iters, fork, N + (fork ? 8 : 0), t_stream, t_stream / N, t_graph, t_graph / N);
      CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
    }
  }
  return 0;
}
