// tools/ubench/pinned_copy.hip -- how fast can 1 GB of recorded draws leave the device, and what does each way of preparing the destination cost?
// (round-5 review: sample() copies 1.05 GB at ~5 GB/s.)  hipcc --offload-arch=gfx950 -O2 tools/ubench/pinned_copy.hip -o /tmp/pinned_copy -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <sys/mman.h>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main(int argc, char **argv) {
  const size_t bytes = (argc > 1 ? atol(argv[1]) : 1024) * (size_t)1 << 20;
  printf("host threads available: %u\n", std::thread::hardware_concurrency());
  void *d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 1, bytes)); CK(hipDeviceSynchronize());
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  double t0, t1;
  // 1. pageable, untouched (calloc-like: what napi_create_arraybuffer hands out)
  { char *h = (char *)calloc(bytes, 1); t0 = now(); CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); t1 = now();
    printf("pageable untouched   : %7.1f ms  %6.2f GB/s\n", t1 - t0, bytes / (t1 - t0) * 1e-6);
    t0 = now(); CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); t1 = now();
    printf("pageable resident    : %7.1f ms  %6.2f GB/s\n", t1 - t0, bytes / (t1 - t0) * 1e-6);
    // 2. hipHostRegister of resident pageable memory, copy, unregister
    t0 = now(); CK(hipHostRegister(h, bytes, hipHostRegisterDefault)); t1 = now(); printf("hipHostRegister (resident pages): %7.1f ms\n", t1 - t0);
    t0 = now(); CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); t1 = now();
    printf("registered copy      : %7.1f ms  %6.2f GB/s\n", t1 - t0, bytes / (t1 - t0) * 1e-6);
    t0 = now(); CK(hipHostUnregister(h)); t1 = now(); printf("hipHostUnregister    : %7.1f ms\n", t1 - t0);
    free(h); }
  { char *h = (char *)calloc(bytes, 1); t0 = now(); CK(hipHostRegister(h, bytes, hipHostRegisterDefault)); t1 = now(); printf("hipHostRegister (untouched pages): %7.1f ms\n", t1 - t0);
    t0 = now(); CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); t1 = now();
    printf("registered copy      : %7.1f ms  %6.2f GB/s\n", t1 - t0, bytes / (t1 - t0) * 1e-6);
    CK(hipHostUnregister(h)); free(h); }
  // 3. hipHostMalloc
  { void *h; t0 = now(); CK(hipHostMalloc(&h, bytes, hipHostMallocDefault)); t1 = now(); printf("hipHostMalloc        : %7.1f ms\n", t1 - t0);
    t0 = now(); CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); t1 = now();
    printf("pinned copy (1st)    : %7.1f ms  %6.2f GB/s\n", t1 - t0, bytes / (t1 - t0) * 1e-6);
    t0 = now(); CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); t1 = now();
    printf("pinned copy (2nd)    : %7.1f ms  %6.2f GB/s\n", t1 - t0, bytes / (t1 - t0) * 1e-6);
    // 4. CPU memcpy from pinned into untouched / resident pageable memory with T threads
    for (int T : {1, 2, 4, 8}) {
      char *dst = (char *)calloc(bytes, 1);
      auto run = [&](const char *what) {
        t0 = now();
        std::vector<std::thread> th;
        for (int k = 0; k < T; ++k) th.emplace_back([&, k] { const size_t a = bytes / T * k, b = k == T - 1 ? bytes : bytes / T * (k + 1); memcpy(dst + a, (char *)h + a, b - a); });
        for (auto &q : th) q.join();
        t1 = now();
        printf("memcpy pinned -> %s, %d threads: %7.1f ms  %6.2f GB/s\n", what, T, t1 - t0, bytes / (t1 - t0) * 1e-6);
      };
      run("untouched"); run("resident ");
      free(dst);
    }
    t0 = now(); CK(hipHostFree(h)); t1 = now(); printf("hipHostFree          : %7.1f ms\n", t1 - t0); }
  // 5. many 64 MB pinned chunks
  { t0 = now(); std::vector<void *> v; for (size_t o = 0; o < bytes; o += (size_t)64 << 20) { void *h; CK(hipHostMalloc(&h, (size_t)64 << 20, hipHostMallocDefault)); v.push_back(h); } t1 = now();
    printf("hipHostMalloc in 64 MB chunks: %7.1f ms\n", t1 - t0); for (void *h : v) CK(hipHostFree(h)); }
  return 0;
}
