// development: what a plain streaming kernel reaches on this box -- the yardstick beside MI355X_MICROARCH.md's 8 TB/s peak that
// bench.py's roofline fractions are quoted against.  16-byte accesses per lane, persistent blocks, buffers far beyond the caches.
//   read   : sum of a buffer (one partial per block)          write : fill
//   copy   : dst = src (1 : 1)                                 mix21 : dst = a + b (2 reads : 1 write, the shape of an accumulating GEMM epilogue)
//   inplace: dst += a (2 reads : 1 write, the write on the row just read)      seg : the read as 64-byte segments of 512-byte rows, a block's
//            four waves taking the segments of the same 16 rows (the request shape of the bottleneck-space kernels); seg 2r1w : the memory
//            side of kf_mid_bwd without its arithmetic (two segment reads, the segment write in place)
//   build (here, no GPU needed):  hipcc --offload-arch=gfx950 -O3 scripts/hbm_probe.hip -o avmoe_amd/lib/variants/hbm_probe
//   run (GPU box):                avmoe_amd/lib/variants/hbm_probe [MiB per buffer, default 1024]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__global__ void __launch_bounds__(256) k_read(const u32x4* __restrict__ a, long n, unsigned* __restrict__ out) {
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const u32x4 v = a[i]; acc ^= v; }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[blockIdx.x] = 1u;      // (keeps the loads alive)
}
__global__ void __launch_bounds__(256) k_write(u32x4* __restrict__ d, long n) {
  const u32x4 v = {1u, 2u, 3u, (unsigned)blockIdx.x};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) d[i] = v;
}
__global__ void __launch_bounds__(256) k_copy(const u32x4* __restrict__ a, u32x4* __restrict__ d, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) d[i] = a[i];
}
__global__ void __launch_bounds__(256) k_inplace(const u32x4* __restrict__ a, u32x4* d, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) d[i] = d[i] + a[i];
}
// rows of 512 bytes read as 64-byte segments: wave w of a block takes segment (w & 3) [and (w & 3) + 4] of 16 consecutive rows per step
// (the request shape of the bottleneck-space kernels: lane = (row r, quarter q))
__global__ void __launch_bounds__(256) k_seg(const u32x4* __restrict__ a, long n, unsigned* __restrict__ out) {
  u32x4 acc = {0u, 0u, 0u, 0u};
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const long rows = n / 32;                                  // 32 x 16 bytes per row
  for (long r0 = (long)blockIdx.x * 16; r0 < rows; r0 += (long)gridDim.x * 16) {
    const u32x4* row = a + (r0 + r) * 32;
    acc ^= row[wave * 4 + q]; acc ^= row[16 + wave * 4 + q];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[blockIdx.x] = 1u;
}
// the memory side of kf_mid_bwd without its arithmetic: two segment reads (a, d) and the segment write back into d
__global__ void __launch_bounds__(256) k_seg3(const u32x4* __restrict__ a, u32x4* d, long n) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const long rows = n / 32;
  for (long r0 = (long)blockIdx.x * 16; r0 < rows; r0 += (long)gridDim.x * 16) {
    const long o = (r0 + r) * 32 + wave * 4 + q;
    const u32x4 x0 = a[o], x1 = a[o + 16], y0 = d[o], y1 = d[o + 16];
    d[o] = x0 + y0; d[o + 16] = x1 + y1;
  }
}
__global__ void __launch_bounds__(256) k_mix21(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ d, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) d[i] = a[i] + b[i];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const long mib = argc > 1 ? atol(argv[1]) : 1024;
  const long bytes = mib << 20, n = bytes / 16;
  u32x4 *a, *b, *d; unsigned* out;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&d, bytes)); CK(hipMalloc(&out, 1 << 20));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes)); CK(hipMemset(d, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grids[] = {256, 512, 768, 1024, 2048, 4096, 16384};
  printf("%ld MiB per buffer; GB/s of bytes moved (best of 5 per grid)\n%8s %10s %10s %10s %10s %10s %10s %10s\n", mib, "blocks", "read", "write", "copy", "mix 2:1", "in place", "seg read", "seg 2r1w");
  for (int g : grids) {
    double best[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int rep = 0; rep < 6; ++rep) {
      for (int k = 0; k < 7; ++k) {
        CK(hipEventRecord(e0, 0));
        if (k == 0) hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, a, n, out);
        else if (k == 1) hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, d, n);
        else if (k == 2) hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, d, n);
        else if (k == 3) hipLaunchKernelGGL(k_mix21, dim3(g), dim3(256), 0, 0, a, b, d, n);
        else if (k == 4) hipLaunchKernelGGL(k_inplace, dim3(g), dim3(256), 0, 0, a, d, n);
        else if (k == 5) hipLaunchKernelGGL(k_seg, dim3(g), dim3(256), 0, 0, a, n, out);
        else hipLaunchKernelGGL(k_seg3, dim3(g), dim3(256), 0, 0, a, d, n);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
        const double moved = (double)bytes * (k == 0 || k == 1 || k == 5 ? 1 : (k == 2 ? 2 : 3));
        const double gbs = moved / (ms * 1e-3) / 1e9;
        if (rep > 0 && gbs > best[k]) best[k] = gbs;
      }
    }
    printf("%8d %10.0f %10.0f %10.0f %10.0f %10.0f %10.0f %10.0f\n", g, best[0], best[1], best[2], best[3], best[4], best[5], best[6]);
  }
  return 0;
}
