// development: what the tile stream of the persistent streaming kernels (one block per CU, direct global -> LDS loads of 1 KB per wave
// instruction, NBUF buffers, counted vmcnt waits: hop1_stream / dx_stream3 / tok_pair2 / dpost_pair) reaches WITHOUT any arithmetic, as a
// function of the tile size, the number of buffers and waves, the barrier and the tile-to-block assignment -- beside hbm_probe.hip's plain
// reader (ordinary loads, thousands of waves).
//   build (here):   hipcc --offload-arch=gfx950 -O3 scripts/lds_stream_probe.hip -o avmoe_amd/lib/variants/lds_stream_probe
//   run (GPU box):  avmoe_amd/lib/variants/lds_stream_probe [MiB, default 480] [1 = pseudo-random contents instead of a constant byte]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// NW waves, PPW 1 KB pieces per wave and tile (tile = NW PPW KB), NBUF buffers: NBUF - 1 tiles in flight after each request
template <int NW, int PPW, int NBUF, bool BARRIER, bool ROUND_ROBIN>
__global__ void __launch_bounds__(64 * NW, 1) k_stream(const char* __restrict__ src, long ntiles, unsigned* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TILE = NW * PPW * 1024;
  static_assert((NBUF - 1) * PPW <= 63, "vmcnt");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  long t0, t1, ts;
  if (ROUND_ROBIN) { t0 = blockIdx.x; t1 = ntiles; ts = gridDim.x; }
  else { t0 = ntiles * blockIdx.x / gridDim.x; t1 = ntiles * (blockIdx.x + 1) / gridDim.x; ts = 1; }
  const unsigned off = (unsigned)(wave * PPW * 1024 + lane * 16);
  auto gload = [&](int buf, long tile) {
    const char* base = src + tile * TILE;
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(base + (off + 1024u * i)), (lptr_t)(smem + buf * TILE + (wave * PPW + i) * 1024), 16, 0, 0);
  };
  long tile = t0;
#pragma unroll
  for (int j = 0; j < NBUF - 1; ++j)
    if (tile + j * ts < t1) gload(j, tile + j * ts);
  unsigned acc = 0;
  for (int it = 0; tile < t1; ++it, tile += ts) {
    const long left = (t1 - 1 - tile) / ts;                 // tiles requested after this one
    if (left >= NBUF - 2) wait_vm<(NBUF - 2) * PPW>();
    else wait_vm<0>();
    if (BARRIER) __builtin_amdgcn_s_barrier();
    acc += *(const unsigned*)(smem + (it % NBUF) * TILE + threadIdx.x * 4);      // (one LDS word per lane: the tile is "used")
    if (BARRIER && NBUF == 2) __builtin_amdgcn_s_barrier();
    if (tile + (NBUF - 1) * ts < t1) gload((it + NBUF - 1) % NBUF, tile + (NBUF - 1) * ts);
  }
  if (acc == 0x12345u) out[blockIdx.x] = 1u;
}


// hop1_stream.hip::kk_hop1_yk's tile loop rebuilt step by step (8 waves, 32 tokens x 1536 bytes, three buffers):
//   WORK bit 0: its address arithmetic (slot -> row, swizzled chunk, 64-bit row offset)     bit 1: the 24 fragment reads per wave (ds_read_b128)
//        bit 2: the 24 matrix instructions on them                                          bit 3: one 8-byte store per lane and tile
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
template <int WORK, int AUX, int STP>
__global__ void __launch_bounds__(512, 1) k_yk(const char* __restrict__ src, long ntiles, long ldy, char* __restrict__ dst, unsigned* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = 8, CH = 96, RB = 16 * CH, NP = 48, BUF = NP * 1024, NLO = 6, KS = 24;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int rt = wave % 4, th = wave / 4;
  auto gload = [&](int buf, long tile) {
    const long m0 = tile * 32;
    char* d = smem + buf * BUF + 1024 * wave;
#pragma unroll
    for (int i = 0; i < NLO; ++i) {
      if (WORK & 1) {
        const int slot = 64 * (wave + NW * i) + lane, row = slot / CH, cc = (slot % CH) ^ (row & 15);
        __builtin_amdgcn_global_load_lds((gptr_t)(src + ((m0 + min(row, 31)) * ldy + cc * 8) * 2), (lptr_t)(d + 1024 * NW * i), 16, 0, AUX);
      } else {
        __builtin_amdgcn_global_load_lds((gptr_t)(src + m0 * 1536 + (unsigned)(1024 * (wave + NW * i) + lane * 16)), (lptr_t)(d + 1024 * NW * i), 16, 0, AUX);
      }
    }
  };
  long tile = ntiles * blockIdx.x / gridDim.x;
  const long t_end = ntiles * (blockIdx.x + 1) / gridDim.x;
  if (tile >= t_end) return;
  bf16x8 af[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) af[ks] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)ks, (unsigned)lane, 3u, 4u});
  gload(0, tile);
  if (tile + 1 < t_end) gload(1, tile + 1);
  int yo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) yo[j] = (16 * th + r) * RB + 64 * (j ^ (r >> 2)) + 16 * (q ^ (r & 3));
  u32x4 xacc = {0u, 0u, 0u, 0u};
  for (int it = 0; tile < t_end; ++it, ++tile) {
    if (tile + 1 < t_end) {
      if (it == 0) wait_vm<NLO>();
      else if (it == 1) wait_vm<NLO + ((WORK & 8) ? 1 : 0)>();
      else wait_vm<NLO + ((WORK & 8) ? 2 : 0)>();
    } else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* sY = smem + (it % 3) * BUF;
    if (tile + 2 < t_end) gload((it + 2) % 3, tile + 2);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (WORK & 2) {
#pragma unroll
      for (int ks = 0; ks < KS; ks += 2) {
        const bf16x8 y0 = *(const bf16x8*)(sY + yo[ks & 3] + 256 * (ks >> 2)), y1 = *(const bf16x8*)(sY + yo[(ks + 1) & 3] + 256 * ((ks + 1) >> 2));
        if (WORK & 4) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y0, af[ks], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y1, af[ks + 1], acc1, 0, 0, 0);
        } else {
          xacc ^= __builtin_bit_cast(u32x4, y0) ^ __builtin_bit_cast(u32x4, y1);
        }
      }
    } else {
      xacc[0] += *(const unsigned*)(sY + tid * 4);
    }
    if (WORK & 8) {
      const f32x4 a = acc0 + acc1;
      const long e = (long)(16 * rt + r) * (ntiles * 32) + tile * 32 + 16 * th + 4 * q;      // [row][token]: the layout of R
      const u32x2 v = u32x2{__builtin_bit_cast(unsigned, a[0]) ^ xacc[0], __builtin_bit_cast(unsigned, a[1]) ^ xacc[1]};
      if (STP == 0) *(u32x2*)(dst + e * 2) = v;
      else if (STP == 1) __builtin_nontemporal_store(v, (u32x2*)(dst + e * 2));
      else if (STP == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(dst + e * 2), "v"(v) : "memory");
      else if (STP == 3) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(dst + e * 2), "v"(v) : "memory");
      else asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(dst + e * 2), "v"(v) : "memory");
    } else {
      xacc[1] ^= __builtin_bit_cast(unsigned, acc0[0] + acc1[1]);
    }
  }
  if ((xacc[0] ^ xacc[1] ^ xacc[2] ^ xacc[3]) == 0x12345u) out[blockIdx.x] = 1u;
}
template <int WORK, int AUX = 0, int STP = 0>
int run_yk(const char* a, long bytes, char* dst, unsigned* out, int cus, hipEvent_t e0, hipEvent_t e1) {
  constexpr int LDS = 3 * 48 * 1024;
  auto fn = k_yk<WORK, AUX, STP>;
  if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return 1;
  const long ntiles = bytes / (48 * 1024);
  double best = 0;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(fn, dim3(cus), dim3(512), LDS, 0, a, ntiles, 768L, dst, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    const double gbs = (double)ntiles * 48 * 1024 / (ms * 1e-3) / 1e9;
    if (rep > 0 && gbs > best) best = gbs;
  }
  printf("yk loop: load aux %d store policy %d %s%s%s%s  %6.0f GB/s\n", AUX, STP, (WORK & 1) ? "[addresses]" : "[plain addresses]", (WORK & 2) ? "[24 fragment reads]" : "", (WORK & 4) ? "[24 MFMA]" : "", (WORK & 8) ? "[store]" : "", best);
  return 0;
}

// dx_stream3's memory side alone: four waves, 52 KB tiles (13 pieces per wave) x 3 buffers, and per 32-token tile the 32 x 768-byte rows of
// one channel group written by the four waves (wave w: bytes 192 w .. 192 w + 191 of every row) as
//   SEG 32: six instructions per 16-token slab, a lane 8 bytes, four lanes = 32 contiguous bytes of a row (the kernel's layout)
//   SEG 64: three instructions per slab, a lane 16 bytes, four lanes = 64 contiguous bytes
//   SEG 0 : whole rows by one wave (a lane 16 bytes, 48 lanes = one row's 768 bytes): what a transposition through LDS would give
template <int SEG, int AUX>
__global__ void __launch_bounds__(256, 1) k_dx3mem(const char* __restrict__ src, long ntiles, char* __restrict__ dst, unsigned* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = 4, PPW = 13, TILE = NW * PPW * 1024, NS = SEG == 32 ? 12 : 6;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, q = lane >> 4;
  const unsigned off = (unsigned)(wave * PPW * 1024 + lane * 16);
  auto gload = [&](int buf, long tile) {
    const char* base = src + tile * TILE;
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(base + (off + 1024u * i)), (lptr_t)(smem + buf * TILE + (wave * PPW + i) * 1024), 16, 0, AUX);
  };
  long tile = ntiles * blockIdx.x / gridDim.x;
  const long t_end = ntiles * (blockIdx.x + 1) / gridDim.x;
  if (tile >= t_end) return;
  gload(0, tile);
  if (tile + 1 < t_end) gload(1, tile + 1);
  unsigned acc = 0;
  for (int it = 0; tile < t_end; ++it, ++tile) {
    const int nl = tile + 1 < t_end ? PPW : 0, ns = it == 0 ? 0 : (it == 1 ? NS : 2 * NS);
    const int n = nl + ns;
    if (n >= 37) wait_vm<37>(); else if (n >= 25) wait_vm<25>(); else if (n >= 19) wait_vm<19>(); else if (n >= 13) wait_vm<13>(); else if (n >= 12) wait_vm<12>(); else if (n >= 6) wait_vm<6>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (tile + 2 < t_end) gload((it + 2) % 3, tile + 2);
    acc += *(const unsigned*)(smem + (it % 3) * TILE + threadIdx.x * 4);
    char* o = dst + tile * 32 * 1536;                          // 32 rows of 1536 bytes (two groups; this block writes group 0's 768)
    const u32x4 v = {acc, (unsigned)lane, 3u, 4u};
    if (SEG == 32) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) *(u32x2*)(o + (16 * h + r) * 1536 + 192 * wave + 32 * ct + 8 * q) = u32x2{v[0], v[1]};
    } else if (SEG == 64) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int cp = 0; cp < 3; ++cp) *(u32x4*)(o + (16 * h + r) * 1536 + 192 * wave + 64 * cp + 16 * q) = v;
    } else {
      // 32 rows x 48 chunks = 1536 chunks = 24 instructions of 64 lanes: 6 per wave
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int slot = 64 * (wave + 4 * i) + lane, row = slot / 48, cc = slot % 48;
        *(u32x4*)(o + row * 1536 + cc * 16) = v;
      }
    }
  }
  if (acc == 0x12345u) out[blockIdx.x] = 1u;
}
template <int SEG, int AUX>
int run_dx3mem(const char* a, long bytes, char* dst, unsigned* out, int cus, hipEvent_t e0, hipEvent_t e1) {
  constexpr int TILE = 52 * 1024, LDS = 3 * TILE;
  auto fn = k_dx3mem<SEG, AUX>;
  if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return 1;
  const long ntiles = bytes / TILE;
  double best = 0;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(fn, dim3(cus), dim3(256), LDS, 0, a, ntiles, dst, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    const double gbs = (double)ntiles * (TILE + 32 * 768) / (ms * 1e-3) / 1e9;
    if (rep > 0 && gbs > best) best = gbs;
  }
  printf("dx_stream3 memory side: 52 KB read + 24 KB written per tile, stores as %s, load aux %d  %6.0f GB/s\n", SEG == 32 ? "32-byte runs (6 x 2 per wave)" : SEG == 64 ? "64-byte runs (3 x 2 per wave)" : "whole rows (6 per wave)       ", AUX, best);
  return 0;
}

// the group split of the X-side kernels: grid (cus / 2, 2), block (x, g) streams bytes [768 g, 768 g + 768) of every 1536-byte row of its row
// range (32-row tiles = 24 pieces, 8 waves x 3, four buffers) -- against the same bytes as whole rows (grid cus, 16-row tiles)
template <bool SPLIT>
__global__ void __launch_bounds__(512, 1) k_halfrows(const char* __restrict__ src, long nrows, unsigned* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = 8, PPW = 3, TILE = NW * PPW * 1024, NBUF = 4;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rpt = SPLIT ? 32 : 16;                                  // rows per tile
  const long ntiles = nrows / rpt;
  const long t0 = ntiles * blockIdx.x / gridDim.x, t1 = ntiles * (blockIdx.x + 1) / gridDim.x;
  unsigned voff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int slot = 64 * (wave * PPW + i) + lane;                  // 16-byte chunk of the tile
    if (SPLIT) { const int row = slot / 48, cc = slot % 48; voff[i] = (unsigned)(row * 1536 + 768 * blockIdx.y + cc * 16); }
    else voff[i] = (unsigned)(slot * 16);
  }
  auto gload = [&](int buf, long tile) {
    const char* base = src + tile * rpt * 1536;
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(base + voff[i]), (lptr_t)(smem + buf * TILE + (wave * PPW + i) * 1024), 16, 0, 0);
  };
  long tile = t0;
#pragma unroll
  for (int j = 0; j < NBUF - 1; ++j)
    if (tile + j < t1) gload(j, tile + j);
  unsigned acc = 0;
  for (int it = 0; tile < t1; ++it, ++tile) {
    if (t1 - 1 - tile >= NBUF - 2) wait_vm<(NBUF - 2) * PPW>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    acc += *(const unsigned*)(smem + (it % NBUF) * TILE + threadIdx.x * 4);
    if (tile + NBUF - 1 < t1) gload((it + NBUF - 1) % NBUF, tile + NBUF - 1);
  }
  if (acc == 0x12345u) out[blockIdx.x] = 1u;
}
template <bool SPLIT>
int run_halfrows(const char* a, long bytes, unsigned* out, int cus, hipEvent_t e0, hipEvent_t e1) {
  constexpr int LDS = 4 * 24 * 1024;
  auto fn = k_halfrows<SPLIT>;
  if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return 1;
  const long nrows = bytes / 1536 / 32 * 32;
  double best = 0;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(fn, SPLIT ? dim3(cus / 2, 2) : dim3(cus), dim3(512), LDS, 0, a, nrows, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    const double gbs = (double)nrows * 1536 / (ms * 1e-3) / 1e9;
    if (rep > 0 && gbs > best) best = gbs;
  }
  printf("rows of 1536 bytes, 24 KB tiles x 4: %s  %6.0f GB/s\n", SPLIT ? "two blocks per row range, 768 bytes of every row each (the group split)" : "whole rows                                                            ", best);
  return 0;
}

__global__ void k_fill_random(unsigned* d, long n) {          // pseudo-random words (a constant fill flatters the memory system: fewer toggles, less power)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    d[i] = (x & 0x7fff7fffu) % 0x3f803f80u;                    // (finite bf16 pairs)
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NW, int PPW, int NBUF, bool BARRIER, bool RR>
int run(const char* a, long bytes, unsigned* out, int cus, hipEvent_t e0, hipEvent_t e1) {
  constexpr int TILE = NW * PPW * 1024, LDS = TILE * NBUF;
  auto fn = k_stream<NW, PPW, NBUF, BARRIER, RR>;
  CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  const long ntiles = bytes / TILE;
  double best = 0;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(fn, dim3(cus), dim3(64 * NW), LDS, 0, a, ntiles, out);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
    const double gbs = (double)ntiles * TILE / (ms * 1e-3) / 1e9;
    if (rep > 0 && gbs > best) best = gbs;
  }
  printf("waves %2d  tile %3d KB  buffers %d (%3d KB LDS, %3d KB in flight)  %s %s  %6.0f GB/s\n", NW, TILE / 1024, NBUF, LDS / 1024, (NBUF - 1) * TILE / 1024,
         BARRIER ? "barrier" : "no barr", RR ? "round-robin" : "contiguous ", best);
  return 0;
}

int main(int argc, char** argv) {
  const long mib = argc > 1 ? atol(argv[1]) : 480;
  const long bytes = mib << 20;
  char* a; unsigned* out;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&out, 1 << 20));
  const bool rnd = argc > 2 && atoi(argv[2]) != 0;
  if (rnd) { hipLaunchKernelGGL(k_fill_random, dim3(4096), dim3(256), 0, 0, (unsigned*)a, bytes / 4); CK(hipDeviceSynchronize()); }
  else CK(hipMemset(a, 1, bytes));
  printf("buffer filled with %s\n", rnd ? "pseudo-random words" : "a constant byte");
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%ld MiB, %d blocks (one per CU)\n", mib, cus);
  // the shapes the kernels use
  run<8, 6, 3, true, false>(a, bytes, out, cus, e0, e1);        // hop1_yk: 48 KB x 3
  run<8, 6, 3, true, true>(a, bytes, out, cus, e0, e1);
  run<8, 6, 3, false, false>(a, bytes, out, cus, e0, e1);
  run<4, 13, 3, true, false>(a, bytes, out, cus, e0, e1);       // dx_stream3: 52 KB x 3
  run<8, 6, 2, true, false>(a, bytes, out, cus, e0, e1);        // double buffering
  // smaller tiles, more of them in flight
  run<8, 3, 6, true, false>(a, bytes, out, cus, e0, e1);        // 24 KB x 6
  run<8, 3, 6, false, false>(a, bytes, out, cus, e0, e1);
  run<8, 2, 9, true, false>(a, bytes, out, cus, e0, e1);        // 16 KB x 9
  run<8, 2, 9, true, true>(a, bytes, out, cus, e0, e1);
  run<8, 1, 18, true, false>(a, bytes, out, cus, e0, e1);       // 8 KB x 18
  run<4, 6, 6, true, false>(a, bytes, out, cus, e0, e1);        // four waves
  run<4, 3, 12, true, false>(a, bytes, out, cus, e0, e1);
  run<16, 3, 3, true, false>(a, bytes, out, cus, e0, e1);       // sixteen waves, 48 KB x 3
  run<16, 1, 9, true, false>(a, bytes, out, cus, e0, e1);       // 16 KB x 9
  run<16, 1, 9, true, true>(a, bytes, out, cus, e0, e1);
  // less in flight
  run<8, 2, 3, true, false>(a, bytes, out, cus, e0, e1);        // 16 KB x 3
  run<8, 2, 5, true, false>(a, bytes, out, cus, e0, e1);        // 16 KB x 5
  run<8, 2, 7, true, false>(a, bytes, out, cus, e0, e1);        // 16 KB x 7
  char* dst; CK(hipMalloc(&dst, 64L * (bytes / 1536) * 2 + 4096));
  run_yk<0>(a, bytes, dst, out, cus, e0, e1);
  run_yk<1>(a, bytes, dst, out, cus, e0, e1);
  run_yk<2>(a, bytes, dst, out, cus, e0, e1);
  run_yk<3>(a, bytes, dst, out, cus, e0, e1);
  run_yk<7>(a, bytes, dst, out, cus, e0, e1);
  run_yk<8>(a, bytes, dst, out, cus, e0, e1);
  run_yk<9>(a, bytes, dst, out, cus, e0, e1);
  run_yk<15>(a, bytes, dst, out, cus, e0, e1);
  run_yk<14>(a, bytes, dst, out, cus, e0, e1);
  // cache policies: loads aux = sc0 (1) / sc1 (2) / nt (4) bits; stores plain / nontemporal / sc0 sc1 / sc1 / sc0
  run_yk<15, 1>(a, bytes, dst, out, cus, e0, e1);
  run_yk<15, 2>(a, bytes, dst, out, cus, e0, e1);
  run_yk<15, 3>(a, bytes, dst, out, cus, e0, e1);
  run_yk<15, 4>(a, bytes, dst, out, cus, e0, e1);
  run_yk<15, 5>(a, bytes, dst, out, cus, e0, e1);
  run_yk<15, 6>(a, bytes, dst, out, cus, e0, e1);
  run_yk<15, 7>(a, bytes, dst, out, cus, e0, e1);
  run_yk<15, 0, 1>(a, bytes, dst, out, cus, e0, e1);
  run_yk<15, 0, 2>(a, bytes, dst, out, cus, e0, e1);
  run_yk<15, 0, 3>(a, bytes, dst, out, cus, e0, e1);
  run_yk<15, 0, 4>(a, bytes, dst, out, cus, e0, e1);
  run_yk<15, 4, 1>(a, bytes, dst, out, cus, e0, e1);
  run_yk<15, 6, 1>(a, bytes, dst, out, cus, e0, e1);
  char* dst2; CK(hipMalloc(&dst2, (bytes / (52 * 1024) + 1) * 32 * 1536));
  run_dx3mem<32, 0>(a, bytes, dst2, out, cus, e0, e1);
  run_dx3mem<32, 2>(a, bytes, dst2, out, cus, e0, e1);
  run_dx3mem<64, 0>(a, bytes, dst2, out, cus, e0, e1);
  run_dx3mem<64, 2>(a, bytes, dst2, out, cus, e0, e1);
  run_dx3mem<0, 0>(a, bytes, dst2, out, cus, e0, e1);
  run_dx3mem<0, 2>(a, bytes, dst2, out, cus, e0, e1);
  run_halfrows<false>(a, bytes, out, cus, e0, e1);
  run_halfrows<true>(a, bytes, out, cus, e0, e1);
  return 0;
}
