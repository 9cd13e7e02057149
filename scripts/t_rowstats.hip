#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../avmoe_amd/csrc/device_utils.h"
using namespace avmoe;
#define HEAD const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63; long row = (long)blockIdx.x * 4 + wave; if (row >= rows) return;
__global__ void k2(const float* X, long rows, int C, float* out) {   // u32x4 + bit_cast + wave_sum
  HEAD const u32x4_t* p = (const u32x4_t*)(X + row * C);
  float s = 0.f; for (int v = lane; v < C/4; v += 64) { u32x4_t w = p[v]; for (int e=0;e<4;e++) s += __builtin_bit_cast(float, w[e]); }
  s = wave_sum(s); if (lane == 0) out[row] = s;
}
__global__ void k4(const float* X, long rows, int C, float* out) {   // uint4 + __uint_as_float
  HEAD const uint4* p = (const uint4*)(X + row * C);
  float s = 0.f; for (int v = lane; v < C/4; v += 64) { uint4 w = p[v]; s += __uint_as_float(w.x)+__uint_as_float(w.y)+__uint_as_float(w.z)+__uint_as_float(w.w); }
  s = wave_sum(s); if (lane == 0) out[row] = s;
}
__global__ void k5(const float* X, long rows, int C, float* out) {   // u32x4 explicit elements
  HEAD const u32x4_t* p = (const u32x4_t*)(X + row * C);
  float s = 0.f; for (int v = lane; v < C/4; v += 64) { u32x4_t w = p[v]; s += __uint_as_float(w.x)+__uint_as_float(w.y)+__uint_as_float(w.z)+__uint_as_float(w.w); }
  s = wave_sum(s); if (lane == 0) out[row] = s;
}
__global__ void k6(const float* X, long rows, int C, float* out) {   // f32x4_t ext vector
  HEAD const f32x4_t* p = (const f32x4_t*)(X + row * C);
  float s = 0.f; for (int v = lane; v < C/4; v += 64) { f32x4_t w = p[v]; for (int e=0;e<4;e++) s += w[e]; }
  s = wave_sum(s); if (lane == 0) out[row] = s;
}
__global__ void k7(const float* X, long rows, int C, float* out) {   // u32x4 via void* (no float* origin)
  HEAD const u32x4_t* p = (const u32x4_t*)((const char*)X + row * C * 4);
  float s = 0.f; for (int v = lane; v < C/4; v += 64) { u32x4_t w = p[v]; for (int e=0;e<4;e++) s += __builtin_bit_cast(float, w[e]); }
  s = wave_sum(s); if (lane == 0) out[row] = s;
}
int main(){
  const int rows=240, C=96;
  std::vector<float> h(rows*C); for(int i=0;i<rows*C;i++) h[i]=sinf(i*0.37f);
  float *d,*o; (void)hipMalloc(&d,h.size()*4); (void)hipMalloc(&o,rows*4);
  (void)hipMemcpy(d,h.data(),h.size()*4,hipMemcpyHostToDevice);
  for (int which=2; which<=7; ++which) { if (which==3) continue;
    (void)hipMemset(o,0,rows*4);
    if(which==2) hipLaunchKernelGGL(k2, dim3(60), dim3(256), 0, 0, d, (long)rows, C, o);
    if(which==4) hipLaunchKernelGGL(k4, dim3(60), dim3(256), 0, 0, d, (long)rows, C, o);
    if(which==5) hipLaunchKernelGGL(k5, dim3(60), dim3(256), 0, 0, d, (long)rows, C, o);
    if(which==6) hipLaunchKernelGGL(k6, dim3(60), dim3(256), 0, 0, d, (long)rows, C, o);
    if(which==7) hipLaunchKernelGGL(k7, dim3(60), dim3(256), 0, 0, d, (long)rows, C, o);
    std::vector<float> r(rows); (void)hipMemcpy(r.data(),o,rows*4,hipMemcpyDeviceToHost);
    double me=0; for(int i=0;i<rows;i++){ double s=0; for(int c=0;c<C;c++) s+=h[i*C+c]; me=fmax(me,fabs(s-r[i])); }
    printf("k%d max err %g\n",which,me);
  }
}
