// accuracy of v_rcp_f64 and of the one-Newton-step f64 reciprocal narrowed to float, against the full expansion (recip64_noscale);
// and of exp via v_exp_f32(x*log2e hi) * (1 + lo*ln2) against exp_1ulp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__device__ double full(double x){ double y=__builtin_amdgcn_rcp(x); y=__builtin_fma(__builtin_fma(-x,y,1.),y,y); y=__builtin_fma(__builtin_fma(-x,y,1.),y,y); const double q=y; return __builtin_fma(__builtin_fma(-x,q,1.),y,q);} 
__device__ float exp_1ulp(float x){ const float H=1.44269502162933349609375f,Lo=1.925963033500011e-8f; const float n=rintf(x*H); const float f=__builtin_fmaf(x,Lo,__builtin_fmaf(x,H,-n)); return ldexpf(__builtin_amdgcn_exp2f(f),(int)n);} 
__device__ float exp_b(float x){ const float H=1.44269502162933349609375f,Lo=1.925963033500011e-8f; const float t=x*H; const float lo=__builtin_fmaf(x,Lo,__builtin_fmaf(x,H,-t)); const float p=__builtin_amdgcn_exp2f(t); return __builtin_fmaf(p, lo*0.693147180559945f, p);} 
__global__ void k(const float* e, int n, double* maxrel_rcp, int* mism1, int* mism0, int* mismE, float* maxrelE, const float* xs){
  int i=blockIdx.x*blockDim.x+threadIdx.x; if(i>=n) return;
  double x=1.+(double)e[i];
  double y0=__builtin_amdgcn_rcp(x); double ref=full(x);
  double rel=fabs(y0-ref)/ref; atomicMax((unsigned long long*)maxrel_rcp, __double_as_longlong(rel));
  double y1=__builtin_fma(__builtin_fma(-x,y0,1.),y0,y0);
  if((float)y1!=(float)ref) atomicAdd(mism1,1);
  if((float)y0!=(float)ref) atomicAdd(mism0,1);
  float a=exp_1ulp(xs[i]), b=exp_b(xs[i]);
  if(a!=b) atomicAdd(mismE,1);
  float r=fabsf(a-b)/fmaxf(a,1e-37f); atomicMax((unsigned*)maxrelE, __float_as_uint(r));
}
int main(){ const int n=1<<24; std::vector<float> e(n),xs(n); srand(1); for(int i=0;i<n;i++){ double u=rand()/(double)RAND_MAX; xs[i]=(float)(-100.+u*180.); if(i&1) xs[i]=(float)(-12.+u*24.); e[i]=expf((float)(u*40.-20.)); }
  float *de,*dx,*dm; double* dr; int* dc; hipMalloc(&de,n*4); hipMalloc(&dx,n*4); hipMalloc(&dr,8); hipMalloc(&dc,12); hipMalloc(&dm,4);
  hipMemcpy(de,e.data(),n*4,hipMemcpyHostToDevice); hipMemcpy(dx,xs.data(),n*4,hipMemcpyHostToDevice); hipMemset(dr,0,8); hipMemset(dc,0,12); hipMemset(dm,0,4);
  k<<<n/256,256>>>(de,n,dr,dc,dc+1,dc+2,dm,dx); double r; int c[3]; float m; hipMemcpy(&r,dr,8,hipMemcpyDeviceToHost); hipMemcpy(c,dc,12,hipMemcpyDeviceToHost); hipMemcpy(&m,dm,4,hipMemcpyDeviceToHost);
  printf("n %d  v_rcp_f64 max rel err %.3e  float mismatches: one Newton step %d, none %d;  exp variants differ %d times, max rel %.3e\n", n, r, c[0], c[1], c[2], m); }
