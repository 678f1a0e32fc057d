#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* y0, double* y2, int n) {
  int i = threadIdx.x; if (i>=n) return;
  double v = x[i];
  double y = __builtin_amdgcn_rsq(v);
  y0[i] = y;
  double t = v*y; double e = fma(-t,y,1.0); y = fma(0.5*y,e,y);
  t = v*y; e = fma(-t,y,1.0); y = fma(0.5*y,e,y);
  y2[i] = y;
}
int main(){
  const int n=8; double hx[n]={1.0,2.0,3.7e3,1e14,2.5e16,7e-3,123456.789,9.99e9};
  double *dx,*d0,*d2; hipMalloc(&dx,n*8);hipMalloc(&d0,n*8);hipMalloc(&d2,n*8);
  hipMemcpy(dx,hx,n*8,hipMemcpyHostToDevice);
  k<<<1,64>>>(dx,d0,d2,n);
  double h0[n],h2[n]; hipMemcpy(h0,d0,n*8,hipMemcpyDeviceToHost); hipMemcpy(h2,d2,n*8,hipMemcpyDeviceToHost);
  for(int i=0;i<n;i++){ double r=1.0/sqrt(hx[i]); printf("x=%g rsq relerr %.3e  nr2 relerr %.3e\n",hx[i],fabs(h0[i]-r)/r,fabs(h2[i]-r)/r);}
}
