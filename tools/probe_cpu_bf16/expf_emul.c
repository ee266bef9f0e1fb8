#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#define N 32
static uint64_t T[N];
static inline double asdouble(uint64_t u){ double d; memcpy(&d,&u,8); return d; }
static inline uint64_t asuint64(double d){ uint64_t u; memcpy(&u,&d,8); return u; }
static float expf_emul(float x, int fused){
  const double InvLn2N = 0x1.71547652b82fep+0 * N, Shift = 0x1.8p+52;
  const double C0 = 0x1.c6af84b912394p-5/N/N/N, C1 = 0x1.ebfce50fac4f3p-3/N/N, C2 = 0x1.62e42ff0c52d6p-1/N;
  double xd = (double)x;
  double z = InvLn2N*xd;
  double kd = z + Shift; uint64_t ki = asuint64(kd); kd -= Shift;
  double r = z - kd;
  uint64_t t = T[ki % N]; t += ki << (52-5);
  double s = asdouble(t);
  double y;
  if(fused){ z = fma(C0, r, C1); double r2 = r*r; y = fma(C2, r, 1.0); y = fma(z, r2, y); y = y*s; }
  else { z = C0*r + C1; double r2=r*r; y = C2*r + 1; y = z*r2 + y; y = y*s; }
  return (float)y;
}
int main(){
  for(int i=0;i<N;i++){ long double v = powl(2.0L, (long double)i/N); T[i] = asuint64((double)v) - ((uint64_t)i << 47); }
  printf("T[1]=%016lx T[31]=%016lx\n", T[1], T[31]);
  srand(1); long bad0=0,bad1=0,n=0;
  for(long it=0; it<40000000; it++){
    float x = -((float)rand()/RAND_MAX)*30.0f; if(it%3==0) x = -((float)rand()/RAND_MAX)*0.5f; if(it%7==0) x = ((float)rand()/RAND_MAX)*20.f;
    float e = expf(x); float a=expf_emul(x,0), b=expf_emul(x,1);
    n++; if(memcmp(&e,&a,4)) bad0++; if(memcmp(&e,&b,4)) bad1++;
  }
  printf("n=%ld mismatches unfused=%ld fused=%ld\n", n, bad0, bad1);
  float x0=0.f; printf("expf(0)=%g emul=%g\n", expf(x0), expf_emul(x0,1));
  return 0;
}
