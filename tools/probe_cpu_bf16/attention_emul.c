#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
static inline float bf2f(uint16_t h){ uint32_t u=(uint32_t)h<<16; float f; memcpy(&f,&u,4); return f; }
static inline uint16_t f2bf(float f){ uint32_t u; memcpy(&u,&f,4); if((u&0x7fffffff)>0x7f800000) return 0x7fc0; u += 0x7fff + ((u>>16)&1); return (uint16_t)(u>>16); }
static inline float u2f(uint32_t u){ float f; memcpy(&f,&u,4); return f; }
static float fexp_u20(float x){
  const float c0=0.00010703434948458272f, c1=0.30354260500649682f, c2=-0.22433836478672356f, c3=-0.079204240219773236f;
  const float log2e = u2f(0x3fb8aa3b), a = 8388608.0f, b = 8388608.0f*127.f;
  float src = x*log2e;
  float fr = src - floorf(src);
  float res = fmaf(fr, c3, c2); res = fmaf(fr, res, c1); res = fmaf(fr, res, c0);
  src = src - res;
  float tmp = fmaf(a, src, b);
  int32_t ci = (int32_t)tmp;           // cvttps
  if(x < u2f(0xc2aeac50)) ci = 0;
  if(x > u2f(0x42b17218)) ci = 0x7F800000;
  float r; memcpy(&r,&ci,4); return r;
}
// AMX-style dot: chunks of 32 sequential; inside even/odd chains. a,b float arrays (bf16-exact), stride sb for b
static inline float amx_dot(const float* a, const float* b, int K, int sb, float c0){
  float C = c0;
  for(int k0=0;k0<K;k0+=32){
    float te=0.f,to=0.f;
    for(int k=k0;k<k0+32 && k<K;k+=2){ te = te + a[k]*b[(size_t)k*sb]; to = to + a[k+1]*b[(size_t)(k+1)*sb]; }
    C = C + (te+to);
  }
  return C;
}
// params: lanes (8/16) for the sum; pv_mode 0: C=dst then += chunks ; 1: D=sum chunks from 0; C = dst + D
void attn_emul(const uint16_t* qb, const uint16_t* kb, const uint16_t* vb, uint16_t* ob, int B, int T, int Cd, int lanes, int pv_mode, int qsplit, int kvsplit){
  float scale = (float)(1.0/sqrt((double)Cd));
  #pragma omp parallel for collapse(2) schedule(dynamic)
  for(int b=0;b<B;b++) for(int i=0;i<T;i++){
    float* q = (float*)malloc(sizeof(float)*Cd); float* s=(float*)malloc(sizeof(float)*kvsplit); float* p=(float*)malloc(sizeof(float)*kvsplit);
    float* dst=(float*)calloc(Cd,sizeof(float));
    static float *kf=0,*vf=0;
    for(int d=0;d<Cd;d++) q[d]=bf2f(qb[((size_t)b*T+i)*Cd+d]);
    float m_old = -INFINITY, sum_old = 0.f;
    for(int n0=0;n0<T;n0+=kvsplit){
      int nb = T-n0<kvsplit? T-n0: kvsplit;
      float bm = -INFINITY;
      for(int j=0;j<nb;j++){
        const uint16_t* kr = kb + ((size_t)b*T+n0+j)*Cd;
        float C=0.f;
        for(int k0=0;k0<Cd;k0+=32){ float te=0.f,to=0.f; for(int k=k0;k<k0+32;k+=2){ te = te + q[k]*bf2f(kr[k]); to = to + q[k+1]*bf2f(kr[k+1]); } C = C + (te+to); }
        s[j] = C*scale; if(s[j]>bm) bm=s[j];
      }
      float m_new = m_old > bm ? m_old : bm;
      // exp and sum with lane structure
      float lane[16]; for(int l=0;l<lanes;l++) lane[l]=0.f;
      int nv = nb/lanes*lanes;
      for(int j=0;j<nv;j++){ float e = fexp_u20(s[j]-m_new); lane[j%lanes] += e; p[j]=bf2f(f2bf(e)); }
      for(int st=lanes/2; st>=1; st/=2) for(int l=0;l<st;l++) lane[l] = lane[l] + lane[l+st];
      float tsum = lane[0];
      for(int j=nv;j<nb;j++){ float e = expf(s[j]-m_new); tsum += e; p[j]=bf2f(f2bf(e)); }
      float exp_tmp = expf(m_old - m_new);
      sum_old = (pv_mode&2)? fmaf(exp_tmp, sum_old, tsum) : tsum + exp_tmp*sum_old;
      m_old = m_new;
      if(n0>0) for(int d=0;d<Cd;d++) dst[d] = dst[d]*exp_tmp;
      for(int d=0;d<Cd;d++){
        float C = ((pv_mode&1)==0 && n0>0)? dst[d] : 0.f;
        for(int j0=0;j0<nb;j0+=32){ float te=0.f,to=0.f; for(int j=j0;j<j0+32;j+=2){ te = te + p[j]*bf2f(vb[((size_t)b*T+n0+j)*Cd+d]); to = to + p[j+1]*bf2f(vb[((size_t)b*T+n0+j+1)*Cd+d]); } C = C + (te+to); }
        dst[d] = ((pv_mode&1)==1 && n0>0)? dst[d] + C : C;
      }
    }
    float rs = 1.0f/sum_old;
    for(int d=0;d<Cd;d++) ob[((size_t)b*T+i)*Cd+d] = f2bf(dst[d]*rs);
    free(q);free(s);free(p);free(dst);
  }
}
