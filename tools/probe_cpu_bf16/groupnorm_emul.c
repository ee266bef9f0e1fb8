#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
static inline float bf2f(uint16_t h){ uint32_t u=(uint32_t)h<<16; float f; memcpy(&f,&u,4); return f; }
static inline uint16_t f2bf(float f){ uint32_t u; memcpy(&u,&f,4); if((u&0x7fffffff)>0x7f800000) return 0x7fc0; u += 0x7fff + ((u>>16)&1); return (uint16_t)(u>>16); }
#define L 8
typedef struct { float v[L]; } vec;
static int VAR=0;
void set_var(int v){VAR=v;}
static void add_moments_vec(int64_t m0_add, const vec* m1_add, const vec* m2_add, int64_t* m0, vec* m1, vec* m2){
  int64_t n = *m0 + m0_add;
  float c = n==0 ? 0.f : (float)m0_add / (float)n;
  float m0f = (float)*m0;
  for(int l=0;l<L;l++){
    float delta = m1_add->v[l] - m1->v[l];
    float m2_tmp = m2->v[l] + m2_add->v[l];
    float c_delta = c*delta;
    float m0_delta = delta*m0f;
    m1->v[l] = (VAR&1)? fmaf(c, delta, m1->v[l]) : m1->v[l] + c_delta;
    m2->v[l] = fmaf(m0_delta, c_delta, m2_tmp);
  }
  *m0 = n;
}
static void add_moments(int64_t m0_add, float m1_add, float m2_add, int64_t* m0, float* m1, float* m2){
  int64_t n = *m0 + m0_add;
  float c = n==0? 0.f : (float)m0_add/(float)n;
  float delta = m1_add - *m1;
  *m1 = (VAR&2)? fmaf(c, delta, *m1) : *m1 + c*delta;
  if(VAR&4) *m2 = *m2 + fmaf(delta*delta*c, (float)(*m0), m2_add);
  else *m2 = *m2 + (m2_add + delta*delta*c*(float)(*m0));
  *m0 = n;
}
// X: bf16 bits, N elements (multiple of 32 assumed); returns mean and biased variance
void rowwise_moments_bf16(const uint16_t* X, int64_t N, float* mean, float* var){
  const int kVec=16, kChunk=16;
  int64_t n = N/kVec, m = (n + kChunk-1)/kChunk;
  int depth = 0; while(((int64_t)1<<depth) < m) depth++;
  int64_t m0_stk[64]; vec m1_stk[64], m2_stk[64];
  memset(m0_stk,0,sizeof m0_stk); memset(m1_stk,0,sizeof m1_stk); memset(m2_stk,0,sizeof m2_stk);
  for(int64_t i=0;i<m;i++){
    const uint16_t* Xp = X + i*kChunk*kVec;
    int64_t m0 = n - i*kChunk < kChunk ? n - i*kChunk : kChunk;
    vec a1={{0}}, b1={{0}}, a2={{0}}, b2={{0}};
    for(int64_t j=0;j<m0;j++){
      float cj = 1.0f/(float)(j+1);
      for(int l=0;l<L;l++){
        float x0 = bf2f(Xp[j*kVec + l]), x1 = bf2f(Xp[j*kVec + 8 + l]);
        float d0 = x0 - a1.v[l], d1 = x1 - b1.v[l];
        a1.v[l] = fmaf(d0, cj, a1.v[l]); b1.v[l] = fmaf(d1, cj, b1.v[l]);
        float e0 = x0 - a1.v[l], e1 = x1 - b1.v[l];
        a2.v[l] = fmaf(d0, e0, a2.v[l]); b2.v[l] = fmaf(d1, e1, b2.v[l]);
      }
    }
    add_moments_vec(m0, &a1, &a2, &m0_stk[0], &m1_stk[0], &m2_stk[0]);
    add_moments_vec(m0, &b1, &b2, &m0_stk[0], &m1_stk[0], &m2_stk[0]);
    int64_t mask = i+1;
    for(int j=1; j<depth && (mask&1)==0; ++j){
      add_moments_vec(m0_stk[j-1], &m1_stk[j-1], &m2_stk[j-1], &m0_stk[j], &m1_stk[j], &m2_stk[j]);
      m0_stk[j-1]=0; memset(&m1_stk[j-1],0,sizeof(vec)); memset(&m2_stk[j-1],0,sizeof(vec));
      mask >>= 1;
    }
  }
  for(int i=1;i<depth;i++) add_moments_vec(m0_stk[i], &m1_stk[i], &m2_stk[i], &m0_stk[0], &m1_stk[0], &m2_stk[0]);
  int64_t m0=0; float m1=0.f, m2=0.f;
  for(int64_t i=n*kVec;i<N;i++){ float x=bf2f(X[i]); float delta=x-m1; ++m0; m1 += delta/(float)m0; m2 += delta*(x-m1); }
  int64_t m0_add = n*kVec/L;
  for(int l=0;l<L;l++) add_moments(m0_add, m1_stk[0].v[l], m2_stk[0].v[l], &m0, &m1, &m2);
  *mean = m1; *var = m2/(float)N;
}
// X [NB][C][HW] bf16, groups G; variant bit0: fma in apply; bit1: rstd via float sqrt instead of double
void group_norm_bf16(const uint16_t* X, const uint16_t* gamma, const uint16_t* beta, uint16_t* Y, int NB, int C, int64_t HW, int G, double eps, int variant, float* stats){
  int D = C/G;
  #pragma omp parallel for schedule(dynamic)
  for(int i=0;i<NB*G;i++){
    const uint16_t* Xp = X + (int64_t)i*D*HW;
    float mean, var; rowwise_moments_bf16(Xp, (int64_t)D*HW, &mean, &var);
    float rstd;
    if(variant&2) rstd = 1.0f/sqrtf(fmaxf(var,0.f) + (float)eps);
    else rstd = (float)(1.0/sqrt((double)fmaxf(var,0.f) + eps));
    if(stats){ stats[2*i]=mean; stats[2*i+1]=rstd; }
    int g = i%G;
    for(int j=0;j<D;j++){
      int c = g*D+j;
      float scale = rstd*bf2f(gamma[c]);
      float bias = -scale*mean + bf2f(beta[c]);
      if(variant&4) bias = fmaf(-scale, mean, bf2f(beta[c]));
      const uint16_t* xp = X + ((int64_t)i*D+j)*HW; uint16_t* yp = Y + ((int64_t)i*D+j)*HW;
      if(variant&1) for(int64_t k=0;k<HW;k++) yp[k] = f2bf(fmaf(scale, bf2f(xp[k]), bias));
      else for(int64_t k=0;k<HW;k++) yp[k] = f2bf(scale*bf2f(xp[k]) + bias);
    }
  }
}
