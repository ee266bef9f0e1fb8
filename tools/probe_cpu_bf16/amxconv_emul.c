// emulation of oneDNN brg_conv_fwd:avx10_1_512_amx bf16 accumulation order (probe)
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <alloca.h>
static inline float bf2f(uint16_t h){ uint32_t u=(uint32_t)h<<16; float f; memcpy(&f,&u,4); return f; }
static inline uint16_t f2bf(float f){ uint32_t u; memcpy(&u,&f,4); if((u&0x7fffffff)>0x7f800000) return 0x7fc0; u += 0x7fff + ((u>>16)&1); return (uint16_t)(u>>16); }
// x: [B][H][W][IC] float (bf16-exact), wt: [KH][KW][IC][OC] float, bias [OC] float; y: [B][OH][OW][OC] bf16 bits
// order 0: for kh,kw,icb ; 1: for icb,kh,kw ; 2: single flattened sequence (kw,kh,ic) chunked by 32
void conv_emul(const float* x, const float* wt, const float* bias, uint16_t* y, int B,int H,int W,int IC,int OC,int KH,int KW,
               int stride,int pad,int OH,int OW,int order, float* yf)
{
  int nicb = (IC+31)/32;
  #pragma omp parallel for collapse(2) schedule(static)
  for(int b=0;b<B;b++) for(int oy=0;oy<OH;oy++){
    float *C=(float*)malloc(sizeof(float)*OC*4), *te=C+OC, *to=C+2*OC, *S=C+3*OC;
    for(int ox=0;ox<OW;ox++){
      for(int o=0;o<OC;o++) C[o]=0.f;
      if(order==2){
        int pos=0; for(int o=0;o<OC;o++){te[o]=0.f;to[o]=0.f;}
        for(int kw=0;kw<KW;kw++) for(int kh=0;kh<KH;kh++) for(int c=0;c<IC;c++){
          int iy=oy*stride-pad+kh, ix=ox*stride-pad+kw;
          float xv = (iy>=0&&iy<H&&ix>=0&&ix<W)? x[((size_t)(b*H+iy)*W+ix)*IC+c] : 0.f;
          const float* wr = wt + ((size_t)(kh*KW+kw)*IC+c)*OC;
          float* t = (pos&1)? to: te;
          for(int o=0;o<OC;o++) t[o] = t[o] + xv*wr[o];
          pos++;
          if(pos==32){ for(int o=0;o<OC;o++){ C[o] = C[o] + (te[o]+to[o]); te[o]=0.f; to[o]=0.f;} pos=0; }
        }
        if(pos) for(int o=0;o<OC;o++) C[o] = C[o] + (te[o]+to[o]);
      } else {
        int n1 = order==0? KH*KW : nicb, n2 = order==0? nicb : KH*KW;

        for(int a=0;a<n1;a++){ if(order==3) for(int o=0;o<OC;o++) S[o]=0.f;
         for(int bb=0;bb<n2;bb++){
          int tap = order==0? a: bb, icb = order==0? bb: a;
          int kh=tap/KW, kw=tap%KW;
          int iy=oy*stride-pad+kh, ix=ox*stride-pad+kw;
          int inb = (iy>=0&&iy<H&&ix>=0&&ix<W);
          for(int o=0;o<OC;o++){te[o]=0.f;to[o]=0.f;}
          if(inb){
            const float* xr = x + ((size_t)(b*H+iy)*W+ix)*IC + icb*32;
            const float* wr = wt + ((size_t)(kh*KW+kw)*IC + icb*32)*OC;
            int kc = IC-icb*32 < 32 ? IC-icb*32 : 32;
            for(int k=0;k<kc;k+=2){
              float x0=xr[k], x1 = (k+1<kc)? xr[k+1]:0.f;
              const float* w0=wr+(size_t)k*OC; const float* w1=wr+(size_t)(k+1<kc?k+1:k)*OC;
              for(int o=0;o<OC;o++){ te[o] = te[o] + x0*w0[o]; to[o] = to[o] + x1*w1[o]; }
            }
          }
          if(order==3) for(int o=0;o<OC;o++) S[o] = S[o] + (te[o]+to[o]);
          else for(int o=0;o<OC;o++) C[o] = C[o] + (te[o]+to[o]);
         }
         if(order==3) for(int o=0;o<OC;o++) C[o] = C[o] + S[o];
        }
      }
      uint16_t* yr = y + ((size_t)(b*OH+oy)*OW+ox)*OC;
      for(int o=0;o<OC;o++){ float v = C[o] + bias[o]; if(yf) yf[((size_t)(b*OH+oy)*OW+ox)*OC+o]=v; yr[o]=f2bf(v); }
    }
    free(C);
  }
}
