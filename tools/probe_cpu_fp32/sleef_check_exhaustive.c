#include <stdio.h>
#include <stdlib.h>
#include <dlfcn.h>
#include <immintrin.h>
#include "sleef_emul.h"
typedef __m512 (*vf)(__m512);
int main(int argc, char** argv){
  void* h = dlopen("/usr/local/lib/python3.10/dist-packages/torch/lib/libtorch_cpu.so", RTLD_NOW | RTLD_GLOBAL);
  if(!h){ printf("dlopen: %s\n", dlerror()); return 1; }
  vf fexp = (vf)dlsym(h, "Sleef_expf16_u10avx512f"), ftanh = (vf)dlsym(h, "Sleef_tanhf16_u10avx512f");
  if(!fexp || !ftanh){ printf("dlsym failed\n"); return 1; }
  long bad_exp = 0, bad_tanh = 0, nan_mism = 0;
  #pragma omp parallel for reduction(+:bad_exp,bad_tanh,nan_mism) schedule(dynamic, 4096)
  for(long blk = 0; blk < (1L<<32)/16; blk++){
    float in[16] __attribute__((aligned(64))), oe[16] __attribute__((aligned(64))), ot[16] __attribute__((aligned(64)));
    for(int l=0;l<16;l++) in[l] = u2f_((uint32_t)(blk*16 + l));
    __m512 v = _mm512_load_ps(in);
    _mm512_store_ps(oe, fexp(v)); _mm512_store_ps(ot, ftanh(v));
    for(int l=0;l<16;l++){
      float a = sleef_expf_u10(in[l]), b = sleef_tanhf_u10(in[l]);
      if(f2u_(a) != f2u_(oe[l])) { if(in[l]!=in[l]) nan_mism++; else { bad_exp++; if(bad_exp < 5) printf("exp x=%a got %a want %a\n", in[l], a, oe[l]); } }
      if(f2u_(b) != f2u_(ot[l])) { if(in[l]!=in[l]) nan_mism++; else { bad_tanh++; if(bad_tanh < 5) printf("tanh x=%a got %a want %a\n", in[l], b, ot[l]); } }
    }
  }
  printf("all 2^32 inputs: expf mismatches %ld, tanhf mismatches %ld (NaN-payload-only mismatches %ld)\n", bad_exp, bad_tanh, nan_mism);
  return 0;
}
