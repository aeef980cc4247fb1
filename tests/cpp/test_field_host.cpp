// Host-side check of the product's field arithmetic (zk_amd/csrc/field.cuh compiled for the CPU, no GPU needed):
// the carry-free 9x29-bit multiplier fe_mul29(a, prepare(c)) must equal the saturated Montgomery product fe_mul(a, c)
// bit for bit -- 200k random pairs plus edge values (0, 1, 2, p-1) per field, all three fields -- and the two-product form
// fe_mul_tt(a, c) (left operand split five bits lower, nothing prepared: prod_reduce) must equal it too, and
// fe_dot2_29(a, prepare(c), a2, prepare(c2)) must equal fe_mul(a, c) + fe_mul(a2, c2) (one reduction for both products).  (fe_mul itself is pinned
// against the oracle through the C ABI host helpers and every GPU parity test.)  Built with clang++ (field.cuh uses
// __builtin_addc).  Driven by tests/test_gpu_cpp_host.py.
#include <cstdio>
#include <cstdlib>
#include "../../zk_amd/csrc/host_field.hpp"
using namespace zk;
static uint64_t sm(uint64_t& x){ uint64_t z=(x+=0x9E3779B97F4A7C15ULL); z=(z^(z>>30))*0xBF58476D1CE4E5B9ULL; z=(z^(z>>27))*0x94D049BB133111EBULL; return z^(z>>31);}
int main(){
  int bad=0;
  for(int f=0;f<3;++f){ const FieldInfo* fi=field_info(f); const FieldParams& P=fi->P; uint64_t st=f+1;
    auto rnd=[&](){ Fe c; while(true){ for(int i=0;i<4;++i){uint64_t x=sm(st); c.v[2*i]=(uint32_t)x; c.v[2*i+1]=(uint32_t)(x>>32);} uint32_t top=P.bits&31; if(top) c.v[7]&=(1u<<top)-1; Fe d; if(sub8(d.v,c.v,P.p)) break;} return c; };
    Fe pm1; { uint32_t one[8]={1,0,0,0,0,0,0,0}; sub8(pm1.v,P.p,one); }
    Fe edge[4]={fe_zero(), fe_one(P), pm1, fe_from_u32(2,P)};
    for(int it=0;it<200000;++it){ Fe a = it<16? edge[it&3] : rnd(); Fe c = it<16? edge[(it>>2)&3] : rnd();
      Fe want=fe_mul(a,c,P); Fe got=fe_mul29(a,mul29_prepare(c,P),P); if(!fe_eq(want,got)){ if(bad<5) printf("MISMATCH field %d it %d\n",f,it); ++bad; }
      Fe gtt=fe_mul_tt(a,c,P); if(!fe_eq(want,gtt)){ if(bad<5) printf("MUL_TT MISMATCH field %d it %d\n",f,it); ++bad; }   // table x table product (prod_reduce)
      Fe a2 = it<64? edge[(it>>4)&3] : rnd(); Fe c2 = it<64? edge[(it>>2)&3] : rnd();
      Fe want2=fe_add(want,fe_mul(a2,c2,P),P); Fe got2=fe_dot2_29(a,mul29_prepare(c,P),a2,mul29_prepare(c2,P),P); if(!fe_eq(want2,got2)){ if(bad<5) printf("DOT2 MISMATCH field %d it %d\n",f,it); ++bad; } }
    printf("field %d inv29=%08x ok\n",f,P.inv29);
  }
  printf(bad?"FAILED %d\n":"all equal%.0d\n",bad); return bad!=0; }
