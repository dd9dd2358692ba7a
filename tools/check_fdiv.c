#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <omp.h>
int main(){ FILE*f=fopen("divs.txt","r"); uint32_t db[16]; for(int i=0;i<16;i++) fscanf(f,"%u",&db[i]);
 for(int k=0;k<16;k++){ float d; memcpy(&d,&db[k],4); float r=1.0f/d; long bad=0; uint32_t firstbad=0;
  uint32_t lo=0x33800000u, hi=0x45800000u; /* [min normal, 4096) */
  #pragma omp parallel for reduction(+:bad)
  for(uint32_t b=lo;b<hi;b++){ float x; memcpy(&x,&b,4); float q=x/d; float q0=x*r; float rem=fmaf(-q0,d,x); float q1=fmaf(rem,r,q0); if(q1!=q){ bad++; if(!firstbad) firstbad=b; } }
  printf("d=%.9g r=%.9g bad=%ld first=%08x\n",d,r,bad,firstbad); }
 return 0; }
