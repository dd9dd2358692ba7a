#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
static inline void my_sincos(float a, float*s, float*c){
  const float TWO_O_PI=0.636619747f;
  const float P1=1.57079637f, P2=-4.37113883e-8f, P3=-1.71512489e-15f; /* pi/2 = P1+P2+P3 */
  float n=rintf(a*TWO_O_PI);
  float r=fmaf(-n,P1,a); r=fmaf(-n,P2,r); r=fmaf(-n,P3,r);
  float z=r*r;
  const float S1=-0.166666666416265235595f,S2=0.0083333293858894631756f,S3=-0.000198393348360966317347f,S4=0.0000027183114939898219064f;
  const float C0=-0.499999997251031003120f,C1=0.0416666233237390631894f,C2=-0.00138867637746099294692f,C3=0.0000243904487962774090654f;
  float sp=fmaf(z,fmaf(z,fmaf(z,S4,S3),S2),S1); float sn=fmaf(r*z,sp,r);
  float cp=fmaf(z,fmaf(z,fmaf(z,C3,C2),C1),C0); float cs=fmaf(z,cp,1.0f);
  int q=(int)n; float ss=(q&1)?cs:sn, cc=(q&1)?sn:cs; if(q&2) ss=-ss; if((q+1)&2) cc=-cc; *s=ss; *c=cc; }
int main(){ FILE*f=fopen("divs.txt","r"); uint32_t db[16]; for(int i=0;i<16;i++) if(fscanf(f,"%u",&db[i])!=1) return 1;
 double maxs=0,maxc=0,maxs_lib=0; double sums=0; long cnt=0; srand(1);
 for(int it=0;it<4000000;it++){ float x = (it&1)? (float)(300.0*rand()/RAND_MAX) : (float)((2.0*rand()/RAND_MAX-1.0)*3.14159265); float xs=x*6.28318548202514648f;
  for(int k=0;k<16;k++){ float d; memcpy(&d,&db[k],4); float a=xs/d; float s,c; my_sincos(a,&s,&c); double es=fabs((double)s-sin((double)a)), ec=fabs((double)c-cos((double)a)); if(es>maxs)maxs=es; if(ec>maxc)maxc=ec; double el=fabs((double)sinf(a)-sin((double)a)); if(el>maxs_lib)maxs_lib=el; sums+=es; cnt++; } }
 printf("max abs err sin %.3e cos %.3e (libm sinf %.3e) mean %.3e\n",maxs,maxc,maxs_lib,sums/cnt); return 0; }
