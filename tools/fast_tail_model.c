// CPU model of the fast Julia distance tail: validates error bound & fallback logic against glibc.
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
static inline float asf(uint32_t u){float f;memcpy(&f,&u,4);return f;}
static inline uint32_t asu(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static uint64_t s=88172645463325252ull;
static inline uint64_t rnd(){s^=s<<13;s^=s>>7;s^=s<<17;return s;}
static float exact(float m2,float d2){ double t=sqrt((double)(m2/d2))*0.5*log(sqrt((double)m2)); return (float)t; }
static int n_fallback=0;
// returns 1 and *out when the fast path decides, 0 for fallback
static double g_maxrel=0; static long g_nfast=0;
static int fast(float m2,float d2,float*out,int perturb){
  float q=m2/d2;
  if(d2==INFINITY && m2>0.0f && m2<INFINITY){ *out=m2<1.0f?-0.0f:0.0f; return 1; }
  if(!(m2>=1e-30f && m2<INFINITY && q>=1e-30f && q<1e30f)) return 0;
  // sqrt(q) in double: rsq seed in float
  float y0f=1.0f/sqrtf(q); 
  if(perturb){ uint32_t u=asu(y0f); u+= (int)(rnd()%9)-4; y0f=asf(u);} // +-4ulp
  double x=(double)q, y=(double)y0f;
  double g=x*y, h=0.5*y;
  double r=fma(-h,g,0.5);
  g=fma(g,r,g); h=fma(h,r,h);
  r=fma(-h,g,0.5);
  g=fma(g,r,g); h=fma(h,r,h);
  double d=fma(-g,g,x);
  g=fma(d,h,g);            // sqrt(q), ~0.5ulp
  // ln(m2)
  double xm=sqrt((double)m2);   // correctly rounded, as the reference forms it
  int e; double m=frexp(xm,&e); // m in [0.5,1)
  if(m<0.70710678118654752){ m*=2.0; e-=1; }  // m in [0.707,1.414)
  double num=m-1.0, den=m+1.0;
  float rf=1.0f/(float)den; if(perturb){uint32_t u=asu(rf); u+=(int)(rnd()%9)-4; rf=asf(u);}
  double ry=(double)rf;
  double er=fma(-den,ry,1.0); ry=fma(ry,er,ry);
  er=fma(-den,ry,1.0); ry=fma(ry,er,ry);
  double sN=num*ry;
  double s2=sN*sN;
  double p=1.0/19.0;
  p=fma(p,s2,1.0/17.0); p=fma(p,s2,1.0/15.0); p=fma(p,s2,1.0/13.0); p=fma(p,s2,1.0/11.0);
  p=fma(p,s2,1.0/9.0); p=fma(p,s2,1.0/7.0); p=fma(p,s2,1.0/5.0); p=fma(p,s2,1.0/3.0);
  p=p*s2; // series minus leading 1
  double lnm=2.0*fma(sN,p,sN);
  double ln=fma((double)e,0.69314718055994530942,lnm);
  double E=g*0.5*ln;   // sqrt(q)*0.5 * (0.5*ln(m2))
  { long double T=sqrtl((long double)q)*0.5L*logl((long double)sqrt((double)m2)); double rel=fabs((double)((E-T)/T)); if(rel>g_maxrel) g_maxrel=rel; }
  float f=(float)E;
  uint32_t fb=asu(f);
  uint32_t ex=fb&0x7f800000u;
  if((fb&0x007fffffu)==0 || ex< (40u<<23) || ex==0x7f800000u) return 0;
  float hf=asf(ex-(24u<<23));
  double rr=fabs(E-(double)f);
  if(!((double)hf-rr > fabs(E)*0x1p-44)) return 0;
  *out=f; return 1;
}
// The march's lean tail (julia_distance_tail_lean): ln(m2) from the float's own exponent and mantissa, no double
// square root of m2.  Returns 0 where the device function hands over to fast().
static int lean(float m2,float d2,float*out,int perturb){
  if(!(m2>=16.0f && m2<1e30f)) return 0;
  float q=m2/d2;
  if(!(q>=1e-30f && q<1e30f)) return 0;
  float y0f=1.0f/sqrtf(q);
  if(perturb){ uint32_t u=asu(y0f); u+= (int)(rnd()%9)-4; y0f=asf(u);}
  double x=(double)q, y=(double)y0f;
  double g=x*y, h=0.5*y;
  double r=fma(-h,g,0.5);
  g=fma(g,r,g); h=fma(h,r,h);
  r=fma(-h,g,0.5);
  g=fma(g,r,g);
  int e; float mf=frexpf(m2,&e);
  if(mf<0.70710678f){ mf*=2.0f; e-=1; }
  double m=(double)mf;
  double num=m-1.0, den=m+1.0;
  float rf=1.0f/(float)den; if(perturb){uint32_t u=asu(rf); u+=(int)(rnd()%9)-4; rf=asf(u);}
  double ry=(double)rf;
  double er=fma(-den,ry,1.0); ry=fma(ry,er,ry);
  er=fma(-den,ry,1.0); ry=fma(ry,er,ry);
  double sN=num*ry;
  double s2=sN*sN;
  double p=1.0/19.0;
  p=fma(p,s2,1.0/17.0); p=fma(p,s2,1.0/15.0); p=fma(p,s2,1.0/13.0); p=fma(p,s2,1.0/11.0);
  p=fma(p,s2,1.0/9.0); p=fma(p,s2,1.0/7.0); p=fma(p,s2,1.0/5.0); p=fma(p,s2,1.0/3.0);
  p=p*s2;
  double lnm=2.0*fma(sN,p,sN);
  double ln=fma((double)e,0.69314718055994530942,lnm);
  double E=g*0.25*ln;
  { long double T=sqrtl((long double)q)*0.5L*logl((long double)sqrt((double)m2)); double rel=fabs((double)((E-T)/T)); if(rel>g_maxrel) g_maxrel=rel; }
  float f=(float)E;
  uint32_t fb=asu(f);
  uint32_t ex=fb&0x7f800000u;
  if((fb&0x007fffffu)==0 || ex< (40u<<23) || ex==0x7f800000u) return 0;
  float hf=asf(ex-(24u<<23));
  double rr=fabs(E-(double)f);
  if(!((double)hf-rr > fabs(E)*0x1p-44)) return 0;
  *out=f; return 1;
}
static int lean_or_fast(float m2,float d2,float*out,int perturb){ return lean(m2,d2,out,perturb) || fast(m2,d2,out,perturb); }
int main(int argc,char**argv){
  long n=argc>1?atol(argv[1]):100000000; long bad=0,fb=0;
  const int generic=argc<=2 || argv[2][0]=='a';   /* 'any': arbitrary operands; 'march'/'n...': operands as the march makes them */
  for(long i=0;i<n;i++){
    // m2: mix of escaped values (1e10..1e20), moderate, and arbitrary
    uint64_t r=rnd(); float m2,d2;
    int mode=(!generic)?0:(r&7);
    if(!generic && argv[2][0]=='n'){ m2=expf((float)((rnd()>>11)*(1.0/9007199254740992.0))*6.0f-3.0f); }
    else if(mode<5){ m2=1e10f*expf((float)((rnd()>>11)*(1.0/9007199254740992.0))*23.0f); }
    else if(mode<7){ m2=expf((float)((rnd()>>11)*(1.0/9007199254740992.0))*40.0f-5.0f); }
    else { m2=asf((uint32_t)rnd()&0x7fffffffu); }
    // randomise mantissa fully
    m2=asf((asu(m2)&0xff800000u)|((uint32_t)rnd()&0x7fffffu));
    d2=(!generic)? m2*expf((float)((rnd()>>11)*(1.0/9007199254740992.0))*40.0f-10.0f) : expf((float)((rnd()>>11)*(1.0/9007199254740992.0))*170.0f-85.0f);
    d2=asf((asu(d2)&0xff800000u)|((uint32_t)rnd()&0x7fffffu));
    if(generic && (r>>8&1023)==0) d2=INFINITY;
    float fo; float ex=exact(m2,d2);
    const int use_lean=argc>3 && argv[3][0]=='l';
    if(use_lean ? lean_or_fast(m2,d2,&fo,1) : fast(m2,d2,&fo,1)){ if(asu(fo)!=asu(ex)){ bad++; if(bad<10) printf("BAD m2=%a d2=%a fast=%a exact=%a\n",m2,d2,fo,ex);} }
    else fb++;
  }
  printf("max rel err of E vs long double: %g = 2^%.1f\n",g_maxrel,log2(g_maxrel));
  printf("n=%ld bad=%ld fallback=%ld (%.3g)\n",n,bad,fb,(double)fb/n);
  return bad!=0;
}
