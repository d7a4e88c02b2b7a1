#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "../../blindshadowremoval_amd/hostsrc/png_unfilter.c"
int main(void){ unsigned s=777; int ok=0,bad=0;
 for(int it=0;it<20000;it++){ s=s*1103515245u+12345u; int bpp=(int[]){1,3,4,2}[(s>>8)&3]; s=s*1103515245u+12345u; int w=1+(s>>8)%70, h=1+(s>>16)%40; int rb=w*bpp;
   uint8_t*raw=malloc((size_t)h*(rb+1)); uint8_t*out=malloc((size_t)h*rb);
   for(int i=0;i<h*(rb+1);i++){ s=s*1103515245u+12345u; raw[i]=(uint8_t)(s>>11); }
   for(int y=0;y<h;y++){ s=s*1103515245u+12345u; raw[(size_t)y*(rb+1)]=(uint8_t)((s>>9)%6); }
   int rc=bsr_png_unfilter(raw,h,rb,bpp,out); if(rc==0) ok++; else bad++; free(raw); free(out); }
 printf("ok %d bad %d\n",ok,bad); return 0; }
