#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "../../blindshadowremoval_amd/hostsrc/inflate.c"
int main(int argc,char**argv){ FILE*f=fopen(argv[1],"rb"); fseek(f,0,SEEK_END); long n=ftell(f); fseek(f,0,SEEK_SET); uint8_t*z=malloc(n); if(fread(z,1,n,f)){} size_t m=atol(argv[2]);
 unsigned s=12345; int ok=0,bad=0;
 for(int it=0;it<20000;it++){ s=s*1103515245u+12345u; long len=n; uint8_t*zz; int k=(s>>16)%4;
   if(k==0){ len=(s>>8)%n; }
   zz=malloc(len+16); memcpy(zz,z,len); memset(zz+len,0,16);
   if(k==1){ for(int j=0;j<3;j++){ s=s*1103515245u+12345u; zz[(s>>8)%len]^=1u<<((s>>4)&7);} }
   if(k==2){ s=s*1103515245u+12345u; zz[(s>>8)%len]=(uint8_t)(s>>3); }
   size_t mm = m; if(k==3){ s=s*1103515245u+12345u; mm = (s>>8)%(2*m)+1; }
   uint8_t*o=malloc(mm+16); int rc=bsr_inflate_zlib(zz,len,o,mm); if(rc==0) ok++; else bad++; free(o); free(zz); }
 printf("ok %d bad %d\n",ok,bad); return 0; }
