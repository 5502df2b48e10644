#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
int main(int argc,char**argv){
  FILE*f=fopen(argv[1],"rb"); size_t off=atol(argv[2]); uint32_t n=(1u<<24)-4096; uint8_t*in=malloc(n+64); fseek(f,off,SEEK_SET); n=fread(in,1,n,f); memset(in+n,0,64);
  uint32_t hts=1u<<24, *ht=calloc(hts,4); uint32_t h1=0; const uint32_t mm=5, shift1=(24-1)/mm+1, bucket=7, checkbits=8, mask=255;
  // initial h1: hash of first mm bytes  (LZBuffer::fill start: for i<minMatch: h1 update)
  for (uint32_t k=0;k<mm;++k) h1=(((h1*5)<<shift1)+(in[k]+1)*123456791u)&(hts-1);
  uint64_t hist[9]={0}, lenhist[9]={0}; uint64_t tot=0;
  for (uint32_t i=0;i<n;++i){
    int c=0;
    for (uint32_t k=0;k<=bucket;++k){ uint32_t p=ht[h1^k]; if (p && i+3<n && (p&mask)==(in[i+3]&mask)){ p>>=checkbits; if (p<i) ++c; } }
    hist[c]++; tot++;
    if (i+mm+4<n){ uint32_t ih=((i*1234547u)>>19)&bucket; ht[h1^ih]=(i<<checkbits)|(in[i+3]&mask); h1=(((h1*5)<<shift1)+(in[i+mm]+1)*123456791u)&(hts-1);}  
  }
  for(int c=0;c<=8;++c) printf("%d valid: %.4f\n",c,(double)hist[c]/tot);
  return 0;}
