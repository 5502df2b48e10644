// model: which positions would a "same offset as the previous position" predictor skip, vs what the greedy chain visits
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
int main(int argc,char**argv){
  FILE*f=fopen(argv[1],"rb"); size_t off=atol(argv[2]); uint32_t n=(1u<<24)-4096; uint8_t*in=malloc(n+64); fseek(f,off,SEEK_SET); n=fread(in,1,n,f); memset(in+n,0,64); fclose(f);
  // visited flags from trace file: uint32 triples (pos,len,off)
  f=fopen(argv[3],"rb"); fseek(f,0,SEEK_END); long sz=ftell(f); fseek(f,0,SEEK_SET); uint32_t*tr=malloc(sz); fread(tr,1,sz,f); fclose(f); long nt=sz/12;
  uint8_t*vis=calloc(n+1,1); { uint32_t cur=0; for(long t=0;t<nt;++t){ uint32_t p=tr[3*t],l=tr[3*t+1]; while(cur<p) vis[cur++]=1; vis[p]=2; cur=p+l; } while(cur<n) vis[cur++]=1; }
  uint32_t hts=1u<<24,*ht=calloc(hts,4),h1=0; const uint32_t mm=5,shift1=(24-1)/mm+1,bucket=7,checkbits=8,mask=255;
  for(uint32_t k=0;k<mm;++k) h1=(((h1*5)<<shift1)+(in[k]+1)*123456791u)&(hts-1);
  uint32_t prev[8]={0}; int nprev=0;
  uint64_t nvis=0, nskip=0, vis_skipped=0, vis_match_skipped=0, unv_eval=0, tot=0;
  for(uint32_t i=0;i<n;++i){
    uint32_t cand[8]; int nc=0;
    for(uint32_t k=0;k<=bucket;++k){ uint32_t p=ht[h1^k]; if(p && i+3<n && (p&mask)==(in[i+3]&mask)){ p>>=checkbits; if(p<i) cand[nc++]=p; } }
    int skip=0;
    for(int a=0;a<nc&&!skip;++a) for(int b=0;b<nprev;++b) if(cand[a]==prev[b]+1){ skip=1; break; }
    tot++; if(vis[i]) nvis++;
    if(skip){ nskip++; if(vis[i]) vis_skipped++; if(vis[i]==2) vis_match_skipped++; } else if(!vis[i]) unv_eval++;
    memcpy(prev,cand,sizeof cand); nprev=nc;
    if(i+mm+4<n){ uint32_t ih=((i*1234547u)>>19)&bucket; ht[h1^ih]=(i<<checkbits)|(in[i+3]&mask); h1=(((h1*5)<<shift1)+(in[i+mm]+1)*123456791u)&(hts-1);} }
  printf("visited %.3f  skipped %.3f  visited-but-skipped %.4f of all (%.3f of visited; match starts %.4f)  evaluated-unvisited %.3f\n",
    (double)nvis/tot,(double)nskip/tot,(double)vis_skipped/tot,(double)vis_skipped/nvis,(double)vis_match_skipped/tot,(double)unv_eval/tot);
  return 0;}
